"""AutoModel plumbing (BASELINE configs[0]): a FunASR-format model directory is resolved through the registry by
name, the checkpoint loads strictly, `generate()` batches inputs and returns `[{key, text}]`. The executable spec in
the reference is tests/test_auto_model.py:43-71 (batch loop over `model.inference(...) -> (results, meta)`)."""
import os

import numpy as np
import pytest
import torch

from funasr_amd import synth
from funasr_amd.auto_model import AutoModel, prepare_data_iterator

from ._model_dir import VOCAB, make_model_dir, write_wav


@pytest.fixture(scope="module")
def model_dir(tmp_path_factory):
    d = str(tmp_path_factory.mktemp("paraformer_tiny"))
    info = make_model_dir(d)
    waves = {}
    for i, n in enumerate((40000, 24000, 56000)):
        p = os.path.join(d, f"utt{i}.wav")
        pcm = write_wav(p, synth.speech_like(n, seed=40 + i))
        waves[p] = pcm
    with open(os.path.join(d, "wav.scp"), "w") as f:
        for i, p in enumerate(waves):
            f.write(f"key_{i} {p}\n")
    return dict(dir=d, waves=waves, **info)


def test_prepare_data_iterator_forms(model_dir):
    d = model_dir["dir"]
    paths = list(model_dir["waves"])
    k, data = prepare_data_iterator(paths[0])
    assert k == ["utt0"] and data == [paths[0]]
    k, data = prepare_data_iterator(os.path.join(d, "wav.scp"))
    assert k == ["key_0", "key_1", "key_2"] and data == paths
    k, data = prepare_data_iterator([paths[1], np.zeros(16000, dtype=np.float32)])
    assert k[0] == "utt1" and k[1].startswith("rand_key_") and len(data) == 2
    with pytest.raises(FileNotFoundError):
        prepare_data_iterator("/nonexistent/a.wav")


def test_build_from_model_dir_on_cpu_and_loud_failure(model_dir):
    am = AutoModel(model=model_dir["dir"], device="cpu", disable_update=True)
    assert type(am.model).__name__ == "Paraformer"
    assert am.kwargs["batch_size"] == 1 and am.kwargs["device"] == "cpu"
    assert am.kwargs["tokenizer"].get_num_vocabulary_size() == len(VOCAB)
    assert am.kwargs["frontend"].output_size() == 560 and tuple(am.kwargs["frontend"].cmvn.shape) == (2, 560)
    assert am.model.vocab_size == len(VOCAB)
    got = am.model.state_dict()
    for k, v in model_dir["sd"].items():
        assert torch.equal(got[k], v), k                       # strict load incl. the training-only embedding
    # no CPU implementation behind the boundary: a clear error, not a silent fallback
    with pytest.raises(RuntimeError, match="GPU"):
        am.generate(input=list(model_dir["waves"])[0])


def test_unknown_model_and_vad_pipeline_raise(model_dir):
    with pytest.raises(FileNotFoundError):
        AutoModel(model="iic/not-a-local-dir", device="cpu")
    with pytest.raises(NotImplementedError):
        AutoModel(model=model_dir["dir"], vad_model="fsmn-vad", device="cpu")


@pytest.mark.gpu
def test_generate_matches_oracle_text(model_dir, cuda):
    from funasr_amd.tokenizer import CharTokenizer, sentence_postprocess
    from oracle import paraformer_oracle as O

    am = AutoModel(model=model_dir["dir"], device="cuda:0", batch_size=2)
    paths = list(model_dir["waves"])
    res = am.generate(input=os.path.join(model_dir["dir"], "wav.scp"))
    assert [r["key"] for r in res] == ["key_0", "key_1", "key_2"]
    assert am.speed_stats["rtf_avg"] is not None and float(am.speed_stats["rtf"]) >= 0
    # the same three clips through the CPU oracle (bs 1, like the reference on cpu) -> ids -> text
    cmvn = am.kwargs["frontend"].cmvn
    tok = CharTokenizer(token_list=VOCAB)
    for r, p in zip(res, paths):
        w = torch.from_numpy(model_dir["waves"][p].astype(np.float32) / 32768.0)
        feats, flens = O.wav_frontend([w], cmvn)
        ref = O.paraformer_greedy(feats, flens, model_dir["sd"], model_dir["cfg"])
        ids = [t for t in ref["raw_ids"][0] if t not in (0, 1, 2)]
        text, _ = sentence_postprocess(tok.ids2tokens(ids))
        assert r["text"] == text, (r, text)
    # mixed list input: path + ndarray + tensor, different batch size per call (runtime cfg override)
    x1 = model_dir["waves"][paths[1]].astype(np.float32) / 32768.0
    res2 = am.generate(input=[paths[0], x1, torch.from_numpy(x1)], batch_size=3)
    assert res2[0]["text"] == res[0]["text"] and res2[1]["text"] == res[1]["text"] == res2[2]["text"]
    assert am.kwargs["batch_size"] == 2 or am.kwargs["batch_size"] == 3
