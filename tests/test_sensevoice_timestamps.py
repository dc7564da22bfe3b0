"""SenseVoiceSmall `output_timestamp` (CTC forced alignment of the decoded pieces, model.py:1036-1112) against goldens made by
the REFERENCE class with its own SentencepiecesTokenizer (oracle/make_golden_sensevoice_ts.py): the host logic on injected
log-probabilities (CPU), whole inference on the GPU."""
import json
import os

import numpy as np
import pytest
import torch

from funasr_amd import synth
from funasr_amd.tokenizer import SentencepiecesTokenizer

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _gold():
    return np.load(os.path.join(GOLD, "sensevoice_ts.npz"), allow_pickle=False)


def _model(g):
    from funasr_amd.sense_voice import SenseVoiceSmall
    cfg = json.loads(str(g["config"]))
    sd = synth.sensevoice_state_dict(cfg, seed=int(g["seed"]))
    sd["ctc.ctc_lo.bias"][0] += float(g["ctc_blank_bias_add"])
    model = SenseVoiceSmall.from_config(cfg)
    model.load_state_dict(sd, strict=False)
    return model


def test_ctc_timestamps_equal_reference_on_injected_log_probabilities():
    g = _gold()
    tok = SentencepiecesTokenizer(os.path.join(GOLD, "sv_bpe.model"))
    model = _model(g)
    cases = json.loads(str(g["injected"]))
    assert len(cases) == 16 and all(c["has_ts"] for c in cases)
    merged = 0
    for ci, c in enumerate(cases):
        lp = g[f"logp_{ci}"]
        # the greedy text the reference decoded from this table (arg-max, unique_consecutive, blanks removed)
        ids = [int(k) for k, _ in __import__("itertools").groupby(lp.argmax(-1).tolist()) if k != 0]
        assert tok.decode(ids) == c["text"]
        stamps, words = model.ctc_timestamps(c["text"], lp[4:], tok)
        assert words == c["words"], (c["text"], words, c["words"])
        assert [[float(a), float(b)] for a, b in stamps] == c["timestamp"]
        merged += len(tok.text2tokens(c["text"])[4:]) - len(words)
    assert merged > 0                                             # pieces really were glued into words somewhere


def test_forced_alignment_properties():
    from funasr_amd.sense_voice import ctc_forced_align
    rng = np.random.default_rng(1)
    for _ in range(50):
        T, C = int(rng.integers(4, 30)), int(rng.integers(3, 9))
        L = int(rng.integers(1, min(T // 2, 6) + 1))
        tg = rng.integers(1, C, size=L)
        lp = np.log(rng.dirichlet(np.ones(C), size=T)).astype(np.float32)
        al = ctc_forced_align(lp, tg)
        collapsed = [k for k, _ in __import__("itertools").groupby(al.tolist())]
        assert [k for k in collapsed if k != 0] == tg.tolist() or L + int((tg[1:] == tg[:-1]).sum()) > T


@pytest.mark.gpu
def test_inference_with_output_timestamp_equals_reference(cuda):
    g = _gold()
    tok = SentencepiecesTokenizer(os.path.join(GOLD, "sv_bpe.model"))
    model = _model(g).to(cuda)
    feats, lens = torch.from_numpy(g["feats"]).to(cuda), torch.from_numpy(g["lens"])
    for mode in ("fp32", "f16x2"):
        model.set_precision(mode)
        res, _ = model.inference(feats, data_lengths=lens, key=[f"u{i}" for i in range(3)], tokenizer=tok, frontend=None,
                                 device=cuda, data_type="fbank", language="auto", output_timestamp=True)
        for r, e in zip(res, json.loads(str(g["e2e"]))):
            assert r["text"] == e["text"]
            assert ("timestamp" in r) == e["has_ts"]
            if e["has_ts"]:
                assert r["words"] == e["words"] and [[float(a), float(b)] for a, b in r["timestamp"]] == e["timestamp"], mode


def test_forced_alignment_equals_the_reference_function_on_random_emissions():
    """funasr/models/sense_voice/utils/ctc_alignment.py `ctc_forced_align` itself (build container only) on 300 random
    emission / target pairs: the host restatement returns the same frame labels"""
    from oracle import ref_import
    if not ref_import.available():
        pytest.skip("reference checkout not present (GPU box)")
    ref_import.install()
    from funasr.models.sense_voice.utils.ctc_alignment import ctc_forced_align as ref
    from funasr_amd.sense_voice import ctc_forced_align as mine
    g = torch.Generator().manual_seed(0)
    checked = 0
    for _ in range(300):
        T = int(torch.randint(1, 40, (1,), generator=g))
        C = int(torch.randint(3, 12, (1,), generator=g))
        L = int(torch.randint(1, max(2, min(T, 8) + 1), (1,), generator=g))
        lp = torch.log_softmax(torch.randn(1, T, C, generator=g) * 2, -1)
        tg = torch.randint(1, C, (1, L), generator=g)
        if L + int((tg[0, 1:] == tg[0, :-1]).sum()) > T:        # no valid path: the reference's output is unspecified
            continue
        r = ref(lp.clone(), tg.clone(), torch.tensor([T]), torch.tensor([L]))[0].numpy()
        assert np.array_equal(r, mine(lp[0].numpy(), tg[0].numpy())), (T, C, L)
        checked += 1
    assert checked > 150
