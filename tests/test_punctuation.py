"""CT-Transformer punctuation path against golden vectors from the reference's own CTTransformer class
(oracle/make_golden_punc.py): the network (oracle on the CPU, HIP on the GPU), the sentence assembly driven with the same
injected predictions as the reference, and the whole inference() end to end."""
import json
import os

import numpy as np
import pytest
import torch

from funasr_amd.ct_transformer import assemble, split_to_mini_sentence, split_words
from funasr_amd.tokenizer import CharTokenizer

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "punc.npz")


def _gold():
    g = np.load(GOLD, allow_pickle=False)
    return g, json.loads(str(g["vocab"])), json.loads(str(g["enc_cfg"]))


def test_split_words_and_mini_sentences():
    assert split_words("今天 hello世界 a b") == ["今", "天", "hello", "世", "界", "a", "b"]
    assert split_words("don't stop") == ["don't", "stop"] and split_words("   ") == []
    w = list(range(45))
    assert [len(x) for x in split_to_mini_sentence(w, 20)] == [20, 20, 5] and split_to_mini_sentence(w[:20], 20) == [w[:20]]


def test_oracle_network_equals_reference_golden():
    from oracle import punc_oracle
    g, vocab, enc = _gold()
    sd = punc_oracle.synthetic_state_dict(len(vocab), enc, seed=int(g["seed"]))
    y = punc_oracle.punc_forward(torch.from_numpy(g["ids"]), torch.from_numpy(g["lens"]), sd, enc)
    for b, n in enumerate(g["lens"].tolist()):
        assert (y[b, :n] - torch.from_numpy(g["logits"])[b, :n]).abs().max().item() < 2e-5


def test_sentence_assembly_equals_reference_with_the_same_injected_network():
    from oracle import punc_oracle
    g, vocab, _ = _gold()
    tok = CharTokenizer(token_list=vocab, unk_symbol="<unk>")
    cases = json.loads(str(g["cases"]))
    assert len(cases) == 60 and any(c["never_end"] and len(c["punc_array"]) > 200 for c in cases)
    for c in cases:
        words = split_words(c["text"])
        ids = tok.encode(words)
        text, marks = assemble(words, ids, lambda x, ne=c["never_end"]: punc_oracle.injected_marks(x, ne), punc_oracle.PUNC_LIST,
                               3, split_size=c["split_size"])
        assert text == c["out"], (c["text"], text, c["out"])
        assert [int(m) for m in marks] == c["punc_array"]


@pytest.mark.gpu
def test_network_and_inference_on_the_gpu_equal_reference(cuda):
    from funasr_amd.ct_transformer import CTTransformer
    from oracle import punc_oracle
    g, vocab, enc = _gold()
    model = CTTransformer(encoder="SANMEncoder", encoder_conf=dict(enc, input_layer="pe"), vocab_size=len(vocab),
                          punc_list=punc_oracle.PUNC_LIST, embed_unit=256, att_unit=256, ignore_id=0, sentence_end_id=3)
    sd = punc_oracle.synthetic_state_dict(len(vocab), enc, seed=int(g["seed"]))
    model.load_state_dict(sd, strict=True)
    model = model.to(cuda)
    y, _ = model.punc_forward(torch.from_numpy(g["ids"]), torch.from_numpy(g["lens"]))
    for b, n in enumerate(g["lens"].tolist()):
        assert (y[b, :n].cpu() - torch.from_numpy(g["logits"])[b, :n]).abs().max().item() < 5e-5
    tok = CharTokenizer(token_list=vocab, unk_symbol="<unk>")
    for c in json.loads(str(g["e2e"])):
        res, _ = model.inference([c["text"]], key=["k"], tokenizer=tok)
        assert res[0]["text"] == c["out"], (c["text"], res[0]["text"], c["out"])
        assert res[0]["punc_array"].tolist() == c["punc_array"]
    assert model.inference(["  "], key=["k"], tokenizer=tok)[0][0] == {"key": "k", "text": "", "punc_array": None}


@pytest.mark.gpu
def test_small_head_attention_kernel(cuda):
    from funasr_amd import ops
    g = torch.Generator().manual_seed(4)
    for B, Tq, Tk, H, dk, lens in ((2, 37, 37, 8, 32, [37, 5]), (1, 220, 220, 8, 32, [220]), (3, 9, 130, 4, 64, [130, 1, 64]),
                                   (1, 5, 700, 2, 16, [700])):
        q = torch.randn(B, Tq, H * dk, generator=g)
        k = torch.randn(B, Tk, H * dk, generator=g)
        v = torch.randn(B, Tk, H * dk, generator=g)
        kl = torch.tensor(lens, dtype=torch.int32)
        qh = q.double().view(B, Tq, H, dk).transpose(1, 2) * dk ** -0.5
        kh = k.double().view(B, Tk, H, dk).transpose(1, 2)
        vh = v.double().view(B, Tk, H, dk).transpose(1, 2)
        m = (torch.arange(Tk)[None, :] >= kl[:, None])[:, None, None, :]
        ref = (torch.softmax((qh @ kh.transpose(-1, -2)).masked_fill(m, float("-inf")), -1) @ vh).transpose(1, 2).reshape(B, Tq, H * dk)
        out = ops.attention_small(q.to(cuda), k.to(cuda), v.to(cuda), kl.to(cuda), H, dk ** -0.5).cpu()
        assert (out.double() - ref).abs().max().item() < 2e-5, (B, Tq, Tk, H, dk)
    tab = torch.randn(50, 256, generator=g).to(cuda)
    ids = torch.tensor([3, 0, 49, 7, 7], dtype=torch.int32)
    assert torch.equal(ops.gather_rows(tab, ids.to(cuda)).cpu(), tab.cpu()[ids.long()])


def _punc_dir(tmp_path):
    from oracle import punc_oracle
    from tests._model_dir import make_punc_model_dir
    g, vocab, enc = _gold()
    sd = punc_oracle.synthetic_state_dict(len(vocab), enc, seed=int(g["seed"]))
    d = str(tmp_path / "punc")
    make_punc_model_dir(d, vocab, enc, sd, punc_oracle.PUNC_LIST)
    return d, g


def test_punc_model_directory_builds(tmp_path):
    from funasr_amd.auto_model import AutoModel
    d, _ = _punc_dir(tmp_path)
    am = AutoModel(model=d, device="cpu")
    assert type(am.model).__name__ == "CTTransformer" and am.kwargs["tokenizer"].encode(["<unk>"]) is not None
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        am.generate(input="今天天气不错")


@pytest.mark.gpu
def test_automodel_punctuates_raw_text_like_the_reference(cuda, tmp_path):
    """AutoModel(model=<ct-punc dir>).generate(input="raw text") -- the reference's documented use of the punctuation
    model -- gives the text / punc_array the reference class gave for the same weights (tests/golden/punc.npz)"""
    from funasr_amd.auto_model import AutoModel
    d, g = _punc_dir(tmp_path)
    am = AutoModel(model=d, device="cuda:0")
    for c in json.loads(str(g["e2e"])):
        r = am.generate(input=c["text"])
        assert len(r) == 1 and r[0]["text"] == c["out"] and r[0]["punc_array"].tolist() == c["punc_array"]
