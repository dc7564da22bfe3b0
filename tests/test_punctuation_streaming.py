"""Realtime punctuation (CTTransformerStreaming over SANMVadEncoder) against golden vectors from the reference's own classes
(oracle/make_golden_punc_streaming.py -> tests/golden/punc_streaming.npz): the oracle's network restatement and the host-side
session logic on the CPU, the HIP network and whole sessions on the GPU."""
import json
import os

import numpy as np
import pytest
import torch

from funasr_amd.ct_transformer import assemble_streaming
from funasr_amd.tokenizer import CharTokenizer

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "punc_streaming.npz")


def _gold():
    g = np.load(GOLD, allow_pickle=False)
    return g, json.loads(str(g["vocab"])), json.loads(str(g["enc_cfg"]))


def test_oracle_vad_encoder_equals_reference_network():
    from oracle import punc_oracle
    g, vocab, enc = _gold()
    sd = punc_oracle.synthetic_state_dict(len(vocab), enc, seed=int(g["seed"]))
    for c in json.loads(str(g["nets"])):
        ids, lens = torch.tensor(c["ids"]), torch.tensor(c["lens"], dtype=torch.int32)
        y = punc_oracle.punc_forward_vad(ids, lens, c["vad"], sd, enc)
        ref = torch.tensor(c["logits"])
        for b, n in enumerate(c["lens"]):
            assert (y[b, :n] - ref[b, :n]).abs().max().item() < 2e-5
    # the masks matter: without them the same weights give other logits
    c = json.loads(str(g["nets"]))[0]
    plain = punc_oracle.punc_forward(torch.tensor(c["ids"]), torch.tensor(c["lens"], dtype=torch.int32), sd, enc)
    assert (plain[0, : c["lens"][0]] - torch.tensor(c["logits"])[0, : c["lens"][0]]).abs().max().item() > 1e-3


def test_session_logic_equals_reference_inference_with_injected_network():
    """every call of every session: returned text, punc_array and the carried words equal the reference class's, both sides
    driven by the same injected predictions"""
    from oracle import punc_oracle
    g, vocab, _ = _gold()
    tok = CharTokenizer(token_list=vocab, unk_symbol="<unk>")
    sessions = json.loads(str(g["sessions"]))
    assert len(sessions) == 30 and any(s["never_end"] for s in sessions)
    for s in sessions:
        cache = {}
        for c in s["calls"]:
            out, marks = assemble_streaming(c["text"], cache, tok.encode,
                                            lambda ids, vad, ne=s["never_end"]: punc_oracle.injected_marks(ids, ne),
                                            punc_oracle.PUNC_LIST, 3, split_size=s["split_size"])
            assert out == c["out"], (c["text"], out, c["out"])
            assert [int(m) for m in marks] == c["punc_array"]
            assert cache["pre_text"] == c["pre_text"]


@pytest.mark.gpu
def test_network_and_sessions_on_the_gpu_equal_reference(cuda):
    from funasr_amd.ct_transformer import CTTransformerStreaming
    from oracle import punc_oracle
    g, vocab, enc = _gold()
    model = CTTransformerStreaming(encoder="SANMVadEncoder", encoder_conf=dict(enc, input_layer="pe"), vocab_size=len(vocab),
                                   punc_list=punc_oracle.PUNC_LIST, embed_unit=256, att_unit=256, ignore_id=0, sentence_end_id=3)
    sd = punc_oracle.synthetic_state_dict(len(vocab), enc, seed=int(g["seed"]))
    model.load_state_dict(sd, strict=True)
    model = model.to(cuda)
    assert model.with_vad()
    for c in json.loads(str(g["nets"])):
        y, _ = model.punc_forward(torch.tensor(c["ids"]), torch.tensor(c["lens"], dtype=torch.int32), torch.tensor(c["vad"], dtype=torch.int32))
        ref = torch.tensor(c["logits"])
        for b, n in enumerate(c["lens"]):
            assert (y[b, :n].cpu() - ref[b, :n]).abs().max().item() < 5e-5, (c["lens"], c["vad"], b)
    tok = CharTokenizer(token_list=vocab, unk_symbol="<unk>")
    for calls in json.loads(str(g["e2e"])):
        cache = {}
        for c in calls:
            res, _ = model.inference([c["text"]], key=["k"], tokenizer=tok, cache=cache)
            assert res[0]["text"] == c["out"], (c["text"], res[0]["text"], c["out"])
            assert res[0]["punc_array"].reshape(-1).tolist() == c["punc_array"] and list(res[0]["punc_array"].shape) == c["punc_shape"]
            assert cache["pre_text"] == c["pre_text"]
    # the plain encoder entry point still runs unmasked after a masked forward (the mask is per call)
    from funasr_amd.ct_transformer import CTTransformer
    plain = CTTransformer(encoder="SANMEncoder", encoder_conf=dict(enc, input_layer="pe"), vocab_size=len(vocab),
                          punc_list=punc_oracle.PUNC_LIST, embed_unit=256, att_unit=256, ignore_id=0, sentence_end_id=3)
    plain.load_state_dict(sd, strict=True)
    plain = plain.to(cuda)
    c = json.loads(str(g["nets"]))[0]
    ids, lens = torch.tensor(c["ids"]), torch.tensor(c["lens"], dtype=torch.int32)
    y, _ = plain.punc_forward(ids, lens)
    ref = punc_oracle.punc_forward(ids, lens, sd, enc)
    assert (y[0, : c["lens"][0]].cpu() - ref[0, : c["lens"][0]]).abs().max().item() < 5e-5


def _dir(tmp_path):
    from oracle import punc_oracle
    from tests._model_dir import make_punc_model_dir
    g, vocab, enc = _gold()
    sd = punc_oracle.synthetic_state_dict(len(vocab), enc, seed=int(g["seed"]))
    d = str(tmp_path / "punc_rt")
    make_punc_model_dir(d, vocab, enc, sd, punc_oracle.PUNC_LIST, model="CTTransformerStreaming", encoder="SANMVadEncoder")
    return d, g


def test_realtime_punc_model_directory_builds(tmp_path):
    from funasr_amd.auto_model import AutoModel
    d, _ = _dir(tmp_path)
    am = AutoModel(model=d, device="cpu")
    assert type(am.model).__name__ == "CTTransformerStreaming" and type(am.model.encoder).__name__ == "SANMVadEncoder"
    assert am.model.with_vad()


@pytest.mark.gpu
def test_automodel_realtime_punctuation_session_equals_reference(cuda, tmp_path):
    """AutoModel(model=<realtime punc dir>).generate(input=piece, cache=session) call after call == the reference class's
    own session on the same weights"""
    from funasr_amd.auto_model import AutoModel
    d, g = _dir(tmp_path)
    am = AutoModel(model=d, device="cuda:0")
    for calls in json.loads(str(g["e2e"])):
        cache = {}
        for c in calls:
            r = am.generate(input=c["text"], cache=cache)
            assert len(r) == 1 and r[0]["text"] == c["out"] and r[0]["punc_array"].reshape(-1).tolist() == c["punc_array"]
            assert cache["pre_text"] == c["pre_text"]
