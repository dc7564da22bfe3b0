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
