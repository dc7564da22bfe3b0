"""CPU restatement of the CT-Transformer punctuation network (TEST INFRASTRUCTURE; nothing in the product path imports this).

`punc_forward` follows CTTransformer.punc_forward (funasr/models/ct_transformer/model.py:105-124): embedding lookup ->
SANMEncoder (the oracle's restatement, generic in d_model / heads) -> Linear(att_unit, n_punc). Pinned to the reference's own
class by tests/golden/punc.npz (oracle/make_golden_punc.py)."""
from __future__ import annotations

from typing import Dict

import torch

from oracle import paraformer_oracle as O

SD = Dict[str, torch.Tensor]
PUNC_LIST = ["<unk>", "_", "，", "。", "？", "、"]


def punc_forward(ids: torch.Tensor, lens: torch.Tensor, sd: SD, enc_cfg: dict) -> torch.Tensor:
    """ids int [B, L], lens [B] -> logits [B, L, n_punc]"""
    x = sd["embed.weight"][ids.long()]
    h, _ = O.sanm_encoder(x, lens, sd, enc_cfg, "encoder.")
    return h @ sd["decoder.weight"].T + sd["decoder.bias"]


def synthetic_state_dict(vocab: int, enc_cfg: dict, n_punc: int = 6, seed: int = 0) -> SD:
    from funasr_amd import synth
    g = torch.Generator().manual_seed(seed)
    sd = synth.encoder_state_dict(enc_cfg, seed=seed + 1, prefix="encoder.")
    sd["embed.weight"] = torch.randn(vocab, enc_cfg["input_size"], generator=g) * 0.5
    sd["decoder.weight"] = torch.randn(n_punc, enc_cfg["output_size"], generator=g) * 0.6
    sd["decoder.bias"] = torch.tensor([-4.0, 1.2, 0.3, 0.0, -0.8, -1.0])[:n_punc]
    return sd


def injected_marks(ids, never_end: bool = False):
    """deterministic stand-in for the network: punc id per position as a function of the ids (both the reference class and
    the HIP-side host logic are driven with it when the sentence assembly is compared)"""
    import numpy as np
    ids = np.asarray(ids, dtype=np.int64)
    L = ids.shape[0]
    h = (ids * 7 + np.arange(L) * 13 + L * 3) % 12
    marks = np.where(h < 7, 1, np.where(h < 9, 2, np.where(h < 10, 3, np.where(h < 11, 4, 5))))
    if never_end:
        marks = np.where((marks == 3) | (marks == 4), 2, marks)
    return marks.astype(np.int64)
