"""CPU restatement of the FSMN-VAD network (TEST INFRASTRUCTURE; nothing in the product path imports this).

`fsmn_forward` follows funasr/models/fsmn_vad_streaming/encoder.py:288-378 (FSMN.forward) with its blocks
(:164-217 BasicBlock, :87-161 FSMNBlock): two affine layers + ReLU, then per block a bias-free projection, the
uni-directional memory `out[t] = x[t] + sum_k w[c, k] * z[t + k * lstride]` over z = [left context | x] (depthwise
cross-correlation, zero or cached left context of (lorder - 1) * lstride frames), an affine layer + ReLU; two output
affine layers and a softmax. `frame_decibel` is ComputeDecibel's expression (model.py:513-530) in numpy.
Pinned to the reference's own FSMN class by tests/golden/vad_encoder.npz (oracle/make_golden_vad.py).
"""
from __future__ import annotations

from typing import Dict, Optional

import numpy as np
import torch

SD = Dict[str, torch.Tensor]


def fsmn_forward(feats: torch.Tensor, sd: SD, cfg: dict, cache: Optional[dict] = None, prefix: str = "") -> torch.Tensor:
    """feats [B, T, input_dim] -> posteriors [B, T, output_dim]; `cache` (dict) carries the left context per block."""
    g = lambda n: sd[prefix + n].float()                                            # noqa: E731
    x = feats.float() @ g("in_linear1.linear.weight").T + g("in_linear1.linear.bias")
    x = torch.relu(x @ g("in_linear2.linear.weight").T + g("in_linear2.linear.bias"))
    L, S = cfg["lorder"], cfg["lstride"]
    ctx = (L - 1) * S
    for i in range(cfg["fsmn_layers"]):
        p = f"fsmn.{i}."
        h = x @ g(p + "linear.linear.weight").T                                      # [B, T, proj]
        w = g(p + "fsmn_block.conv_left.weight")[:, 0, :, 0]                         # [proj, L]
        if cache is not None:
            left = cache.get(p, torch.zeros(h.shape[0], ctx, h.shape[2]))
        else:
            left = torch.zeros(h.shape[0], ctx, h.shape[2])
        z = torch.cat([left, h], dim=1)                                              # [B, ctx + T, proj]
        if cache is not None:
            cache[p] = z[:, z.shape[1] - ctx:].clone()
        T = h.shape[1]
        mem = h.clone()
        for k in range(L):
            mem = mem + z[:, k * S: k * S + T] * w[:, k]
        x = torch.relu(mem @ g(p + "affine.linear.weight").T + g(p + "affine.linear.bias"))
    x = x @ g("out_linear1.linear.weight").T + g("out_linear1.linear.bias")
    x = x @ g("out_linear2.linear.weight").T + g("out_linear2.linear.bias")
    return torch.softmax(x, dim=-1)


def frame_decibel(wav: np.ndarray, n_frames: int, frame_len: int = 400, frame_shift: int = 160) -> np.ndarray:
    w = np.asarray(wav, dtype=np.float32)[: (n_frames - 1) * frame_shift + frame_len]
    frames = w[np.arange(0, w.shape[0] - frame_len + 1, frame_shift)[:, None] + np.arange(frame_len)]
    return 10 * np.log10(np.sum(np.square(frames), axis=1) + 0.000001)


def synthetic_state_dict(cfg: dict, seed: int = 0, prefix: str = "") -> SD:
    """seeded weights with the reference's key names and shapes; scaled so that the posteriors are not saturated"""
    g = torch.Generator().manual_seed(seed)
    sd: SD = {}

    def lin(name, o, i, bias=True):
        sd[prefix + name + ".weight"] = torch.randn(o, i, generator=g) / (i ** 0.5)
        if bias:
            sd[prefix + name + ".bias"] = 0.1 * torch.randn(o, generator=g)
    lin("in_linear1.linear", cfg["input_affine_dim"], cfg["input_dim"])
    lin("in_linear2.linear", cfg["linear_dim"], cfg["input_affine_dim"])
    for i in range(cfg["fsmn_layers"]):
        lin(f"fsmn.{i}.linear.linear", cfg["proj_dim"], cfg["linear_dim"], bias=False)
        sd[prefix + f"fsmn.{i}.fsmn_block.conv_left.weight"] = torch.randn(cfg["proj_dim"], 1, cfg["lorder"], 1, generator=g) / cfg["lorder"]
        lin(f"fsmn.{i}.affine.linear", cfg["linear_dim"], cfg["proj_dim"])
    lin("out_linear1.linear", cfg["output_affine_dim"], cfg["linear_dim"])
    lin("out_linear2.linear", cfg["output_dim"], cfg["output_affine_dim"])
    return sd
