"""CPU restatement of the CT-Transformer punctuation network (TEST INFRASTRUCTURE; nothing in the product path imports this).

`punc_forward` follows CTTransformer.punc_forward (funasr/models/ct_transformer/model.py:105-124): embedding lookup ->
SANMEncoder (the oracle's restatement, generic in d_model / heads) -> Linear(att_unit, n_punc). Pinned to the reference's own
class by tests/golden/punc.npz (oracle/make_golden_punc.py). `punc_forward_vad` / `sanm_vad_encoder`: the realtime model
(CTTransformerStreaming over SANMVadEncoder), pinned by tests/golden/punc_streaming.npz (oracle/make_golden_punc_streaming.py)."""
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


def sanm_vad_encoder(xs: torch.Tensor, lens: torch.Tensor, vad_indexes, sd: SD, cfg: dict, prefix: str = "encoder.",
                     eps: float = 1e-12) -> torch.Tensor:
    """SANMVadEncoder.forward (funasr/models/ct_transformer_streaming/encoder.py:355-430, input_layer "pe"): the blocks of
    the SAN-M encoder (FSMN masked by the key padding only) with a [B, T, T] attention mask per block -- padding & causal
    everywhere (`no_future_masks`, :372-374), padding & `vad_mask(T, vad_pos)` (transformer/utils/mask.py:38-52) in the last
    block of `encoders` (:401-415) -- applied as in attention.py:284-306 (-inf before the softmax, 0 after)."""
    import torch.nn.functional as F
    B, T, _ = xs.shape
    D, H = cfg["output_size"], cfg["attention_heads"]
    dk = D // H
    left_pad = (cfg["kernel_size"] - 1) // 2 + max(cfg.get("sanm_shfit", 0), 0)
    key_mask = torch.arange(T)[None, :] < lens[:, None].to(torch.int64)                      # [B, T]
    causal = torch.tril(torch.ones(T, T, dtype=torch.bool))[None] & key_mask[:, None, :]      # [B, T, T]
    corner = key_mask[:, None, :].expand(B, T, T).clone()
    for b in range(B):
        vp = int(vad_indexes[b])
        if 0 < vp < T:
            corner[b, : vp - 1, vp:] = False
    x = xs * D ** 0.5
    x = x + O.sinusoidal_pe(T, xs.shape[-1])[None]
    names = [prefix + "encoders0.0."] + [prefix + f"encoders.{i}." for i in range(cfg["num_blocks"] - 1)]
    for li, p in enumerate(names):
        allow = corner if (li >= 1 and li + 1 == len(names)) else causal
        in_dim = x.shape[-1]
        xn = O._ln(x, sd, p + "norm1", eps)
        qkv = F.linear(xn, sd[p + "self_attn.linear_q_k_v.weight"], sd[p + "self_attn.linear_q_k_v.bias"])
        q, k, v = torch.split(qkv, D, dim=-1)
        mem = O._fsmn(v, sd[p + "self_attn.fsmn_block.weight"], key_mask[:, :, None].float(), left_pad)
        qh = q.reshape(B, T, H, dk).transpose(1, 2) * dk ** (-0.5)
        kh = k.reshape(B, T, H, dk).transpose(1, 2)
        vh = v.reshape(B, T, H, dk).transpose(1, 2)
        ban = ~allow[:, None]
        attn = torch.softmax((qh @ kh.transpose(-2, -1)).masked_fill(ban, float("-inf")), dim=-1).masked_fill(ban, 0.0)
        att = F.linear((attn @ vh).transpose(1, 2).reshape(B, T, D), sd[p + "self_attn.linear_out.weight"],
                       sd[p + "self_attn.linear_out.bias"])
        y = att + mem
        x = x + y if in_dim == D else y
        h = torch.relu(F.linear(O._ln(x, sd, p + "norm2", eps), sd[p + "feed_forward.w_1.weight"], sd[p + "feed_forward.w_1.bias"]))
        x = x + F.linear(h, sd[p + "feed_forward.w_2.weight"], sd[p + "feed_forward.w_2.bias"])
    return O._ln(x, sd, prefix + "after_norm", eps)


def punc_forward_vad(ids: torch.Tensor, lens: torch.Tensor, vad_indexes, sd: SD, enc_cfg: dict) -> torch.Tensor:
    """CTTransformerStreaming.punc_forward (ct_transformer_streaming/model.py:59-72): ids int [B, L] -> logits [B, L, n_punc]"""
    x = sd["embed.weight"][ids.long()]
    return sanm_vad_encoder(x, lens, vad_indexes, sd, enc_cfg) @ sd["decoder.weight"].T + sd["decoder.bias"]


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
