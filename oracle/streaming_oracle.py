"""CPU oracle for the STREAMING Paraformer step (600 ms chunks).  TEST INFRASTRUCTURE ONLY.

Functional restatement, with explicit state dictionaries, of what the reference computes per chunk:
  * WavFrontendOnline.forward                funasr/frontends/wav_frontend.py:507-642 (+ forward_fbank :395-462,
                                             apply_lfr :349-380, forward_lfr_cmvn :464-505)
  * SANMEncoderChunkOpt.forward_chunk        funasr/models/scama/encoder.py:480-549, EncoderLayerSANM.forward_chunk
                                             funasr/models/sanm/encoder.py:150-184,
                                             MultiHeadedAttentionSANM.forward_chunk funasr/models/sanm/attention.py:329-366,
                                             StreamSinusoidalPositionEncoder funasr/models/transformer/embedding.py:435-482
  * CifPredictorV2.forward_chunk             funasr/models/paraformer/cif_predictor.py:316-412
  * ParaformerSANMDecoder.forward_chunk      funasr/models/paraformer/decoder.py:515-582, DecoderLayerSANM.forward_chunk
                                             :190-230, decoder FSMN cache funasr/models/sanm/attention.py:583-631,
                                             MultiHeadedAttentionCrossAtt.forward_chunk :815-844
  * ParaformerStreaming.inference / generate_chunk   funasr/models/paraformer_streaming/model.py:552-763
Pinned by oracle/make_golden_streaming.py, which drives the reference's own ParaformerStreaming / WavFrontendOnline
classes over a seeded clip and stores every per-chunk tensor in tests/golden/streaming.npz
(tests/test_oracle_streaming.py). Only tests/, smoke() and bench legs may import this module.
"""
from __future__ import annotations

import math
from typing import Dict, List, Optional, Tuple

import numpy as np
import torch
import torch.nn.functional as F

from . import paraformer_oracle as O

Tensor = torch.Tensor
SD = Dict[str, Tensor]


# ====================================================================================================== frontend
def frontend_init() -> dict:
    """WavFrontendOnline.init_cache (:644-662), the fields the ASR features depend on."""
    return dict(input_cache=torch.empty(0), lfr_splice_cache=None)


def online_lfr(feats: Tensor, lfr_m: int, lfr_n: int, is_final: bool) -> Tuple[Tensor, Tensor]:
    """WavFrontendOnline.apply_lfr (:349-380). `feats` already starts with the splice cache (left context), so row i
    is frames [n*i, n*i + m). Not final: only rows whose m frames exist are emitted; the frames from row `last_idx`
    on are kept as the next splice cache. Final: ceil((T - (m-1)//2) / n) rows, missing frames = the last frame."""
    T, D = feats.shape
    t_lfr = int(np.ceil((T - (lfr_m - 1) // 2) / lfr_n))
    last_idx = (T - lfr_m) // lfr_n + 1
    rows = t_lfr if is_final else last_idx
    rows = max(rows, 0)
    splice_idx = min(T - 1, rows * lfr_n)
    if rows > 0:
        idx = (torch.arange(rows)[:, None] * lfr_n + torch.arange(lfr_m)[None, :]).clamp(max=T - 1)
        out = feats[idx].reshape(rows, lfr_m * D)
    else:
        out = torch.zeros(0, lfr_m * D)
    return out.to(torch.float32), feats[splice_idx:]


def frontend_step(wav: Tensor, cache: dict, cmvn: Optional[Tensor], is_final: bool, n_mels: int = 80,
                  lfr_m: int = 7, lfr_n: int = 6, fs: int = 16000) -> Tensor:
    """One WavFrontendOnline.forward call (batch 1) for `wav` [n] in [-1, 1]. Returns feats [t, n_mels*lfr_m]
    (t may be 0)."""
    win, hop = int(25 * fs / 1000), int(10 * fs / 1000)
    x = torch.cat((cache["input_cache"], wav.to(torch.float32)))
    n = x.numel()
    frame_num = int((n - win) / hop + 1) if n >= win else 0
    frame_num = frame_num if frame_num >= 1 else 0
    cache["input_cache"] = x[n - (n - frame_num * hop):] if (n - frame_num * hop) > 0 else x[:0]
    # NB the reference slices input[:, -(L - frame_num*hop):]; with a zero remainder that is input[:, -0:] = everything
    if n - frame_num * hop == 0:
        cache["input_cache"] = x
    if frame_num:
        fb = O.kaldi_fbank(x * (1 << 15), n_mels, 25.0, 10.0, float(fs))
        assert fb.shape[0] == frame_num
        if cache["lfr_splice_cache"] is None:
            cache["lfr_splice_cache"] = fb[0:1].repeat((lfr_m - 1) // 2, 1)
        if fb.shape[0] + cache["lfr_splice_cache"].shape[0] >= lfr_m:
            feats = torch.cat((cache["lfr_splice_cache"], fb), 0)
            out, cache["lfr_splice_cache"] = online_lfr(feats, lfr_m, lfr_n, is_final)
        else:
            cache["lfr_splice_cache"] = torch.cat((cache["lfr_splice_cache"], fb), 0)
            return torch.zeros(0, n_mels * lfr_m)
    else:
        if not is_final or cache["lfr_splice_cache"] is None:
            return torch.zeros(0, n_mels * lfr_m)
        out, cache["lfr_splice_cache"] = online_lfr(cache["lfr_splice_cache"], lfr_m, lfr_n, True)
    if cmvn is not None and out.shape[0] > 0:
        out = O.apply_cmvn(out, cmvn)
    return out


# ======================================================================================================= encoder
def model_init(cfg: dict, chunk_size=(0, 10, 5), enc_look_back: int = 4, dec_look_back: int = 1) -> dict:
    """ParaformerStreaming.init_cache (paraformer_streaming/model.py:511-550)."""
    D = cfg["encoder"]["output_size"]
    Din = cfg["encoder"]["input_size"]
    n_dec = cfg["decoder"]["att_layer_num"]
    n_enc = cfg["encoder"]["num_blocks"]
    return dict(
        chunk_size=list(chunk_size), enc_look_back=enc_look_back, dec_look_back=dec_look_back,
        start_idx=0, feats=torch.zeros(1, chunk_size[0] + chunk_size[2], Din), tail_chunk=False,
        enc_kv=[None] * n_enc, cif_hidden=torch.zeros(1, 1, D), cif_alphas=torch.zeros(1, 1),
        dec_fsmn=[None] * n_dec, dec_kv=[None] * n_dec, frontend=frontend_init(), prev_samples=torch.empty(0))


def _attend(q: Tensor, k: Tensor, v: Tensor, n_heads: int) -> Tensor:
    """unmasked scaled dot-product attention (forward_attention with mask None, attention.py:270-306)."""
    B, Tq, D = q.shape
    return O._mha(q, k, v, torch.ones(B, k.shape[1], dtype=torch.bool), n_heads)


def encoder_chunk(xs: Tensor, st: dict, sd: SD, cfg: dict, prefix: str = "encoder.", eps: float = 1e-12) -> Tensor:
    """SANMEncoderChunkOpt.forward_chunk. xs [1, n, Din] un-scaled online features (for a tail chunk the caller
    passes st["feats"] itself, exactly like ParaformerStreaming.inference :715-720). Returns [1, W, D]."""
    D, H = cfg["output_size"], cfg["attention_heads"]
    cs = st["chunk_size"]
    xs = xs * D ** 0.5                      # in the reference this is IN PLACE on the caller's tensor (:496) ...
    n = xs.shape[1]
    pe = O.sinusoidal_pe(n + st["start_idx"], xs.shape[-1])[st["start_idx"]:st["start_idx"] + n]
    st["start_idx"] += n
    if st["tail_chunk"]:
        st["feats"] = xs                    # ... so for the tail chunk the cached window ends up scaled a second
        x = xs                              # time and is used WITHOUT the new position encoding (:501-502)
    else:
        x = torch.cat((st["feats"], xs + pe[None]), dim=1)
        st["feats"] = x[:, -(cs[0] + cs[2]):, :]
    names = [prefix + "encoders0.0."] + [prefix + f"encoders.{i}." for i in range(cfg["num_blocks"] - 1)]
    left_pad = (cfg["kernel_size"] - 1) // 2 + max(cfg.get("sanm_shfit", 0), 0)
    ones = torch.ones(1, x.shape[1], 1)
    for li, p in enumerate(names):
        in_dim = x.shape[-1]
        xn = O._ln(x, sd, p + "norm1", eps)
        qkv = F.linear(xn, sd[p + "self_attn.linear_q_k_v.weight"], sd[p + "self_attn.linear_q_k_v.bias"])
        q, k, v = torch.split(qkv, D, dim=-1)
        k_all, v_all = k, v
        if st["enc_look_back"] > 0 or st["enc_look_back"] == -1:
            # the reference slices k_h[:, :, :-(chunk_size[2])] (sanm/attention.py:345-346,356-357): with chunk_size[2] == 0 that
            # is [:-0] = [:0], an EMPTY stride -- nothing is ever cached and a look-back setting has no effect
            keep = k.shape[1] - cs[2] if cs[2] > 0 else 0
            k_stride, v_stride = k[:, :max(keep, 0)], v[:, :max(keep, 0)]
            if st["enc_kv"][li] is not None:
                ck, cv = st["enc_kv"][li]
                k_all, v_all = torch.cat((ck, k), 1), torch.cat((cv, v), 1)
                ck, cv = torch.cat((ck, k_stride), 1), torch.cat((cv, v_stride), 1)
                if st["enc_look_back"] != -1:
                    m = st["enc_look_back"] * cs[1]
                    ck, cv = ck[:, -m:], cv[:, -m:]
                st["enc_kv"][li] = (ck, cv)
            else:
                st["enc_kv"][li] = (k_stride, v_stride)
        mem = O._fsmn(v, sd[p + "self_attn.fsmn_block.weight"], ones, left_pad)
        att = F.linear(_attend(q, k_all, v_all, H), sd[p + "self_attn.linear_out.weight"],
                       sd[p + "self_attn.linear_out.bias"])
        y = att + mem
        x = x + y if in_dim == D else y
        h = torch.relu(F.linear(O._ln(x, sd, p + "norm2", eps), sd[p + "feed_forward.w_1.weight"],
                                sd[p + "feed_forward.w_1.bias"]))
        x = x + F.linear(h, sd[p + "feed_forward.w_2.weight"], sd[p + "feed_forward.w_2.bias"])
    return O._ln(x, sd, prefix + "after_norm", eps)


# ===================================================================================================== predictor
def predictor_chunk(hidden: Tensor, st: dict, sd: SD, cfg: dict, is_final: bool, prefix: str = "predictor."):
    """CifPredictorV2.forward_chunk: alphas of the window, zeroed outside the chunk's own frames, the un-fired
    remainder of the previous chunk in front, sequential float32 integrate-and-fire. Returns (embeds [1, n, D] or
    None, n, alphas_used [16 or 17])."""
    B, T, D = hidden.shape
    l, r = cfg["l_order"], cfg["r_order"]
    cs = st["chunk_size"]
    ctx = F.pad(hidden.transpose(1, 2), (l, r))
    out = torch.relu(F.conv1d(ctx, sd[prefix + "cif_conv1d.weight"], sd[prefix + "cif_conv1d.bias"])).transpose(1, 2)
    out = F.linear(out, sd[prefix + "cif_output.weight"], sd[prefix + "cif_output.bias"])
    alphas = torch.relu(torch.sigmoid(out) * cfg.get("smooth_factor", 1.0) - cfg.get("noise_threshold", 0.0)).squeeze(-1)
    alphas[:, :cs[0]] = 0.0
    if not is_final:
        alphas[:, cs[0] + cs[1]:] = 0.0
    hid = torch.cat((st["cif_hidden"], hidden), 1)
    alphas = torch.cat((st["cif_alphas"], alphas), 1)
    if is_final:
        hid = torch.cat((hid, torch.zeros(B, 1, D)), 1)
        alphas = torch.cat((alphas, torch.full((B, 1), float(cfg["tail_threshold"]))), 1)
    thr = torch.tensor(float(cfg["threshold"]))
    integrate = torch.tensor(0.0)
    frames = torch.zeros(D)
    fired: List[Tensor] = []
    for t in range(alphas.shape[1]):
        a = alphas[0, t]
        if a + integrate < thr:
            integrate = integrate + a
            frames = frames + a * hid[0, t]
        else:
            frames = frames + (thr - integrate) * hid[0, t]
            fired.append(frames)
            integrate = integrate + a
            integrate = integrate - thr
            frames = integrate * hid[0, t]
    st["cif_alphas"] = integrate.reshape(1, 1)
    st["cif_hidden"] = (frames / integrate if float(integrate) > 0.0 else frames).reshape(1, 1, D)
    if not fired:
        return None, 0, alphas[0]
    return torch.stack(fired)[None], len(fired), alphas[0]


# ======================================================================================================= decoder
def decoder_chunk(memory: Tensor, embeds: Tensor, st: dict, sd: SD, cfg: dict, prefix: str = "decoder.",
                  eps: float = 1e-12) -> Tensor:
    """ParaformerSANMDecoder.forward_chunk: causal FSMN with a carried left context of kernel_size - 1 frames,
    cross-attention over [cached K/V of the previous memory | this chunk's memory]. Returns logits [1, n, V]."""
    D, H = embeds.shape[-1], cfg["attention_heads"]
    K = cfg["kernel_size"]
    cs = st["chunk_size"]
    left = (K - 1) // 2 + max(cfg.get("sanm_shfit", 0), 0)
    x = embeds
    for i in range(cfg["att_layer_num"]):
        p = prefix + f"decoders.{i}."
        t = O._dec_ffn(O._ln(x, sd, p + "norm1", eps), sd, p, eps)
        tn = O._ln(t, sd, p + "norm2", eps).transpose(1, 2)                    # [1, D, n]
        n = tn.shape[2]
        if st["dec_fsmn"][i] is None:
            xin = F.pad(tn, (left, K - 1 - left))
        else:
            xin = torch.cat((st["dec_fsmn"][i][:, :, 1:], tn), dim=2)[:, :, -(K + n - 1):]
        st["dec_fsmn"][i] = xin
        y = F.conv1d(xin, sd[p + "self_attn.fsmn_block.weight"], groups=D)
        x = x + (y + tn).transpose(1, 2)
        xn = O._ln(x, sd, p + "norm3", eps)
        q = F.linear(xn, sd[p + "src_attn.linear_q.weight"], sd[p + "src_attn.linear_q.bias"])
        kv = F.linear(memory, sd[p + "src_attn.linear_k_v.weight"], sd[p + "src_attn.linear_k_v.bias"])
        k, v = torch.split(kv, D, dim=-1)
        if st["dec_look_back"] > 0:
            m = st["dec_look_back"] * cs[1]
            if st["dec_kv"][i] is not None:
                k, v = torch.cat((st["dec_kv"][i][0], k), 1), torch.cat((st["dec_kv"][i][1], v), 1)
            st["dec_kv"][i] = (k[:, -m:], v[:, -m:])
        x = x + F.linear(_attend(q, k, v, H), sd[p + "src_attn.linear_out.weight"], sd[p + "src_attn.linear_out.bias"])
    p = prefix + "decoders3.0."
    x = O._dec_ffn(O._ln(x, sd, p + "norm1", eps), sd, p, eps)
    hidden = O._ln(x, sd, prefix + "after_norm", eps)
    return F.linear(hidden, sd[prefix + "output_layer.weight"], sd[prefix + "output_layer.bias"])


# ================================================================================================= session glue
def generate_chunk(feats: Tensor, st: dict, sd: SD, cfg: dict, is_final: bool, trace: Optional[list] = None) -> List[int]:
    """ParaformerStreaming.generate_chunk (:552-648), greedy: encode_chunk -> predictor chunk -> (if any token
    fired) decoder chunk -> argmax -> drop sos/eos/blank. `feats` [1, n, 560]."""
    enc = encoder_chunk(feats, st, sd, cfg["encoder"])
    embeds, n, alphas = predictor_chunk(enc, st, sd, cfg["predictor"], is_final)
    rec = dict(enc=enc, n=n, alphas=alphas)
    ids: List[int] = []
    if n >= 1:
        logits = decoder_chunk(enc, embeds, st, sd, cfg["decoder"])
        raw = torch.log_softmax(logits, dim=-1)[0].argmax(-1).tolist()
        ids = [t for t in raw if t not in (0, 1, 2)]
        rec.update(embeds=embeds, raw_ids=raw)
    if trace is not None:
        trace.append(rec)
    return ids


def streaming_inference(wav: Tensor, st: dict, sd: SD, cfg: dict, cmvn: Optional[Tensor], is_final: bool,
                        trace: Optional[list] = None) -> List[int]:
    """ParaformerStreaming.inference (:650-763) for one call with `wav` [n] (batch 1): chunks of
    chunk_size[1]*960 samples, leftover samples carried in st["prev_samples"]; on the final call a last piece shorter
    than 960 samples re-feeds the cached window ("tail chunk", :715-720)."""
    stride = st["chunk_size"][1] * 960
    audio = torch.cat((st["prev_samples"], wav.to(torch.float32)))
    n = int(len(audio) // stride + int(is_final))
    m = int(len(audio) % stride * (1 - int(is_final)))
    tokens: List[int] = []
    for i in range(n):
        fin = is_final and i == n - 1
        piece = audio[i * stride:(i + 1) * stride]
        if fin and len(piece) < 960:
            st["tail_chunk"] = True
            feats = st["feats"]
        else:
            feats = frontend_step(piece, st["frontend"], cmvn, fin)[None]
        tokens.extend(generate_chunk(feats, st, sd, cfg, fin, trace))
    st["prev_samples"] = audio[-m:] if m > 0 else torch.empty(0)
    return tokens
