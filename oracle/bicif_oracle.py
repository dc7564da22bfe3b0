"""CPU restatement of the timestamp predictor of BiCifParaformer / SeACo-Paraformer.  TEST INFRASTRUCTURE: only tests/,
__graft_entry__.smoke() and bench.py's cpu_baseline leg may import this; the product path never does.

Follows `CifPredictorV3` (funasr/models/bicif_paraformer/cif_predictor.py:121-384) at inference and the calls
`BiCifParaformer.inference` makes around it (funasr/models/bicif_paraformer/model.py:330-418). Pinned to the reference's
own classes by oracle/make_golden_bicif.py -> tests/golden/bicif.npz (tests/test_bicif.py).

Differences from CifPredictorV2 that matter for parity:
  * the integrate-and-fire is the sequential fp32 loop `cif` (:39-86), not the fp64 prefix-sum `cif_v1`;
  * a second head predicts frame weights on a 3x upsampled time axis (ConvTranspose1d(k = stride = 3) -> BLSTM ->
    Linear(2 idim, 1) -> sigmoid -> relu(a * smooth_factor2 - noise_threshold2)), rescaled so that every utterance sums to
    its token count and integrated with `cif_wo_hidden` at threshold 1 - 1e-4 (`get_upsample_timestamp`, :301-352).
"""
from __future__ import annotations

from typing import Dict, Tuple

import torch
import torch.nn.functional as F

Tensor = torch.Tensor
SD = Dict[str, Tensor]


def lstm_direction(x: Tensor, w_ih: Tensor, w_hh: Tensor, b_ih: Tensor, b_hh: Tensor, reverse: bool = False) -> Tensor:
    """one direction of torch.nn.LSTM (batch_first, zero initial state), gate order i, f, g, o. x [B, T, I] -> [B, T, H]"""
    B, T, _ = x.shape
    H = w_hh.shape[1]
    pre = F.linear(x, w_ih, b_ih)
    h = torch.zeros(B, H)
    c = torch.zeros(B, H)
    out = torch.zeros(B, T, H)
    steps = range(T - 1, -1, -1) if reverse else range(T)
    for t in steps:
        g = pre[:, t] + F.linear(h, w_hh, b_hh)
        i, f, gg, o = g.split(H, dim=1)
        c = torch.sigmoid(f) * c + torch.sigmoid(i) * torch.tanh(gg)
        h = torch.sigmoid(o) * torch.tanh(c)
        out[:, t] = h
    return out


def lstm(x: Tensor, sd: SD, prefix: str, layers: int = 1, bidirectional: bool = False) -> Tensor:
    """torch.nn.LSTM(batch_first=True) forward from its state_dict entries (weight_ih_l{k}[_reverse], ...)"""
    for k in range(layers):
        outs = []
        for suffix, rev in (("", False), ("_reverse", True)) if bidirectional else (("", False),):
            outs.append(lstm_direction(x, sd[f"{prefix}weight_ih_l{k}{suffix}"], sd[f"{prefix}weight_hh_l{k}{suffix}"],
                                       sd[f"{prefix}bias_ih_l{k}{suffix}"], sd[f"{prefix}bias_hh_l{k}{suffix}"], rev))
        x = torch.cat(outs, dim=-1)
    return x


def cif_loop(hidden: Tensor, alphas: Tensor, threshold: float = 1.0) -> Tuple[Tensor, Tensor]:
    """`cif` (:39-86): sequential fp32 integrate-and-fire. Returns (frames [B, max round(sum alpha), D], fires [B, T])."""
    B, T, D = hidden.shape
    integrate = torch.zeros(B)
    frame = torch.zeros(B, D)
    fires, frames = [], []
    for t in range(T):
        alpha = alphas[:, t]
        completion = torch.ones(B) - integrate
        integrate = integrate + alpha
        fires.append(integrate)
        fire = integrate >= threshold
        integrate = torch.where(fire, integrate - torch.ones(B), integrate)
        cur = torch.where(fire, completion, alpha)
        remains = alpha - cur
        frame = frame + cur[:, None] * hidden[:, t, :]
        frames.append(frame)
        frame = torch.where(fire[:, None], remains[:, None] * hidden[:, t, :], frame)
    fires_t = torch.stack(fires, 1)
    frames_t = torch.stack(frames, 1)
    n_max = int(torch.round(alphas.sum(-1)).int().max())
    out = torch.zeros(B, n_max, D)
    for b in range(B):
        sel = frames_t[b, fires_t[b] >= threshold]
        out[b, : sel.shape[0]] = sel[:n_max]          # the reference raises when an utterance fires more than n_max times
    return out, fires_t


def cif_wo_hidden(alphas: Tensor, threshold: float) -> Tensor:
    """`cif_wo_hidden` (:89-118): the running integral per frame (before the threshold is taken off), fp32 sequential"""
    B, T = alphas.shape
    integrate = torch.zeros(B)
    fires = []
    thr = torch.ones(B) * threshold
    for t in range(T):
        integrate = integrate + alphas[:, t]
        fires.append(integrate)
        integrate = torch.where(integrate >= threshold, integrate - thr, integrate)
    return torch.stack(fires, 1)


def _conv_relu(hidden: Tensor, sd: SD, cfg: dict, prefix: str) -> Tensor:
    ctx = F.pad(hidden.transpose(1, 2), (cfg["l_order"], cfg["r_order"]))
    return torch.relu(F.conv1d(ctx, sd[prefix + "cif_conv1d.weight"], sd[prefix + "cif_conv1d.bias"]))     # [B, D, T]


def predictor_v3(hidden: Tensor, lens: Tensor, sd: SD, cfg: dict, prefix: str = ""):
    """CifPredictorV3.forward without target labels (:215-299) -> (acoustic_embeds [B, N, D], token_num [B] (floored),
    alphas [B, T+1], cif_peak [B, T+1]). The upsampled head the reference also evaluates here only feeds a training loss."""
    B, T, D = hidden.shape
    mask = (torch.arange(T)[None, :] < lens[:, None].to(torch.int64)).float()
    out = _conv_relu(hidden, sd, cfg, prefix).transpose(1, 2)
    out = F.linear(out, sd[prefix + "cif_output.weight"], sd[prefix + "cif_output.bias"])
    alphas = torch.relu(torch.sigmoid(out) * cfg.get("smooth_factor", 1.0) - cfg.get("noise_threshold", 0.0))
    alphas = (alphas * mask.unsqueeze(-1)).squeeze(-1)
    token_num = alphas.sum(-1)
    tail = cfg.get("tail_threshold", 0.0)
    if tail > 0.0:                                                   # tail_process_fn (:354-383), always with the mask
        zeros = torch.zeros(B, 1)
        tmask = torch.cat([torch.ones(B, 1), mask], 1) - torch.cat([mask, zeros], 1)
        alphas = torch.cat([alphas, zeros], 1) + tmask * tail
        hidden = torch.cat([hidden, torch.zeros(B, 1, D)], 1)
        token_num = torch.floor(alphas.sum(-1))
    embeds, peaks = cif_loop(hidden, alphas, cfg.get("threshold", 1.0))
    if tail > 0.0:
        embeds = embeds[:, : int(token_num.max().to(torch.int32)), :]
    return embeds, token_num, alphas, peaks


def upsample_timestamp(hidden: Tensor, lens: Tensor, token_num: Tensor, sd: SD, cfg: dict, prefix: str = ""):
    """CifPredictorV3.get_upsample_timestamp (:301-352) -> (us_alphas [B, U T], us_cif_peak [B, U T])"""
    B, T, D = hidden.shape
    U = cfg["upsample_times"]
    src = _conv_relu(hidden, sd, cfg, prefix) if cfg.get("use_cif1_cnn", True) else hidden.transpose(1, 2)
    up = F.conv_transpose1d(src, sd[prefix + "upsample_cnn.weight"], sd[prefix + "upsample_cnn.bias"], stride=U).transpose(1, 2)
    kind = cfg.get("upsample_type", "cnn")
    if kind == "cnn_blstm":
        up = lstm(up, sd, prefix + "blstm.", layers=1, bidirectional=True)
    elif kind != "cnn":
        raise NotImplementedError(kind)
    a2 = torch.sigmoid(F.linear(up, sd[prefix + "cif_output2.weight"], sd[prefix + "cif_output2.bias"]))
    a2 = torch.relu(a2 * cfg.get("smooth_factor2", 1.0) - cfg.get("noise_threshold2", 0.0))
    mask2 = (torch.arange(U * T)[None, :] < (U * lens[:, None].to(torch.int64))).float()
    a2 = a2.squeeze(-1) * mask2
    total = a2.sum(-1)
    a2 = a2 * (token_num / total)[:, None]
    return a2, cif_wo_hidden(a2, cfg.get("threshold", 1.0) - 1e-4)


def predictor_v3_state_dict(cfg: dict, seed: int = 1, prefix: str = "", cif_bias: float = -1.5) -> SD:
    """seeded weights with the reference's key names; the second head is biased towards small positive weights so that the
    rescaled integral fires about once per token like a trained model"""
    from funasr_amd import synth
    sd = synth.predictor_state_dict(cfg, seed=seed, prefix=prefix, cif_bias=cif_bias)
    g = torch.Generator().manual_seed(seed + 77)
    D, U = cfg["idim"], cfg["upsample_times"]
    sd[prefix + "upsample_cnn.weight"] = torch.randn(D, D, U, generator=g) * (1.0 / D ** 0.5)
    sd[prefix + "upsample_cnn.bias"] = torch.randn(D, generator=g) * 0.02
    wide = D
    if cfg.get("upsample_type", "cnn") == "cnn_blstm":
        for suffix in ("", "_reverse"):
            sd[prefix + f"blstm.weight_ih_l0{suffix}"] = torch.randn(4 * D, D, generator=g) * (1.0 / D ** 0.5)
            sd[prefix + f"blstm.weight_hh_l0{suffix}"] = torch.randn(4 * D, D, generator=g) * (0.7 / D ** 0.5)
            sd[prefix + f"blstm.bias_ih_l0{suffix}"] = torch.randn(4 * D, generator=g) * 0.1
            sd[prefix + f"blstm.bias_hh_l0{suffix}"] = torch.randn(4 * D, generator=g) * 0.1
        wide = 2 * D
    sd[prefix + "cif_output2.weight"] = torch.randn(1, wide, generator=g) * (2.0 / wide ** 0.5)
    sd[prefix + "cif_output2.bias"] = torch.full((1,), -0.5)
    return sd


def bicif_greedy(feats: Tensor, lens: Tensor, sd: SD, cfg: dict, sos: int = 1, eos: int = 2, blank: int = 0):
    """Device half of BiCifParaformer.inference (model.py:330-372): encode -> CifPredictorV3 -> decoder -> arg-max, plus the
    upsampled weights / peaks `ts_prediction_lfr6_standard` turns into token times."""
    from oracle import paraformer_oracle as O
    enc, olens = O.sanm_encoder(feats, lens, sd, cfg["encoder"], "encoder.")
    embeds, token_num, alphas, peaks = predictor_v3(enc, olens, sd, cfg["predictor"], "predictor.")
    tok = token_num.round().long()
    res = dict(enc=enc, olens=olens, alphas=alphas, peaks=peaks, token_num=tok, embeds=embeds)
    B = feats.shape[0]
    if int(tok.max()) < 1:
        res.update(ids=[[] for _ in range(B)], raw_ids=[[] for _ in range(B)], us_alphas=None, us_peaks=None)
        return res
    logits = O.paraformer_decoder(enc, olens, embeds, tok, sd, cfg["decoder"], "decoder.")
    raw, ids = [], []
    for b in range(B):
        y = logits[b, : int(tok[b])].argmax(-1).tolist()
        raw.append(y)
        ids.append([t for t in y if t not in (sos, eos, blank)])
    usa, usp = upsample_timestamp(enc, olens, tok, sd, cfg["predictor"], "predictor.")
    res.update(logits=logits, ids=ids, raw_ids=raw, us_alphas=usa, us_peaks=usp)
    return res
