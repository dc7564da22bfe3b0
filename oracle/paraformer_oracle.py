"""CPU oracle for the Paraformer / SenseVoice inference hot path.  TEST INFRASTRUCTURE ONLY.

A plain functional restatement (torch CPU tensors, float32 unless the reference says otherwise) of what the
reference computes on its CPU path, written from the reference's semantics and pinned against the reference
itself: `oracle/make_golden.py` imports the reference's own nn.Modules from /root/reference (and the
reference-vendored kaldi-native-fbank built by oracle/Makefile into oracle/_ref/) and stores their outputs under
tests/golden/; tests/test_oracle.py checks every function here against those fixtures (parity PINNED for the
neural path; the fbank front end is pinned to kaldi-native-fbank within a stated tolerance because
torchaudio -- the reference's actual fbank -- is a third-party dependency that is absent here).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module. The product path
(funasr_amd/) never does and fails loudly without the HIP extension.

Each function cites the reference lines it restates (paths relative to /root/reference).
"""
from __future__ import annotations

import math
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np
import torch
import torch.nn.functional as F

Tensor = torch.Tensor
SD = Dict[str, Tensor]


# ====================================================================================================== frontend
def _next_pow2(n: int) -> int:
    return 1 if n == 0 else 2 ** (n - 1).bit_length()


def kaldi_mel_banks(num_bins: int, padded_window: int, sample_freq: float, low_freq: float = 20.0,
                    high_freq: float = 0.0) -> Tensor:
    """Triangular mel filters [num_bins, padded_window // 2], Kaldi semantics as evaluated by
    torchaudio.compliance.kaldi.get_mel_banks (float32 tensor math, python-float scalars); same rule as
    runtime/onnxruntime/third_party/kaldi-native-fbank/kaldi-native-fbank/csrc/mel-computations.cc:118-210."""
    num_fft_bins = padded_window // 2
    nyquist = 0.5 * sample_freq
    if high_freq <= 0.0:
        high_freq += nyquist
    fft_bin_width = sample_freq / padded_window
    mel_low = 1127.0 * math.log(1.0 + low_freq / 700.0)
    mel_high = 1127.0 * math.log(1.0 + high_freq / 700.0)
    delta = (mel_high - mel_low) / (num_bins + 1)
    b = torch.arange(num_bins, dtype=torch.float32).unsqueeze(1)
    left = mel_low + b * delta
    center = mel_low + (b + 1.0) * delta
    right = mel_low + (b + 2.0) * delta
    mel = (1127.0 * (1.0 + (fft_bin_width * torch.arange(num_fft_bins, dtype=torch.float32)) / 700.0).log()).unsqueeze(0)
    up = (mel - left) / (center - left)
    down = (right - mel) / (right - center)
    return torch.max(torch.zeros(1), torch.min(up, down))


def kaldi_window(window_type: str, window_size: int, blackman_coeff: float = 0.42) -> Tensor:
    """torchaudio.compliance.kaldi._feature_window_function in float32 (hamming | hanning | povey | rectangular | blackman);
    Kaldi's definitions: kaldi-native-fbank/csrc/feature-window.cc:25-55."""
    if window_type == "hamming":
        return torch.hamming_window(window_size, periodic=False, alpha=0.54, beta=0.46, dtype=torch.float32)
    if window_type == "hanning":
        return torch.hann_window(window_size, periodic=False, dtype=torch.float32)
    if window_type == "povey":
        return torch.hann_window(window_size, periodic=False, dtype=torch.float32).pow(0.85)
    if window_type == "rectangular":
        return torch.ones(window_size, dtype=torch.float32)
    if window_type == "blackman":
        a = 2 * math.pi / (window_size - 1)
        i = torch.arange(window_size, dtype=torch.float32)
        return (blackman_coeff - 0.5 * torch.cos(a * i) + (0.5 - blackman_coeff) * torch.cos(2 * a * i)).to(torch.float32)
    raise ValueError("Invalid window type " + window_type)


def kaldi_fbank(waveform: Tensor, num_mel_bins: int = 80, frame_length_ms: float = 25.0, frame_shift_ms: float = 10.0,
                sample_frequency: float = 16000.0, preemphasis: float = 0.97, low_freq: float = 20.0,
                high_freq: float = 0.0, dither: float = 0.0, generator=None, window_type: str = "hamming",
                snip_edges: bool = True) -> Tensor:
    """Kaldi `compute-fbank-feats` for a [n] float32 waveform (already scaled by 32768), options as WavFrontend passes
    them (funasr/frontends/wav_frontend.py:171-181): dither, window_type and snip_edges from the constructor,
    remove_dc_offset, power spectrum, log, energy_floor 0, no energy column. Evaluated the way
    torchaudio.compliance.kaldi.fbank does it (float32 torch ops); Kaldi semantics per
    kaldi-native-fbank/csrc/feature-window.cc:66-90,152-171,186-244 and feature-fbank.cc:75-106.
    Returns [T_fb, num_mel_bins] float32."""
    wave = waveform.to(torch.float32)
    window_shift = int(sample_frequency * frame_shift_ms * 0.001)
    # (frame_length_ms may be the 0-dim float32 tensor of short_clip_frame_length_ms: the product is then float32 tensor arithmetic,
    #  as in torchaudio's _get_waveform_and_window_properties when WavFrontend hands it a tensor)
    window_size = int(sample_frequency * frame_length_ms * 0.001)
    padded = _next_pow2(window_size)
    n = wave.numel()
    if snip_edges:
        if n < window_size:
            return torch.zeros(0, num_mel_bins)
        m = 1 + (n - window_size) // window_shift                  # feature-window.cc:76-86
        frames = wave.as_strided((m, window_size), (window_shift, 1)).clone()
    else:
        # (n + shift / 2) / shift frames, frame f starts at f * shift + shift / 2 - window / 2 (feature-window.cc:55-64,87-89);
        # samples outside [0, n) are mirrored: s < 0 -> -s - 1, s >= n -> 2 n - 1 - s (feature-window.cc:152-171)
        m = (n + window_shift // 2) // window_shift
        if m == 0:
            return torch.zeros(0, num_mel_bins)
        idx = (torch.arange(m)[:, None] * window_shift + (window_shift // 2 - window_size // 2) + torch.arange(window_size)[None, :])
        while bool(((idx < 0) | (idx >= n)).any()):
            idx = torch.where(idx < 0, -idx - 1, idx)
            idx = torch.where(idx >= n, 2 * n - 1 - idx, idx)
        frames = wave[idx]
    if dither != 0.0:
        # torchaudio.compliance.kaldi._get_window: strided_input + randn(strided_input.shape) * dither -- one draw per
        # (frame, sample), before the DC removal (the reference default, wav_frontend.py:106; statistical parity only)
        frames = frames + torch.randn(frames.shape, generator=generator) * dither
    frames = frames - frames.mean(dim=1, keepdim=True)              # remove_dc_offset (:186-196)
    prev = torch.cat([frames[:, :1], frames[:, :-1]], dim=1)        # x[-1] := x[0]  (:204-215)
    frames = frames - preemphasis * prev
    window = kaldi_window(window_type, window_size)
    frames = frames * window.unsqueeze(0)
    if padded != window_size:
        frames = F.pad(frames, (0, padded - window_size))
    spectrum = torch.fft.rfft(frames).abs().pow(2.0)                # power spectrum [m, padded/2+1]
    mel = kaldi_mel_banks(num_mel_bins, padded, sample_frequency, low_freq, high_freq)
    mel = F.pad(mel, (0, 1))                                        # Nyquist bin carries zero weight
    e = spectrum @ mel.T
    return torch.max(e, torch.tensor(torch.finfo(torch.float32).eps)).log()   # feature-fbank.cc:102-106


def short_clip_frame_length_ms(n_samples: int, fs: int = 16000, frame_length: int = 25):
    """WavFrontend.forward's per-utterance window, funasr/frontends/wav_frontend.py:176:
    `frame_length=min(self.frame_length, waveform_length / self.fs * 1000)` with waveform_length = input_lengths[i], a 0-dim integer
    TENSOR (extract_fbank hands tensors): the quotient and product are float32 tensor arithmetic, and `min` keeps the int 25 unless
    the tensor is smaller. A clip shorter than 25 ms is analysed with ONE window of (about) its own length."""
    return min(frame_length, torch.tensor(int(n_samples)) / fs * 1000)


def short_clip_window_size(n_samples: int, fs: int = 16000, frame_length: int = 25) -> int:
    """samples of the analysis window for a clip of n_samples (torchaudio: int(sample_frequency * frame_length * 0.001))"""
    return int(float(fs) * short_clip_frame_length_ms(n_samples, fs, frame_length) * 0.001)


def fbank_tables(num_mel_bins: int = 80, window_size: int = 400, sample_frequency: float = 16000.0,
                 low_freq: float = 20.0, high_freq: float = 0.0) -> Tuple[Tensor, Tensor]:
    """(window [window_size], dense mel [num_mel_bins, padded/2+1]) exactly as kaldi_fbank uses them."""
    padded = _next_pow2(window_size)
    window = torch.hamming_window(window_size, periodic=False, alpha=0.54, beta=0.46, dtype=torch.float32)
    mel = F.pad(kaldi_mel_banks(num_mel_bins, padded, sample_frequency, low_freq, high_freq), (0, 1))
    return window, mel.contiguous()


def apply_lfr(feats: Tensor, lfr_m: int, lfr_n: int) -> Tensor:
    """Low-frame-rate stacking, funasr/frontends/wav_frontend.py:63-86:
    out[i, 80j:80j+80] = feats[clamp(n*i + j - (m-1)//2, 0, T-1)], i < ceil(T / n)."""
    T, D = feats.shape
    T_out = (T + lfr_n - 1) // lfr_n
    idx = (torch.arange(T_out)[:, None] * lfr_n + torch.arange(lfr_m)[None, :] - (lfr_m - 1) // 2).clamp(0, T - 1)
    return feats[idx].reshape(T_out, lfr_m * D).to(torch.float32)


def load_cmvn(path: str) -> Tensor:
    """Kaldi-nnet am.mvn parser, funasr/frontends/wav_frontend.py:15-43 -> [2, dim] (shift row, scale row)."""
    with open(path, "r", encoding="utf-8") as f:
        lines = f.readlines()
    shift: List[str] = []
    scale: List[str] = []
    for i, line in enumerate(lines):
        item = line.split()
        if not item:
            continue
        if item[0] in ("<AddShift>", "<Rescale>"):
            nxt = lines[i + 1].split()
            if nxt[0] == "<LearnRateCoef>":
                vals = nxt[3:len(nxt) - 1]
                if item[0] == "<AddShift>":
                    shift = vals
                else:
                    scale = vals
    return torch.tensor(np.array([np.array(shift).astype(np.float32), np.array(scale).astype(np.float32)]))


def apply_cmvn(feats: Tensor, cmvn: Tensor) -> Tensor:
    """(x + shift) * scale, funasr/frontends/wav_frontend.py:46-60."""
    d = feats.shape[1]
    return ((feats + cmvn[0:1, :d]) * cmvn[1:2, :d]).to(torch.float32)


def utterance_mvn(x: Tensor, ilens: Optional[Tensor] = None, norm_means: bool = True, norm_vars: bool = False,
                  eps: float = 1.0e-20) -> Tensor:
    """UtteranceMVN at inference, funasr/models/normalize/utterance_mvn.py:51-96, step by step -- including that with norm_means
    the padded rows come out as -mean and (with norm_vars too) take part in the variance, whose divisor is sqrt(std) (:85-87).
    x [B, T, D] -> a new tensor (the reference works in place)."""
    x = x.clone()
    B, T = x.shape[:2]
    if ilens is None:
        ilens = torch.full((B,), T)
    lens_f = ilens.to(x.dtype).view(-1, 1, 1)
    pad = torch.arange(T)[None, :, None] >= ilens.to(torch.int64).view(-1, 1, 1)
    x = x.masked_fill(pad, 0.0)
    mean = x.sum(dim=1, keepdim=True) / lens_f
    if norm_means:
        x = x - mean
        if norm_vars:
            var = x.pow(2).sum(dim=1, keepdim=True) / lens_f
            std = torch.clamp(var.sqrt(), min=eps)
            x = x / std.sqrt()
        return x
    if norm_vars:
        y = (x - mean).masked_fill(pad, 0.0)
        var = y.pow(2).sum(dim=1, keepdim=True) / lens_f
        x = x / torch.clamp(var.sqrt(), min=eps)
    return x


def global_mvn(x: Tensor, ilens: Optional[Tensor], mean: Tensor, std: Tensor, norm_means: bool = True,
               norm_vars: bool = True) -> Tensor:
    """GlobalMVN.forward, funasr/models/normalize/global_mvn.py:66-92: x - mean, padded rows zero, / std (mean / std in x's dtype)."""
    x = x.clone()
    B, T = x.shape[:2]
    if ilens is None:
        ilens = torch.full((B,), T)
    pad = torch.arange(T)[None, :, None] >= ilens.to(torch.int64).view(-1, 1, 1)
    if norm_means:
        x = x - mean.to(x.dtype)
    x = x.masked_fill(pad, 0.0)
    if norm_vars:
        x = x / std.to(x.dtype)
    return x


def wav_frontend(waves: Sequence[Tensor], cmvn: Optional[Tensor], n_mels: int = 80, frame_length: int = 25,
                 frame_shift: int = 10, lfr_m: int = 7, lfr_n: int = 6, fs: int = 16000,
                 return_fbank: bool = False, window_type: str = "hamming", snip_edges: bool = True):
    """WavFrontend.forward (funasr/frontends/wav_frontend.py:149-196) with dither = 0: per utterance
    wave * 2^15 -> fbank -> LFR -> CMVN, then zero padding to the longest utterance.
    Returns (feats [B, T, n_mels*lfr_m], lens int32 [B])."""
    feats, lens, fbs = [], [], []
    for w in waves:
        w = w.to(torch.float32) * (1 << 15)
        ms = short_clip_frame_length_ms(w.numel(), fs, frame_length)     # wav_frontend.py:176
        mat = kaldi_fbank(w, n_mels, ms, frame_shift, float(fs), window_type=window_type, snip_edges=snip_edges)
        fbs.append(mat)
        if lfr_m != 1 or lfr_n != 1:
            mat = apply_lfr(mat, lfr_m, lfr_n)
        if cmvn is not None:
            mat = apply_cmvn(mat, cmvn)
        feats.append(mat)
        lens.append(mat.shape[0])
    out = torch.nn.utils.rnn.pad_sequence(feats, batch_first=True, padding_value=0.0)
    lens_t = torch.tensor(lens, dtype=torch.int32)
    if return_fbank:
        return out, lens_t, fbs
    return out, lens_t


# ======================================================================================================= encoder
def sinusoidal_pe(timesteps: int, depth: int, start: int = 0) -> Tensor:
    """SinusoidalPositionEncoder.encode, funasr/models/transformer/embedding.py:396-432: positions are 1-based,
    sin half then cos half (not interleaved), all in float32. Returns [timesteps, depth]."""
    positions = torch.arange(1 + start, timesteps + start + 1)[None, :].type(torch.float32)
    inc = torch.log(torch.tensor([10000], dtype=torch.float32)) / (depth / 2 - 1)
    inv = torch.exp(torch.arange(depth / 2).type(torch.float32) * (-inc))
    scaled = positions.reshape(1, -1, 1) * inv.reshape(1, 1, -1)
    return torch.cat([torch.sin(scaled), torch.cos(scaled)], dim=2)[0].type(torch.float32)


def _ln(x: Tensor, sd: SD, name: str, eps: float) -> Tensor:
    """LayerNorm over the last dim, funasr/models/transformer/layer_norm.py:13-38."""
    return F.layer_norm(x, (x.shape[-1],), sd[name + ".weight"], sd[name + ".bias"], eps)


def _fsmn(x: Tensor, w: Tensor, mask: Tensor, left_pad: int) -> Tensor:
    """FSMN memory block, funasr/models/sanm/attention.py:216-239 / :583-631: depthwise conv over time with
    zero padding (left_pad, K-1-left_pad) on the masked input, plus the masked input, masked again.
    x [B, T, D], w [D, 1, K], mask [B, T, 1] float."""
    K = w.shape[-1]
    xin = x * mask
    y = F.conv1d(F.pad(xin.transpose(1, 2), (left_pad, K - 1 - left_pad)), w, groups=w.shape[0]).transpose(1, 2)
    return (y + xin) * mask


def _mha(q: Tensor, k: Tensor, v: Tensor, key_mask: Tensor, n_heads: int) -> Tensor:
    """Scaled dot-product attention with key padding, funasr/models/sanm/attention.py:270-306,324-326:
    q is scaled by d_k^-0.5 BEFORE the product, masked scores are -inf before softmax and 0 after.
    q [B, Tq, D], k/v [B, Tk, D], key_mask bool [B, Tk] (True = valid) -> [B, Tq, D] (heads merged)."""
    B, Tq, D = q.shape
    Tk = k.shape[1]
    dk = D // n_heads
    qh = q.reshape(B, Tq, n_heads, dk).transpose(1, 2) * dk ** (-0.5)
    kh = k.reshape(B, Tk, n_heads, dk).transpose(1, 2)
    vh = v.reshape(B, Tk, n_heads, dk).transpose(1, 2)
    scores = torch.matmul(qh, kh.transpose(-2, -1))
    pad = ~key_mask[:, None, None, :]
    attn = torch.softmax(scores.masked_fill(pad, float("-inf")), dim=-1).masked_fill(pad, 0.0)
    return torch.matmul(attn, vh).transpose(1, 2).contiguous().view(B, Tq, D)


def sanm_encoder_block(x: Tensor, sd: SD, p: str, key_mask: Tensor, n_heads: int, d_model: int, left_pad: int,
                       eps: float) -> Tensor:
    """EncoderLayerSANM.forward, funasr/models/sanm/encoder.py:72-148 (pre-norm, no concat_after):
    x = [x +] (attn(LN1 x) + fsmn(v))  (residual only when in_size == size), x = x + FFN(LN2 x)."""
    in_dim = x.shape[-1]
    xn = _ln(x, sd, p + "norm1", eps)
    qkv = F.linear(xn, sd[p + "self_attn.linear_q_k_v.weight"], sd[p + "self_attn.linear_q_k_v.bias"])
    q, k, v = torch.split(qkv, d_model, dim=-1)
    mem = _fsmn(v, sd[p + "self_attn.fsmn_block.weight"], key_mask[:, :, None].float(), left_pad)
    att = F.linear(_mha(q, k, v, key_mask, n_heads), sd[p + "self_attn.linear_out.weight"],
                   sd[p + "self_attn.linear_out.bias"])
    y = att + mem
    x = x + y if in_dim == d_model else y
    h = torch.relu(F.linear(_ln(x, sd, p + "norm2", eps), sd[p + "feed_forward.w_1.weight"],
                            sd[p + "feed_forward.w_1.bias"]))
    return x + F.linear(h, sd[p + "feed_forward.w_2.weight"], sd[p + "feed_forward.w_2.bias"])


def sanm_encoder(xs: Tensor, lens: Tensor, sd: SD, cfg: dict, prefix: str = "", eps: float = 1e-12,
                 run_blocks: int = -1, collect: Optional[list] = None) -> Tuple[Tensor, Tensor]:
    """SANMEncoder.forward (funasr/models/sanm/encoder.py:392-461, input_layer "pe") and, with cfg["tp_blocks"],
    SenseVoiceEncoderSmall.forward (funasr/models/sense_voice/model.py:623-655, eps 1e-5).
    xs [B, T, input_size] un-scaled features, lens [B]. run_blocks >= 0 returns the raw stream after that many
    blocks. Returns (out [B, T, output_size], olens)."""
    B, T, _ = xs.shape
    D, H = cfg["output_size"], cfg["attention_heads"]
    left_pad = (cfg["kernel_size"] - 1) // 2 + max(cfg.get("sanm_shfit", 0), 0)
    key_mask = torch.arange(T)[None, :] < lens[:, None].to(torch.int64)
    x = xs * D ** 0.5
    x = x + sinusoidal_pe(T, xs.shape[-1])[None]
    names = [prefix + "encoders0.0."] + [prefix + f"encoders.{i}." for i in range(cfg["num_blocks"] - 1)]
    tp = [prefix + f"tp_encoders.{i}." for i in range(cfg.get("tp_blocks", 0))]
    allb = names + tp
    nrun = len(allb) if run_blocks < 0 else min(run_blocks, len(allb))
    for i in range(nrun):
        x = sanm_encoder_block(x, sd, allb[i], key_mask, H, D, left_pad, eps)
        if collect is not None:
            collect.append(x)
        if tp and i + 1 == len(names) and (run_blocks < 0 or nrun > len(names)):
            x = _ln(x, sd, prefix + "after_norm", eps)
    olens = key_mask.sum(1).to(torch.int32)
    if run_blocks >= 0:
        return x, olens
    x = _ln(x, sd, prefix + ("tp_norm" if tp else "after_norm"), eps)
    return x, olens


# ===================================================================================================== predictor
def cif_fires(alphas: Tensor) -> Tuple[Tensor, Tensor]:
    """cif_wo_hidden_v1, funasr/models/paraformer/cif_predictor.py:818-850 (threshold 1.0): float64 prefix sum
    rounded to float32, a frame fires when floor() of the prefix sum increases. Returns (fires, fire_idxs)."""
    ps = torch.cumsum(alphas, dim=1, dtype=torch.float64).to(torch.float32)
    fl = torch.floor(ps)
    prev = torch.floor(torch.roll(ps, 1, dims=1))
    prev[:, 0] = 0
    fire = (fl - prev) > 0
    fires = torch.zeros_like(alphas)
    fires[fire] = 1
    fires = fires + ps - fl
    return fires, fire


def cif_frames(hidden: Tensor, alphas: Tensor) -> Tuple[Tensor, Tensor, Tensor]:
    """cif_v1, funasr/models/paraformer/cif_predictor.py:853-908: token k of an utterance is
    P[f_k] - P[f_{k-1}] + rem_{k-1} h[f_{k-1}] - rem_k h[f_k] with P = cumsum_t(alpha_t h_t) (float32 tensor,
    float64 accumulation inside ATen's CPU cumsum) and rem = frac(fires). Returns (frames [B, Nmax, D] with
    Nmax = max(round(sum alpha)), fires, n_fired [B])."""
    fires, fire = cif_fires(alphas)
    B, T, D = hidden.shape
    n_fired = fire.sum(1)
    n_max = int(torch.round(alphas.sum(-1)).int().max())
    out = torch.zeros(B, n_max, D, dtype=hidden.dtype)
    if int(fire.sum()) == 0:
        return out, fires, n_fired
    P = torch.cumsum(alphas.unsqueeze(-1) * hidden, dim=1)
    rem = fires - torch.floor(fires)
    for b in range(B):
        idx = torch.nonzero(fire[b]).flatten()
        if idx.numel() == 0:
            continue
        fr = P[b, idx]
        rh = rem[b, idx].unsqueeze(-1) * hidden[b, idx]
        sh_fr = torch.cat([torch.zeros(1, D), fr[:-1]], 0)
        sh_rh = torch.cat([torch.zeros(1, D), rh[:-1]], 0)
        tok = fr - sh_fr + sh_rh - rh
        out[b, : tok.shape[0]] = tok[:n_max]
    return out, fires, n_fired


def cif_predictor(hidden: Tensor, lens: Tensor, sd: SD, cfg: dict, prefix: str = ""):
    """CifPredictorV2.forward at inference (target_label None), funasr/models/paraformer/cif_predictor.py:253-314
    + tail_process_fn :414-446. hidden [B, T, D], lens [B].
    Returns (acoustic_embeds [B, N, D], token_num [B] float (floored), alphas [B, T+1], cif_peak [B, T+1])."""
    B, T, D = hidden.shape
    l, r = cfg["l_order"], cfg["r_order"]
    mask = (torch.arange(T)[None, :] < lens[:, None].to(torch.int64)).float()
    ctx = F.pad(hidden.transpose(1, 2), (l, r))
    out = torch.relu(F.conv1d(ctx, sd[prefix + "cif_conv1d.weight"], sd[prefix + "cif_conv1d.bias"])).transpose(1, 2)
    out = F.linear(out, sd[prefix + "cif_output.weight"], sd[prefix + "cif_output.bias"])
    alphas = torch.sigmoid(out)
    alphas = torch.relu(alphas * cfg.get("smooth_factor", 1.0) - cfg.get("noise_threshold", 0.0))
    alphas = (alphas * mask.unsqueeze(-1)).squeeze(-1)
    token_num = alphas.sum(-1)
    tail = cfg.get("tail_threshold", 0.0)
    if tail > 0.0:
        zeros = torch.zeros(B, 1)
        if cfg.get("tail_mask", True):
            tmask = torch.cat([torch.ones(B, 1), mask], 1) - torch.cat([mask, zeros], 1)
            alphas = torch.cat([alphas, zeros], 1) + tmask * tail
        else:
            alphas = torch.cat([alphas, torch.full((B, 1), tail)], 1)
        hidden = torch.cat([hidden, torch.zeros(B, 1, D)], 1)
        token_num = torch.floor(alphas.sum(-1))
    embeds, peaks, _ = cif_frames(hidden, alphas)
    if tail > 0.0:
        embeds = embeds[:, : int(token_num.max().to(torch.int32)), :]
    return embeds, token_num, alphas, peaks


# ======================================================================================================= decoder
def _dec_ffn(x: Tensor, sd: SD, p: str, eps: float) -> Tensor:
    """PositionwiseFeedForwardDecoderSANM, funasr/models/sanm/positionwise_feed_forward.py:12-33."""
    h = torch.relu(F.linear(x, sd[p + "feed_forward.w_1.weight"], sd[p + "feed_forward.w_1.bias"]))
    return F.linear(_ln(h, sd, p + "feed_forward.norm", eps), sd[p + "feed_forward.w_2.weight"])


def paraformer_decoder(memory: Tensor, mem_lens: Tensor, embeds: Tensor, tok_lens: Tensor, sd: SD, cfg: dict,
                       prefix: str = "", eps: float = 1e-12, return_hidden: bool = False):
    """ParaformerSANMDecoder.forward, funasr/models/paraformer/decoder.py:397-449 with DecoderLayerSANM.forward
    :78-121: per block  t = FFN(LN1 x); x = x + FSMN(LN2 t); x = x + CrossAtt(LN3 x, memory); then decoders3
    (FFN only, no residual), after_norm, output_layer. Returns logits [B, N, V] (pre-softmax)."""
    B, N, D = embeds.shape
    T = memory.shape[1]
    H = cfg["attention_heads"]
    left_pad = (cfg["kernel_size"] - 1) // 2 + max(cfg.get("sanm_shfit", 0), 0)
    tgt_mask = (torch.arange(N)[None, :] < tok_lens[:, None].to(torch.int64)).float()[:, :, None]
    mem_mask = torch.arange(T)[None, :] < mem_lens[:, None].to(torch.int64)
    x = embeds
    for i in range(cfg["att_layer_num"]):
        p = prefix + f"decoders.{i}."
        t = _dec_ffn(_ln(x, sd, p + "norm1", eps), sd, p, eps)
        x = x + _fsmn(_ln(t, sd, p + "norm2", eps), sd[p + "self_attn.fsmn_block.weight"], tgt_mask, left_pad)
        xn = _ln(x, sd, p + "norm3", eps)
        q = F.linear(xn, sd[p + "src_attn.linear_q.weight"], sd[p + "src_attn.linear_q.bias"])
        kv = F.linear(memory, sd[p + "src_attn.linear_k_v.weight"], sd[p + "src_attn.linear_k_v.bias"])
        k, v = torch.split(kv, D, dim=-1)
        x = x + F.linear(_mha(q, k, v, mem_mask, H), sd[p + "src_attn.linear_out.weight"],
                         sd[p + "src_attn.linear_out.bias"])
    # decoders2 (decoder.py:363-380, :436-437): num_blocks - att_layer_num blocks with the FSMN built at sanm_shfit 0, no cross-attention
    for i in range(max(cfg.get("num_blocks", cfg["att_layer_num"]) - cfg["att_layer_num"], 0)):
        p = prefix + f"decoders2.{i}."
        t = _dec_ffn(_ln(x, sd, p + "norm1", eps), sd, p, eps)
        x = x + _fsmn(_ln(t, sd, p + "norm2", eps), sd[p + "self_attn.fsmn_block.weight"], tgt_mask, (cfg["kernel_size"] - 1) // 2)
    p = prefix + "decoders3.0."
    x = _dec_ffn(_ln(x, sd, p + "norm1", eps), sd, p, eps)
    hidden = _ln(x, sd, prefix + "after_norm", eps)
    if prefix + "output_layer.weight" not in sd:          # use_output_layer false (SeACo's bias decoder, decoder.py:332-335)
        return hidden
    logits = F.linear(hidden, sd[prefix + "output_layer.weight"], sd[prefix + "output_layer.bias"])
    return (logits, hidden) if return_hidden else logits


def contextual_decoder(memory: Tensor, mem_lens: Tensor, embeds: Tensor, tok_lens: Tensor, contextual_info: Tensor, sd: SD, cfg: dict,
                       clas_scale: float = 1.0, prefix: str = "", eps: float = 1e-12, return_hidden: bool = False):
    """ContextualParaformerDecoder.forward, funasr/models/contextual_paraformer/decoder.py:293-352: att_layer_num - 1 standard blocks
    (`decoders`), then `last_decoder` (ContextualDecoderLayer :53-96: x_self_attn = x + FSMN(LN2 FFN(LN1 x)); x_src_attn = CrossAtt(LN3
    x_self_attn, memory) WITHOUT its residual), the hotword branch cx = CrossAtt_bias(LN3_bias x_self_attn, contextual_info) (ContextualBiasDecoder
    :114-130, every hotword row valid), x = x_self_attn + bias_output([x_src_attn | cx * clas_scale]) (Conv1d(2D, D, 1, bias=False), :335-338),
    decoders2 if any, decoders3, after_norm, output_layer. contextual_info [B or 1, n_hot, D]. Returns logits [B, N, V]."""
    B, N, D = embeds.shape
    T = memory.shape[1]
    H = cfg["attention_heads"]
    left_pad = (cfg["kernel_size"] - 1) // 2 + max(cfg.get("sanm_shfit", 0), 0)
    tgt_mask = (torch.arange(N)[None, :] < tok_lens[:, None].to(torch.int64)).float()[:, :, None]
    mem_mask = torch.arange(T)[None, :] < mem_lens[:, None].to(torch.int64)
    L = cfg["att_layer_num"]

    def cross(xn, mem, mask, p):
        q = F.linear(xn, sd[p + "linear_q.weight"], sd[p + "linear_q.bias"])
        kv = F.linear(mem, sd[p + "linear_k_v.weight"], sd[p + "linear_k_v.bias"])
        k, v = torch.split(kv, D, dim=-1)
        return F.linear(_mha(q, k, v, mask, H), sd[p + "linear_out.weight"], sd[p + "linear_out.bias"])

    x = embeds
    for i in range(L - 1):
        p = prefix + f"decoders.{i}."
        t = _dec_ffn(_ln(x, sd, p + "norm1", eps), sd, p, eps)
        x = x + _fsmn(_ln(t, sd, p + "norm2", eps), sd[p + "self_attn.fsmn_block.weight"], tgt_mask, left_pad)
        x = x + cross(_ln(x, sd, p + "norm3", eps), memory, mem_mask, p + "src_attn.")
    p = prefix + "last_decoder."
    t = _dec_ffn(_ln(x, sd, p + "norm1", eps), sd, p, eps)
    x_self = x + _fsmn(_ln(t, sd, p + "norm2", eps), sd[p + "self_attn.fsmn_block.weight"], tgt_mask, left_pad)
    x_src = cross(_ln(x_self, sd, p + "norm3", eps), memory, mem_mask, p + "src_attn.")
    ctx = contextual_info.expand(B, -1, -1) if contextual_info.shape[0] != B else contextual_info
    hot_mask = torch.ones(B, ctx.shape[1], dtype=torch.bool)
    cx = cross(_ln(x_self, sd, prefix + "bias_decoder.norm3", eps), ctx, hot_mask, prefix + "bias_decoder.src_attn.")
    cat = torch.cat([x_src, cx * clas_scale], dim=2)
    x = x_self + F.conv1d(cat.transpose(1, 2), sd[prefix + "bias_output.weight"]).transpose(1, 2)
    for i in range(max(cfg.get("num_blocks", L) - L, 0)):
        p = prefix + f"decoders2.{i}."
        t = _dec_ffn(_ln(x, sd, p + "norm1", eps), sd, p, eps)
        x = x + _fsmn(_ln(t, sd, p + "norm2", eps), sd[p + "self_attn.fsmn_block.weight"], tgt_mask, (cfg["kernel_size"] - 1) // 2)
    p = prefix + "decoders3.0."
    x = _dec_ffn(_ln(x, sd, p + "norm1", eps), sd, p, eps)
    hidden = _ln(x, sd, prefix + "after_norm", eps)
    logits = F.linear(hidden, sd[prefix + "output_layer.weight"], sd[prefix + "output_layer.bias"])
    return (logits, hidden) if return_hidden else logits


def paraformer_greedy(feats: Tensor, lens: Tensor, sd: SD, cfg: dict, sos: int = 1, eos: int = 2, blank: int = 0):
    """Device half of Paraformer.inference, funasr/models/paraformer/model.py:596-666: encode -> predictor ->
    round().long() -> decoder -> log_softmax -> argmax per valid token -> drop sos/eos/blank.
    Returns dict(enc, alphas, peaks, token_num, embeds, hidden, logits, ids (list of lists), raw_ids)."""
    enc, olens = sanm_encoder(feats, lens, sd, cfg["encoder"], "encoder.")
    embeds, token_num, alphas, peaks = cif_predictor(enc, olens, sd, cfg["predictor"], "predictor.")
    tok = token_num.round().long()
    res = dict(enc=enc, olens=olens, alphas=alphas, peaks=peaks, token_num=tok, embeds=embeds)
    if int(tok.max()) < 1:
        res.update(logits=None, ids=[[] for _ in range(feats.shape[0])], raw_ids=[[] for _ in range(feats.shape[0])])
        return res
    logits, hidden = paraformer_decoder(enc, olens, embeds, tok, sd, cfg["decoder"], "decoder.", return_hidden=True)
    res.update(hidden=hidden)          # the output layer's input (tests re-score it with a calibrated output layer)
    logp = torch.log_softmax(logits, dim=-1)
    raw, ids = [], []
    for b in range(feats.shape[0]):
        y = logp[b, : int(tok[b])].argmax(-1).tolist()
        raw.append(y)
        ids.append([t for t in y if t not in (sos, eos, blank)])
    res.update(logits=logits, ids=ids, raw_ids=raw)
    return res


# ==================================================================================================== SenseVoice
def sensevoice_prepare(feats: Tensor, lens: Tensor, embed_weight: Tensor, language_id: int = 0, textnorm_id: int = 15):
    """Query-frame prefix of SenseVoiceSmall.inference, funasr/models/sense_voice/model.py:971-995: the rows
    embed[language], embed[1], embed[2], embed[textnorm] (withitn 14 / woitn 15) are put IN FRONT of the speech
    features ([lang, event, emo, textnorm, speech...]; lengths + 4)."""
    B = feats.shape[0]
    lang = embed_weight[language_id][None, None, :].repeat(B, 1, 1)
    ev_emo = embed_weight[torch.tensor([1, 2])][None].repeat(B, 1, 1)
    tn = embed_weight[textnorm_id][None, None, :].repeat(B, 1, 1)
    speech = torch.cat((tn, feats), dim=1)
    speech = torch.cat((lang, ev_emo, speech), dim=1)
    return speech, lens + 4


def sensevoice_greedy(feats: Tensor, lens: Tensor, sd: SD, cfg: dict, language_id: int = 0, textnorm_id: int = 15,
                      blank: int = 0):
    """SenseVoiceSmall.inference device half, funasr/models/sense_voice/model.py:987-1028: prefix, encoder
    (eps 1e-5), ctc.log_softmax (funasr/models/ctc/ctc.py:192-203), argmax, unique_consecutive, drop blank."""
    speech, slens = sensevoice_prepare(feats, lens, sd["embed.weight"], language_id, textnorm_id)
    enc, olens = sanm_encoder(speech, slens, sd, cfg["encoder"], "encoder.", eps=1e-5)
    logp = torch.log_softmax(F.linear(enc, sd["ctc.ctc_lo.weight"], sd["ctc.ctc_lo.bias"]), dim=-1)
    ids, frame_ids = [], []
    for b in range(feats.shape[0]):
        y = logp[b, : int(olens[b])].argmax(-1)
        frame_ids.append(y.tolist())
        y = torch.unique_consecutive(y, dim=-1)
        ids.append([int(t) for t in y.tolist() if t != blank])
    return dict(enc=enc, olens=olens, logp=logp, ids=ids, frame_ids=frame_ids)
