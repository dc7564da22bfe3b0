"""Seeded synthetic checkpoints and audio with the reference's state_dict key names and shapes.

There are no real weights or corpora in the build environment (no network), so parity tests and bench.py run on
random-initialised models of the exact Paraformer-large / SenseVoiceSmall architecture. Scales are chosen
"trained-like": residual stream O(1-5) after 50 blocks, CIF weights summing to ~4 tokens per second
(SURVEY.md section 8d). Key names follow funasr/models/{sanm/encoder.py,paraformer/cif_predictor.py,
paraformer/decoder.py,sense_voice/model.py}; a real model.pt uses the same keys.

Everything is generated with a CPU torch.Generator, so the same seed gives the same tensors on every machine.
"""
from __future__ import annotations

import math
from typing import Dict

import torch

PARAFORMER_LARGE = dict(
    frontend=dict(fs=16000, window="hamming", n_mels=80, frame_length=25, frame_shift=10, lfr_m=7, lfr_n=6,
                  dither=0.0),
    encoder=dict(input_size=560, output_size=512, attention_heads=4, linear_units=2048, num_blocks=50,
                 kernel_size=11, sanm_shfit=0),
    predictor=dict(idim=512, threshold=1.0, l_order=1, r_order=1, tail_threshold=0.45, smooth_factor=1.0,
                   noise_threshold=0.0, tail_mask=True),
    decoder=dict(vocab_size=8404, encoder_output_size=512, attention_heads=4, linear_units=2048, num_blocks=16,
                 att_layer_num=16, kernel_size=11, sanm_shfit=0),
)

SENSEVOICE_SMALL = dict(
    frontend=dict(fs=16000, window="hamming", n_mels=80, frame_length=25, frame_shift=10, lfr_m=7, lfr_n=6,
                  dither=0.0),
    encoder=dict(input_size=560, output_size=512, attention_heads=4, linear_units=2048, num_blocks=50, tp_blocks=20,
                 kernel_size=11, sanm_shfit=0),
    vocab_size=25055,
)


def tiny(cfg: dict, enc_blocks: int = 2, dec_blocks: int = 2, vocab: int = 97, tp_blocks: int | None = None) -> dict:
    """Same architecture with fewer blocks / smaller vocabulary, for tests the CPU oracle finishes in seconds."""
    import copy

    c = copy.deepcopy(cfg)
    c["encoder"]["num_blocks"] = enc_blocks
    if "tp_blocks" in c["encoder"] and tp_blocks is not None:
        c["encoder"]["tp_blocks"] = tp_blocks
    if "decoder" in c:
        c["decoder"]["num_blocks"] = dec_blocks
        c["decoder"]["att_layer_num"] = dec_blocks
        c["decoder"]["vocab_size"] = vocab
    if "vocab_size" in c:
        c["vocab_size"] = vocab
    return c


class _Rng:
    def __init__(self, seed: int):
        self.g = torch.Generator(device="cpu").manual_seed(seed)

    def normal(self, *shape, std=1.0, mean=0.0):
        return torch.randn(*shape, generator=self.g, dtype=torch.float32) * std + mean


def _linear(sd, rng, name, out_f, in_f, gain=1.0, bias=True, bias_std=0.02):
    sd[name + ".weight"] = rng.normal(out_f, in_f, std=gain / math.sqrt(in_f))
    if bias:
        sd[name + ".bias"] = rng.normal(out_f, std=bias_std)


def _ln(sd, rng, name, dim):
    sd[name + ".weight"] = rng.normal(dim, std=0.1, mean=1.0)
    sd[name + ".bias"] = rng.normal(dim, std=0.05)


def encoder_state_dict(cfg: dict, seed: int = 0, prefix: str = "") -> Dict[str, torch.Tensor]:
    """Keys of SANMEncoder (sanm/encoder.py:351-377) / SenseVoiceEncoderSmall (sense_voice/model.py:575-621)."""
    rng = _Rng(seed)
    sd: Dict[str, torch.Tensor] = {}
    D, F, Din, K = cfg["output_size"], cfg["linear_units"], cfg["input_size"], cfg["kernel_size"]
    blocks = [("encoders0.0", Din)] + [(f"encoders.{i}", D) for i in range(cfg["num_blocks"] - 1)]
    blocks += [(f"tp_encoders.{i}", D) for i in range(cfg.get("tp_blocks", 0))]
    for name, in_dim in blocks:
        p = prefix + name
        _ln(sd, rng, p + ".norm1", in_dim)
        _linear(sd, rng, p + ".self_attn.linear_q_k_v", 3 * D, in_dim, gain=1.0)
        sd[p + ".self_attn.fsmn_block.weight"] = rng.normal(D, 1, K, std=0.15)
        _linear(sd, rng, p + ".self_attn.linear_out", D, D, gain=0.5)
        _ln(sd, rng, p + ".norm2", D)
        _linear(sd, rng, p + ".feed_forward.w_1", F, D, gain=1.0)
        _linear(sd, rng, p + ".feed_forward.w_2", D, F, gain=0.5)
    _ln(sd, rng, prefix + "after_norm", D)
    if cfg.get("tp_blocks", 0) > 0:
        _ln(sd, rng, prefix + "tp_norm", D)
    return sd


def predictor_state_dict(cfg: dict, seed: int = 1, prefix: str = "", cif_bias: float = -1.5) -> Dict[str, torch.Tensor]:
    """Keys of CifPredictorV2 (paraformer/cif_predictor.py:241-243)."""
    rng = _Rng(seed)
    D = cfg["idim"]
    taps = cfg["l_order"] + cfg["r_order"] + 1
    w_out = rng.normal(1, D, std=2.0 / math.sqrt(D))
    w_out = w_out - w_out.mean()      # no seed-dependent offset: z ~ N(bias, ~1.4^2) for unit-variance hidden
    sd = {
        prefix + "cif_conv1d.weight": rng.normal(D, D, taps, std=1.0 / math.sqrt(D * taps)),
        prefix + "cif_conv1d.bias": rng.normal(D, std=0.02),
        prefix + "cif_output.weight": w_out,
        # -1.5 (the value the golden fixtures were made with) gives ~1.3 tokens/s on speech_like() clips through the
        # 50-block synthetic encoder; bench.py passes -0.1 = mean alpha ~0.24 = ~4 tokens/s (N ~ 120 per 30 s clip,
        # SURVEY.md 8d), measured with tools/calibrate_alpha.py
        prefix + "cif_output.bias": torch.tensor([cif_bias], dtype=torch.float32),
    }
    return sd


def decoder_state_dict(cfg: dict, seed: int = 2, prefix: str = "", with_embed: bool = False) -> Dict[str, torch.Tensor]:
    """Keys of ParaformerSANMDecoder (paraformer/decoder.py:329-392)."""
    rng = _Rng(seed)
    sd: Dict[str, torch.Tensor] = {}
    D, F, K, V = cfg["encoder_output_size"], cfg["linear_units"], cfg["kernel_size"], cfg["vocab_size"]

    def ffn(p):
        _ln(sd, rng, p + ".norm1", D)
        _linear(sd, rng, p + ".feed_forward.w_1", F, D)
        _ln(sd, rng, p + ".feed_forward.norm", F)
        _linear(sd, rng, p + ".feed_forward.w_2", D, F, gain=0.5, bias=False)

    for i in range(cfg["att_layer_num"]):
        p = prefix + f"decoders.{i}"
        ffn(p)
        _ln(sd, rng, p + ".norm2", D)
        sd[p + ".self_attn.fsmn_block.weight"] = rng.normal(D, 1, K, std=0.15)
        _ln(sd, rng, p + ".norm3", D)
        _linear(sd, rng, p + ".src_attn.linear_q", D, D)
        _linear(sd, rng, p + ".src_attn.linear_k_v", 2 * D, D)
        _linear(sd, rng, p + ".src_attn.linear_out", D, D, gain=0.5)
    for i in range(max(cfg.get("num_blocks", cfg["att_layer_num"]) - cfg["att_layer_num"], 0)):   # decoders2 (decoder.py:363-380)
        p = prefix + f"decoders2.{i}"
        ffn(p)
        _ln(sd, rng, p + ".norm2", D)
        sd[p + ".self_attn.fsmn_block.weight"] = rng.normal(D, 1, K, std=0.15)
    ffn(prefix + "decoders3.0")
    _ln(sd, rng, prefix + "after_norm", D)
    _linear(sd, rng, prefix + "output_layer", V, D)
    if with_embed:
        sd[prefix + "embed.0.weight"] = rng.normal(V, D, std=0.02)   # training-only (decoder.py:314-317)
    return sd


BENCH_CIF_BIAS = -0.1


def paraformer_state_dict(cfg: dict = PARAFORMER_LARGE, seed: int = 0, cif_bias: float = -1.5) -> Dict[str, torch.Tensor]:
    sd = {}
    sd.update(encoder_state_dict(cfg["encoder"], seed * 3 + 0, "encoder."))
    sd.update(predictor_state_dict(cfg["predictor"], seed * 3 + 1, "predictor.", cif_bias=cif_bias))
    sd.update(decoder_state_dict(cfg["decoder"], seed * 3 + 2, "decoder."))
    return sd


def sensevoice_state_dict(cfg: dict = SENSEVOICE_SMALL, seed: int = 0) -> Dict[str, torch.Tensor]:
    """Keys of SenseVoiceSmall (sense_voice/model.py:702-737): encoder.*, embed.weight, ctc.ctc_lo.*"""
    sd = {}
    sd.update(encoder_state_dict(cfg["encoder"], seed * 3 + 0, "encoder."))
    rng = _Rng(seed * 3 + 1)
    D = cfg["encoder"]["output_size"]
    sd["embed.weight"] = rng.normal(16, cfg["encoder"]["input_size"], std=0.5)
    _linear(sd, rng, "ctc.ctc_lo", cfg["vocab_size"], D)
    return sd


def synthetic_cmvn(dim: int = 560, seed: int = 7):
    """Stand-in for am.mvn rows (<AddShift>, <Rescale>): shift ~ -mean(log-mel), scale ~ 1/std."""
    rng = _Rng(seed)
    shift = -(8.0 + rng.normal(dim, std=1.0))
    scale = 0.2 + 0.05 * torch.rand(dim, generator=rng.g)
    return shift.float(), scale.float()


def speech_like(n_samples: int, seed: int, fs: int = 16000) -> torch.Tensor:
    """Deterministic speech-like clip: harmonics of a slowly varying f0, syllable-rate amplitude
    modulation, a little noise, peak 0.3. float32 in [-1, 1]."""
    g = torch.Generator(device="cpu").manual_seed(1234 + seed)
    t = torch.arange(n_samples, dtype=torch.float64) / fs
    r = torch.rand(8, generator=g, dtype=torch.float64)
    f0 = 80.0 + 220.0 * r[0] + 20.0 * torch.sin(2 * math.pi * (0.3 + r[1]) * t)
    phase = 2 * math.pi * torch.cumsum(f0, 0) / fs
    nh = 3 + int(r[2] * 3)
    x = torch.zeros(n_samples, dtype=torch.float64)
    for h in range(1, nh + 1):
        x += torch.sin(h * phase + 6.28 * r[3] * h) / h
    am = 0.55 + 0.45 * torch.sin(2 * math.pi * (3.0 + 3.0 * r[4]) * t + 6.28 * r[5])
    x = x * am
    x = x + 0.01 * torch.randn(n_samples, generator=g, dtype=torch.float64)
    x = x / x.abs().max() * 0.3
    return x.float()


def confident_output_layer(weight: torch.Tensor, bias: torch.Tensor, hidden: torch.Tensor, margin: float = 4.0,
                           merge_cos: float = 0.98, chunk: int = 2048):
    """A "trained-like" output layer from a random one. A random Linear over 8404 classes puts the top two logits of a
    position ~0.24 sigma apart on average with a FLAT density of gaps near zero: about one position in 10^4 is a near-tie
    that any fp32 summation order flips, so id-level parity on random weights cannot tell a benign near-tie from a small
    systematic error. A trained model is confident on in-distribution speech. This is the closed-form stand-in for that
    training, on the hidden states `hidden` [n, D] the layer actually sees (the decoder's after_norm output at the token
    positions of a calibration batch; the encoder output at valid frames for a CTC head), in ONE step (no iteration):
      * position k is labelled with the class the random layer prefers for its CENTRED hidden state hc_k = h_k - mean (as
        diverse a label sequence as before); positions whose centred states nearly coincide (cosine > merge_cos) take the
        label of the first of them;
      * W_c += M hc_k / |hc_k|^2 and b_c -= M hc_k . mean / |hc_k|^2 for every position k labelled c: adds exactly M to
        position k's own logit of class c and M cos(hc_k, hc_j) |hc_j| / |hc_k| at any other position j. With
        M = margin / (1 - merge_cos) a position's own class leads every class labelled at a non-merged position by >= ~margin.
    Pure torch, runs on the tensors' device. Returns (weight, bias, stats); stats: min / median top-2 gap, the share of
    positions with a top-2 gap above 1e-3, distinct classes, merged positions."""
    W, b = weight.detach().clone().float(), bias.detach().clone().float()
    H = hidden.detach().float()
    n = H.shape[0]
    mean = H.mean(0)
    Hc = H - mean
    nrm2 = (Hc * Hc).sum(1, keepdim=True).clamp_min(1e-12)
    Hn = Hc / nrm2.sqrt()
    blocks = [slice(r0, min(n, r0 + chunk)) for r0 in range(0, n, chunk)]
    cls = torch.cat([(Hc[r] @ W.T).argmax(1) for r in blocks])
    rep = torch.cat([((Hn[r] @ Hn.T) > merge_cos).float().argmax(1) for r in blocks])     # first near-duplicate (<= own index)
    for _ in range(64):
        nxt = rep[rep]
        if bool(torch.equal(nxt, rep)):
            break
        rep = nxt
    cls = cls[rep]
    M = margin / (1.0 - merge_cos)
    d = Hc / nrm2
    W.index_add_(0, cls, M * d)
    b.index_add_(0, cls, -M * (d @ mean))
    gaps = []
    for r in blocks:
        v = (H[r] @ W.T + b).topk(2, dim=1).values
        gaps.append(v[:, 0] - v[:, 1])
    gaps = torch.cat(gaps)
    ar = torch.arange(n, device=H.device)
    stats = dict(min_gap=float(gaps.min()), median_gap=float(gaps.median()), frac_gap_above_1e3=float((gaps > 1e-3).float().mean()),
                 distinct_classes=int(cls.unique().numel()), merged_positions=int((rep != ar).sum()), positions=int(n), boost=M)
    return W, b, stats


def make_paraformer_confident(model, feats: torch.Tensor, flens, margin: float = 4.0):
    """Calibrate the output layer of a (synthetic) Paraformer mirror on one batch of features with confident_output_layer:
    runs the model's own device path up to the decoder's hidden states at the token positions, rewrites
    `decoder.output_layer.{weight,bias}` in place and returns ({state_dict key: cpu tensor} for the CPU oracle, stats)."""
    enc, olens = model.encode(feats, flens)
    embeds, token_num, _, _ = model.calc_predictor(enc, olens)
    tok = [int(round(v)) for v in token_num.tolist()]
    hid, _ = model.decoder(enc, olens, embeds, torch.tensor(tok), return_hidden=True)
    H = torch.cat([hid[b, : tok[b]] for b in range(len(tok)) if tok[b] > 0])
    lay = model.decoder.output_layer
    W, b, stats = confident_output_layer(lay.weight.to(H.device), lay.bias.to(H.device), H, margin=margin)
    with torch.no_grad():
        lay.weight.copy_(W.to(lay.weight.device))
        lay.bias.copy_(b.to(lay.bias.device))
    model.decoder.mark_dirty()
    return {"decoder.output_layer.weight": W.cpu(), "decoder.output_layer.bias": b.cpu()}, stats
