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
