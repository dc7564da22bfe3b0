"""SAN-M encoder on gfx950 (csrc: gemm_f32, attention_f32, rowwise kernels, scheduled by engine_encoder.hip).

Host-side mirrors of
  * `SANMEncoder` (funasr/models/sanm/encoder.py:187-461, `encoder_classes["SANMEncoder"]`): same constructor
    keywords for the Paraformer recipe (input_layer "pe", normalize_before, selfattention_layer_type "sanm"), same
    state_dict keys (encoders0.0.*, encoders.{i}.*, after_norm.*), `output_size()`,
    `forward(xs_pad [B, T, input_size], ilens [B]) -> (xs [B, T, output_size], olens [B], None)`;
  * `SenseVoiceEncoderSmall` (funasr/models/sense_voice/model.py:488-655): + tp_encoders.{i}.*, tp_norm.*, LayerNorm
    eps 1e-5, `forward(xs_pad, ilens) -> (xs, olens)`.
Parameters are held as nn.Parameters (so .to(), state_dict(), load_state_dict(strict=True) behave like the
reference) and mirrored into library-owned HBM before the first forward.
"""
from __future__ import annotations

import os

from typing import List, Optional

import torch
import torch.nn as nn

from . import _lib
from .hip_module import Holder, HipModule, depthwise, device_lens, host_i32, layer_norm, linear, stream_ptr
from .register import tables


def sinusoidal_position_table(timesteps: int, depth: int, start: int = 0, device=None) -> torch.Tensor:
    """[timesteps, depth] table of SinusoidalPositionEncoder.encode (funasr/models/transformer/embedding.py:396-420),
    evaluated with the same float32 torch ops on the host so the device adds bit-identical values."""
    positions = torch.arange(1 + start, timesteps + start + 1)[None, :].type(torch.float32)
    inc = torch.log(torch.tensor([10000], dtype=torch.float32)) / (depth / 2 - 1)
    inv = torch.exp(torch.arange(depth / 2).type(torch.float32) * (-inc))
    scaled = positions.reshape(1, -1, 1) * inv.reshape(1, 1, -1)
    pe = torch.cat([torch.sin(scaled), torch.cos(scaled)], dim=2)[0].type(torch.float32).contiguous()
    return pe if device is None else pe.to(device)


def _block(in_dim: int, d_model: int, ffn: int, kernel_size: int) -> nn.Module:
    b = Holder()
    b.norm1 = layer_norm(in_dim)
    b.self_attn = Holder()
    b.self_attn.linear_q_k_v = linear(3 * d_model, in_dim)
    b.self_attn.fsmn_block = depthwise(d_model, kernel_size)
    b.self_attn.linear_out = linear(d_model, d_model)
    b.norm2 = layer_norm(d_model)
    b.feed_forward = Holder()
    b.feed_forward.w_1 = linear(ffn, d_model)
    b.feed_forward.w_2 = linear(d_model, ffn)
    return b


class _SANMEncoderBase(HipModule):
    _prefix = "pf_encoder"

    def __init__(self, input_size: int, output_size: int, attention_heads: int, linear_units: int, num_blocks: int,
                 tp_blocks: int, kernel_size: int, sanm_shfit: int, ln_eps: float):
        super().__init__()
        self._input_size, self._output_size = input_size, output_size
        self.attention_heads, self.linear_units = attention_heads, linear_units
        self.num_blocks, self.tp_blocks = num_blocks, tp_blocks
        self.kernel_size, self.sanm_shfit, self.ln_eps = kernel_size, sanm_shfit, ln_eps
        self.encoders0 = nn.ModuleList([_block(input_size, output_size, linear_units, kernel_size)])
        self.encoders = nn.ModuleList([_block(output_size, output_size, linear_units, kernel_size)
                                       for _ in range(num_blocks - 1)])
        if tp_blocks > 0:
            self.tp_encoders = nn.ModuleList([_block(output_size, output_size, linear_units, kernel_size)
                                              for _ in range(tp_blocks)])
        self.after_norm = layer_norm(output_size)
        if tp_blocks > 0:
            self.tp_norm = layer_norm(output_size)
        self._pe = None

    def output_size(self) -> int:
        return self._output_size

    def _default_precision(self) -> str:
        """The measured mode wherever its kernels exist (heads of d_k = 128, d_model % 256 == 0: Paraformer, SenseVoice and
        their variants), fp32 otherwise (the CT-Transformer's small heads)."""
        ok = self._output_size % 256 == 0 and self._output_size // self.attention_heads == 128 and self.linear_units % 256 == 0
        return "f16x2" if ok else "fp32"

    def _mode(self) -> str:
        return getattr(self, "_precision", None) or self._default_precision()

    def set_precision(self, mode: Optional[str] = None):
        """"f16x2" (the default where supported, see _default_precision): fp32-class results with every GEMM / attention
        operand split into two fp16 planes (x 2^e = hi + lo), three fp16 MFMA products per operand pair, fp32 accumulate --
        meets every fp32 parity bar (tests/test_parity_gpu.py). "fp32": every product on the exact fp32 MFMA (the opt-out;
        2.4x slower). "bf16x3": fp32-class results from three bf16 planes, six products. "bf16": bf16 operands with fp32
        accumulation / residual stream / LayerNorm / softmax -- bf16-class error (what the reference's `bf16=True` asks for).
        None restores the default."""
        if mode is None:
            self._precision = None
            return self
        if mode not in ("fp32", "bf16", "bf16x3", "f16x2"):
            raise ValueError("precision must be 'fp32', 'bf16', 'bf16x3' or 'f16x2'")
        self._precision = mode
        if self._handle is not None:
            _lib.check(_lib.load().pf_encoder_set_precision(self._handle, {"fp32": 0, "bf16": 1, "bf16x3": 2, "f16x2": 3}[mode]),
                       "pf_encoder_set_precision")
        return self

    ALL_ROWS = 1 << 30

    def set_row_packing(self, extra_rows: Optional[int] = ALL_ROWS):
        """f16x2 mode: how many rows of each sequence the encoder computes. `extra_rows` = k: the len_b valid rows plus k of the
        padding rows behind them (capped at T), sequences laid out back to back -- what a consumer that reads only a prefix
        needs (CTC: 0, CIF predictor: 1); the other rows of the output are zero. ALL_ROWS (default): every row of [B, T],
        still without alignment padding between sequences. None: the padded layout (every row, 16-row aligned sequences)."""
        self._row_packing = extra_rows
        return self

    def set_option(self, key: str, value: int):
        """Schedule options of the f16x2 mode (`pf_encoder_set_option`): "fuse_row" (1 default: linear_out / w_2 with the
        residual adds and the following LayerNorm in the GEMM epilogue; 0: separate launches, bitwise equal), "fsmn_fused" (1
        default: the FSMN memory block computed inside linear_out's epilogue; 0: its own launch, bitwise equal), "attn_variant"
        (3 default: lazy rescale; 1 pipelined; 0 plain), "row_bm" (rows per block of the full-row GEMMs: 0 default = by the
        batch's row count, 96 / 128 / 129 forced; bitwise equal)."""
        if not hasattr(self, "_options"):
            self._options = {}
        self._options[str(key)] = int(value)
        return self

    def _make_config(self):
        return _lib.pf_encoder_config(self._input_size, self._output_size, self.attention_heads, self.linear_units,
                                      self.num_blocks, self.tp_blocks, self.kernel_size, self.sanm_shfit, self.ln_eps)

    def _pe_table(self, T: int, dev) -> torch.Tensor:
        if self._pe is None or self._pe.shape[0] < T or self._pe.device != dev:
            self._pe = sinusoidal_position_table(max(T, 512), self._input_size, device=dev)
        return self._pe

    def _apply_settings(self, lib, h):
        """precision, row packing and schedule options of this module -> the handle (before every forward through it)"""
        _lib.check(lib.pf_encoder_set_precision(h, {"fp32": 0, "bf16": 1, "bf16x3": 2, "f16x2": 3}[self._mode()]),
                   "pf_encoder_set_precision")
        pack = getattr(self, "_row_packing", self.ALL_ROWS)
        _lib.check(lib.pf_encoder_set_row_packing(h, -1 if pack is None else int(pack)), "pf_encoder_set_row_packing")
        for key, value in getattr(self, "_options", {}).items():
            _lib.check(lib.pf_encoder_set_option(h, key.encode(), int(value)), "pf_encoder_set_option")

    def _run(self, xs_pad: torch.Tensor, ilens, run_blocks: int = -1):
        lib, h = self._ensure_handle()
        self._apply_settings(lib, h)
        dev = self._handle_device
        xs = xs_pad.to(device=dev, dtype=torch.float32).contiguous()
        B, T, Din = xs.shape
        if Din != self._input_size:
            raise ValueError(f"expected feature dim {self._input_size}, got {Din}")
        lens_c, lens = host_i32(ilens, B)
        width = self._output_size if run_blocks != 0 else Din
        out = torch.empty(B, T, width, device=dev, dtype=torch.float32)
        pe = self._pe_table(T, dev)
        with torch.cuda.device(dev):
            _lib.check(lib.pf_encoder_forward(h, xs.data_ptr(), lens_c, B, T, pe.data_ptr(), out.data_ptr(),
                                              run_blocks, stream_ptr()), "pf_encoder_forward")
        olens = device_lens(lens, dev)                               # (no synchronisation: hip_module.device_lens)
        return out, olens


@tables.register("encoder_classes", "SANMEncoder")
class SANMEncoder(_SANMEncoderBase):
    def __init__(self, input_size: int, output_size: int = 256, attention_heads: int = 4, linear_units: int = 2048,
                 num_blocks: int = 6, dropout_rate: float = 0.1, positional_dropout_rate: float = 0.1,
                 attention_dropout_rate: float = 0.0, input_layer: Optional[str] = "conv2d", pos_enc_class=None,
                 normalize_before: bool = True, concat_after: bool = False, positionwise_layer_type: str = "linear",
                 positionwise_conv_kernel_size: int = 1, padding_idx: int = -1, interctc_layer_idx: List[int] = [],
                 interctc_use_conditioning: bool = False, kernel_size: int = 11, sanm_shfit: int = 0,
                 lora_list: List[str] = None, lora_rank: int = 8, lora_alpha: int = 16, lora_dropout: float = 0.1,
                 selfattention_layer_type: str = "sanm", tf2torch_tensor_name_prefix_torch: str = "encoder",
                 tf2torch_tensor_name_prefix_tf: str = "seq2seq/encoder", **kwargs):
        if input_layer != "pe" or not normalize_before or concat_after or selfattention_layer_type != "sanm" \
                or positionwise_layer_type != "linear" or lora_list or interctc_layer_idx:
            raise NotImplementedError("SANMEncoder(HIP): only the Paraformer recipe (input_layer='pe', "
                                      "normalize_before, sanm self-attention, linear FFN, no LoRA/interCTC) is built")
        super().__init__(input_size, output_size, attention_heads, linear_units, num_blocks, 0, kernel_size,
                         sanm_shfit, 1e-12)

    def forward(self, xs_pad: torch.Tensor, ilens, prev_states=None, ctc=None):
        out, olens = self._run(xs_pad, ilens)
        return out, olens, None


@tables.register("encoder_classes", "SANMVadEncoder")
class SANMVadEncoder(SANMEncoder):
    """`SANMVadEncoder` (funasr/models/ct_transformer_streaming/encoder.py:175-430, the encoder of CTTransformerStreaming):
    the SAN-M encoder with the same parameters and state_dict keys, whose self-attention is causal in every block and masked
    by the VAD corner (transformer/utils/mask.py:38-52) in the last one. `forward(xs_pad, ilens, vad_indexes)`. fp32 mode."""

    def _default_precision(self) -> str:
        return "fp32"

    def forward(self, xs_pad: torch.Tensor, ilens, vad_indexes=None, prev_states=None, ctc=None):
        if self._mode() != "fp32":
            raise NotImplementedError("SANMVadEncoder(HIP): the masked attention is built for the fp32 mode")
        B = xs_pad.shape[0]
        vad = [0] * B if vad_indexes is None else [int(v) for v in torch.as_tensor(vad_indexes).reshape(-1).tolist()]
        if len(vad) != B:
            raise ValueError(f"vad_indexes holds {len(vad)} values for a batch of {B}")
        lib, h = self._ensure_handle()
        arr = (_lib.C.c_int32 * B)(*vad)
        _lib.check(lib.pf_encoder_set_vad_mask(h, arr, B), "pf_encoder_set_vad_mask")
        try:
            out, olens = self._run(xs_pad, ilens)
        finally:
            _lib.check(lib.pf_encoder_set_vad_mask(h, None, 0), "pf_encoder_set_vad_mask")
        return out, olens, None


@tables.register("encoder_classes", "SenseVoiceEncoderSmall")
class SenseVoiceEncoderSmall(_SANMEncoderBase):
    def __init__(self, input_size: int, output_size: int = 256, attention_heads: int = 4, linear_units: int = 2048,
                 num_blocks: int = 6, tp_blocks: int = 0, dropout_rate: float = 0.1,
                 positional_dropout_rate: float = 0.1, attention_dropout_rate: float = 0.0,
                 stochastic_depth_rate: float = 0.0, input_layer: Optional[str] = "conv2d", pos_enc_class=None,
                 normalize_before: bool = True, concat_after: bool = False, positionwise_layer_type: str = "linear",
                 positionwise_conv_kernel_size: int = 1, padding_idx: int = -1, kernel_size: int = 11,
                 sanm_shfit: int = 0, selfattention_layer_type: str = "sanm", **kwargs):
        if input_layer not in ("pe", None, "conv2d") or not normalize_before or concat_after:
            raise NotImplementedError("SenseVoiceEncoderSmall(HIP): unsupported configuration")
        super().__init__(input_size, output_size, attention_heads, linear_units, num_blocks, tp_blocks, kernel_size,
                         sanm_shfit, 1e-5)

    def forward(self, xs_pad: torch.Tensor, ilens):
        return self._run(xs_pad, ilens)
