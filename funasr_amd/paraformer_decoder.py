"""Paraformer SAN-M decoder on gfx950.

Host-side mirror of `ParaformerSANMDecoder` (funasr/models/paraformer/decoder.py:233-449,
`decoder_classes["ParaformerSANMDecoder"]`): same constructor keywords, state_dict keys (decoders.{i}.*,
decoders3.0.*, after_norm.*, output_layer.*, embed.0.weight), and
`forward(hs_pad, hlens, ys_in_pad, ys_in_lens) -> (logits [B, N, V], olens)`.
`greedy()` is the fused route used by Paraformer.inference: the vocabulary projection and the arg-max run in one
kernel and the [B, N, 8404] logits never reach HBM.
"""
from __future__ import annotations

from typing import List

import torch
import torch.nn as nn

from . import _lib
from .hip_module import Holder, HipModule, ParamHolder, depthwise, device_lens, host_i32, layer_norm, linear, stream_ptr
from .register import tables


def _ffn(block, d_model, ffn):
    block.norm1 = layer_norm(d_model)
    block.feed_forward = Holder()
    block.feed_forward.w_1 = linear(ffn, d_model)
    block.feed_forward.norm = layer_norm(ffn)
    block.feed_forward.w_2 = linear(d_model, ffn, bias=False)


def _block2(d_model, ffn, kernel_size):
    """decoders2 block: DecoderLayerSANM(self_attn=MultiHeadedAttentionSANMDecoder(sanm_shfit=0), src_attn=None) -- norm1 / FFN / norm2 /
    FSMN, no norm3, no cross-attention (decoder.py:363-380, :65-68)"""
    b = Holder()
    _ffn(b, d_model, ffn)
    b.norm2 = layer_norm(d_model)
    b.self_attn = Holder()
    b.self_attn.fsmn_block = depthwise(d_model, kernel_size)
    return b


def _block(d_model, ffn, kernel_size):
    b = Holder()
    _ffn(b, d_model, ffn)
    b.norm2 = layer_norm(d_model)
    b.self_attn = Holder()
    b.self_attn.fsmn_block = depthwise(d_model, kernel_size)
    b.norm3 = layer_norm(d_model)
    b.src_attn = Holder()
    b.src_attn.linear_q = linear(d_model, d_model)
    b.src_attn.linear_k_v = linear(2 * d_model, d_model)
    b.src_attn.linear_out = linear(d_model, d_model)
    return b


@tables.register("decoder_classes", "ParaformerSANMDecoder")
class ParaformerSANMDecoder(HipModule):
    _prefix = "pf_decoder"
    _skip_keys = ("embed.",)      # token embedding: training / sampler only (decoder.py:314-317)

    def __init__(self, vocab_size: int, encoder_output_size: int, attention_heads: int = 4, linear_units: int = 2048,
                 num_blocks: int = 6, dropout_rate: float = 0.1, positional_dropout_rate: float = 0.1,
                 self_attention_dropout_rate: float = 0.0, src_attention_dropout_rate: float = 0.0,
                 input_layer: str = "embed", use_output_layer: bool = True, wo_input_layer: bool = False,
                 pos_enc_class=None, normalize_before: bool = True, concat_after: bool = False,
                 att_layer_num: int = 6, kernel_size: int = 21, sanm_shfit: int = 0, lora_list: List[str] = None,
                 lora_rank: int = 8, lora_alpha: int = 16, lora_dropout: float = 0.1,
                 chunk_multiply_factor: tuple = (1,), tf2torch_tensor_name_prefix_torch: str = "decoder",
                 tf2torch_tensor_name_prefix_tf: str = "seq2seq/decoder", **kwargs):
        super().__init__()
        if not normalize_before or concat_after or lora_list:
            raise NotImplementedError("ParaformerSANMDecoder(HIP): normalize_before, no concat_after, no LoRA is built")
        if sanm_shfit is None:
            sanm_shfit = (kernel_size - 1) // 2
        D = encoder_output_size
        self.vocab_size, self.d_model, self.attention_heads, self.linear_units = vocab_size, D, attention_heads, linear_units
        self.att_layer_num, self.num_blocks, self.kernel_size, self.sanm_shfit = att_layer_num, num_blocks, kernel_size, sanm_shfit
        if not wo_input_layer and input_layer == "embed":
            self.embed = nn.ModuleList([ParamHolder((vocab_size, D), None)])
        self.decoders = nn.ModuleList([_block(D, linear_units, kernel_size) for _ in range(att_layer_num)])
        # decoder.py:363-380: num_blocks - att_layer_num blocks without cross-attention (None when there are none)
        self.decoders2 = (nn.ModuleList([_block2(D, linear_units, kernel_size) for _ in range(num_blocks - att_layer_num)])
                          if num_blocks - att_layer_num > 0 else None)
        last = Holder()
        _ffn(last, D, linear_units)
        self.decoders3 = nn.ModuleList([last])
        self.after_norm = layer_norm(D)
        # SeACo's bias decoder is built with use_output_layer false: forward() returns the hidden states (decoder.py:332-335)
        self.output_layer = linear(vocab_size, D) if use_output_layer else None

    def _default_precision(self) -> str:
        ok = self.d_model % 256 == 0 and self.d_model // self.attention_heads == 128 and self.linear_units % 256 == 0
        return "f16x2" if ok else "fp32"

    def _mode(self) -> str:
        return getattr(self, "_precision", None) or self._default_precision()

    def set_precision(self, mode=None):
        """"f16x2" (default where supported: fp32-class results from two-plane fp16 operands on the fp16 matrix cores), "fp32"
        (exact fp32 MFMA, the opt-out), "bf16x3", "bf16" (bf16 operands on the greedy route); see SANMEncoder.set_precision.
        None restores the default."""
        if mode is not None and mode not in ("fp32", "bf16", "bf16x3", "f16x2"):
            raise ValueError("precision must be 'fp32', 'bf16', 'bf16x3' or 'f16x2'")
        self._precision = mode
        return self

    def _make_config(self):
        return _lib.pf_decoder_config(self.vocab_size if self.output_layer is not None else 0, self.d_model, self.attention_heads, self.linear_units,
                                      self.att_layer_num, self.kernel_size, self.sanm_shfit, 1e-12)

    def _after_create(self, lib, handle):
        if self.decoders2 is not None:
            _lib.check(lib.pf_decoder_set_decoders2(handle, len(self.decoders2)), "pf_decoder_set_decoders2")

    def _apply_settings(self):
        lib, h = self._ensure_handle()
        _lib.check(lib.pf_decoder_set_precision(h, {"fp32": 0, "bf16": 1, "bf16x3": 2, "f16x2": 3}[self._mode()]),
                   "pf_decoder_set_precision")
        return lib, h

    def _run(self, hs_pad, hlens, ys_in_pad, ys_in_lens, want_logits: bool, want_ids: bool, want_hidden: bool = False):
        lib, h = self._apply_settings()
        dev = self._handle_device
        mem = hs_pad.to(device=dev, dtype=torch.float32).contiguous()
        emb = ys_in_pad.to(device=dev, dtype=torch.float32).contiguous()
        B, T, D = mem.shape
        N = emb.shape[1]
        mlen_c, _ = host_i32(hlens, B)
        tlen_c, tlens = host_i32(ys_in_lens, B)
        logits = torch.empty(B, N, self.vocab_size, device=dev, dtype=torch.float32) if want_logits else None
        ids = torch.empty(B, N, device=dev, dtype=torch.int32) if want_ids else None
        hid = torch.empty(B, N, D, device=dev, dtype=torch.float32) if want_hidden else None
        with torch.cuda.device(dev):
            _lib.check(lib.pf_decoder_forward(h, mem.data_ptr(), mlen_c, emb.data_ptr(), tlen_c, B, T, N,
                                              logits.data_ptr() if want_logits else None,
                                              ids.data_ptr() if want_ids else None,
                                              hid.data_ptr() if want_hidden else None, stream_ptr()),
                       "pf_decoder_forward")
        olens = device_lens(tlens, dev, torch.int64)
        return logits, ids, hid, olens

    def forward_asf6(self, hs_pad, hlens, ys_in_pad, ys_in_lens, n_blocks_before: int = 5) -> torch.Tensor:
        """decoder.py:485-513 reduced to what SeACo's filter uses (seaco_paraformer/model.py:326-329): the attention
        probabilities of block `n_blocks_before` for sequence 0, summed over heads and token positions -> [T] (device)."""
        lib, h = self._ensure_handle()
        dev = self._handle_device
        mem = hs_pad.to(device=dev, dtype=torch.float32).contiguous()
        emb = ys_in_pad.to(device=dev, dtype=torch.float32).contiguous()
        B, T, _ = mem.shape
        N = emb.shape[1]
        mlen_c, _ = host_i32(hlens, B)
        tlen_c, _ = host_i32(ys_in_lens, B)
        scores = torch.empty(T, device=dev, dtype=torch.float32)
        with torch.cuda.device(dev):
            _lib.check(lib.pf_decoder_asf_scores(h, mem.data_ptr(), mlen_c, emb.data_ptr(), tlen_c, B, T, N, n_blocks_before,
                                                 scores.data_ptr(), stream_ptr()), "pf_decoder_asf_scores")
        return scores

    def forward(self, hs_pad, hlens, ys_in_pad, ys_in_lens, chunk_mask=None, return_hidden: bool = False,
                return_both: bool = False):
        if chunk_mask is not None:
            raise NotImplementedError("chunk_mask is a training/streaming feature")
        if self.output_layer is None:
            _, _, hid, olens = self._run(hs_pad, hlens, ys_in_pad, ys_in_lens, want_logits=False, want_ids=False, want_hidden=True)
            return hid, olens
        logits, _, hid, olens = self._run(hs_pad, hlens, ys_in_pad, ys_in_lens, want_logits=not return_hidden or return_both,
                                          want_ids=False, want_hidden=return_hidden or return_both)
        if return_both:
            return logits, hid, olens
        if return_hidden:
            return hid, olens
        return logits, olens

    def greedy(self, hs_pad, hlens, ys_in_pad, ys_in_lens):
        """int32 [B, N] arg-max token ids (== log_softmax(...).argmax(-1), paraformer/model.py:345,642)."""
        _, ids, _, olens = self._run(hs_pad, hlens, ys_in_pad, ys_in_lens, want_logits=False, want_ids=True)
        return ids, olens
