"""ContextualParaformer (CLAS hotword biasing) on gfx950.

Host-side mirrors of `ContextualParaformerDecoder` (funasr/models/contextual_paraformer/decoder.py:133-352,
`decoder_classes["ContextualParaformerDecoder"]`) and `ContextualParaformer`
(funasr/models/contextual_paraformer/model.py:46-673, `model_classes["ContextualParaformer"]`) for greedy inference:
same constructor keywords and state_dict keys (decoder.decoders.{i}.*, decoder.last_decoder.*, decoder.bias_decoder.*,
decoder.bias_output.weight, bias_encoder.*, bias_embed.weight), `inference(..., hotword=..., clas_scale=...)`.

Device work: the hotword encoder is the embedding gather + the single-layer LSTM kernel (lstm.hip; the state at each
hotword's last token = h_n of the reference's packed sequence, model.py:353-365); the decoder handle runs the standard
blocks, then the last block's FSMN-side state queries the hotword embeddings through `bias_decoder` and the 1x1
`bias_output` fuses both attention outputs (pf_decoder_forward_contextual, fp32 kernels).
"""
from __future__ import annotations

import logging
import os
from typing import List, Optional

import torch
import torch.nn as nn

from . import _lib, ops
from .hip_module import Holder, ParamHolder, host_i32, layer_norm, linear, stream_ptr
from .paraformer import Paraformer
from .paraformer_decoder import ParaformerSANMDecoder, _block
from .register import tables
from .seaco_paraformer import load_seg_dict, seg_tokenize


@tables.register("decoder_classes", "ContextualParaformerDecoder")
class ContextualParaformerDecoder(ParaformerSANMDecoder):
    _create_name = "pf_decoder_create_contextual"

    def __init__(self, vocab_size: int, encoder_output_size: int, attention_heads: int = 4, linear_units: int = 2048,
                 num_blocks: int = 6, att_layer_num: int = 6, kernel_size: int = 21, sanm_shfit: int = 0, **kwargs):
        super().__init__(vocab_size, encoder_output_size, attention_heads=attention_heads, linear_units=linear_units,
                         num_blocks=num_blocks, att_layer_num=att_layer_num, kernel_size=kernel_size, sanm_shfit=sanm_shfit,
                         **kwargs)
        D = encoder_output_size
        # decoders: att_layer_num - 1 standard blocks; the last one lives under `last_decoder` (decoder.py:222-253)
        self.decoders = nn.ModuleList([_block(D, linear_units, kernel_size) for _ in range(att_layer_num - 1)])
        self.last_decoder = _block(D, linear_units, kernel_size)
        self.bias_decoder = Holder()
        self.bias_decoder.norm3 = layer_norm(D)
        self.bias_decoder.src_attn = Holder()
        self.bias_decoder.src_attn.linear_q = linear(D, D)
        self.bias_decoder.src_attn.linear_k_v = linear(2 * D, D)
        self.bias_decoder.src_attn.linear_out = linear(D, D)
        self.bias_output = ParamHolder((D, 2 * D, 1), None)            # Conv1d(2D, D, 1, bias=False)

    def _run_contextual(self, hs_pad, hlens, ys_in_pad, ys_in_lens, contextual_info, clas_scale, want_logits, want_ids):
        lib, h = self._ensure_handle()
        _lib.check(lib.pf_decoder_set_precision(h, 0), "pf_decoder_set_precision")        # the hotword branch is fp32
        dev = self._handle_device
        mem = hs_pad.to(device=dev, dtype=torch.float32).contiguous()
        emb = ys_in_pad.to(device=dev, dtype=torch.float32).contiguous()
        B, T, D = mem.shape
        N = emb.shape[1]
        ctx = contextual_info.to(device=dev, dtype=torch.float32)
        if ctx.shape[0] != B:
            ctx = ctx.expand(B, -1, -1)
        ctx = ctx.contiguous()
        mlen_c, _ = host_i32(hlens, B)
        tlen_c, tlens = host_i32(ys_in_lens, B)
        logits = torch.empty(B, N, self.vocab_size, device=dev, dtype=torch.float32) if want_logits else None
        ids = torch.empty(B, N, device=dev, dtype=torch.int32) if want_ids else None
        with torch.cuda.device(dev):
            _lib.check(lib.pf_decoder_forward_contextual(h, mem.data_ptr(), mlen_c, emb.data_ptr(), tlen_c, ctx.data_ptr(),
                                                         ctx.shape[1], float(clas_scale), B, T, N,
                                                         logits.data_ptr() if want_logits else None,
                                                         ids.data_ptr() if want_ids else None, None, stream_ptr()),
                       "pf_decoder_forward_contextual")
        return logits, ids, torch.tensor(tlens, dtype=torch.int64, device=dev)

    def forward(self, hs_pad, hlens, ys_in_pad, ys_in_lens, contextual_info=None, clas_scale: float = 1.0,
                return_hidden: bool = False, **kwargs):
        if contextual_info is None or return_hidden:
            raise NotImplementedError("ContextualParaformerDecoder(HIP): forward needs contextual_info and returns logits")
        logits, _, olens = self._run_contextual(hs_pad, hlens, ys_in_pad, ys_in_lens, contextual_info, clas_scale, True, False)
        return logits, olens

    def greedy(self, hs_pad, hlens, ys_in_pad, ys_in_lens, contextual_info=None, clas_scale: float = 1.0):
        _, ids, olens = self._run_contextual(hs_pad, hlens, ys_in_pad, ys_in_lens, contextual_info, clas_scale, False, True)
        return ids, olens


@tables.register("model_classes", "ContextualParaformer")
class ContextualParaformer(Paraformer):
    _unwrap_key_lists = False       # contextual_paraformer/model.py:463-540 uses `key[i]` as it comes

    def __init__(self, *args, **kwargs):
        super().__init__(*args, **kwargs)
        inner_dim = kwargs.get("inner_dim", 256)
        self.bias_encoder_type = kwargs.get("bias_encoder_type", "lstm")
        self.use_decoder_embedding = kwargs.get("use_decoder_embedding", False)
        if self.bias_encoder_type != "lstm":
            # the reference's `mean` branch builds no bias_encoder, yet cal_decoder_with_predictor calls it unconditionally
            # (contextual_paraformer/model.py:81-82,357,371): that option cannot decode there either (tests/test_reference_unreachable_options.py)
            raise NotImplementedError("ContextualParaformer(HIP): bias_encoder_type must be 'lstm' (the published models); the "
                                      "reference's 'mean' option has no inference path either (its decoder call needs bias_encoder)")
        if inner_dim != self.encoder.output_size():
            raise NotImplementedError("ContextualParaformer(HIP): inner_dim must equal the decoder width (the hotword "
                                      "embeddings are the bias decoder's keys and values)")
        D = inner_dim
        self.bias_encoder = Holder()                                  # nn.LSTM(inner_dim, inner_dim, 1, batch_first=True)
        for name, shape in (("weight_ih_l0", (4 * D, D)), ("weight_hh_l0", (4 * D, D)), ("bias_ih_l0", (4 * D,)), ("bias_hh_l0", (4 * D,))):
            self.bias_encoder.register_parameter(name, nn.Parameter(torch.zeros(*shape), requires_grad=False))
        self.bias_embed = nn.Embedding(self.vocab_size, D)
        self.bias_embed.weight.requires_grad_(False)
        self.hotword_list = None
        self._hw_cache = {}

    # --------------------------------------------------------------------------------------------------- hotwords
    def _decoder_logits(self, enc, olens, embeds, tok) -> torch.Tensor:
        """The beam-search route ranks the hotword-biased scores too: the reference hands hw_list / clas_scale to
        cal_decoder_with_predictor before either branch (contextual_paraformer/model.py:467-494)."""
        hw = self._hotword_embeddings(self.hotword_list)
        return self.decoder(enc, olens, embeds, tok, contextual_info=hw[None], clas_scale=self.clas_scale)[0]

    def generate_hotwords_list(self, hotword_list_or_file, tokenizer=None, frontend=None) -> Optional[List[List[int]]]:
        """model.py:534-657: a local .txt file (one hotword per line) or a space-separated string; every entry is tokenised
        (through seg_dict when the model directory has one) and [sos] is appended as the no-hotword entry"""
        seg_dict = None
        cmvn_file = getattr(frontend, "cmvn_file", None)
        if cmvn_file is not None:
            seg_path = os.path.join(os.path.dirname(cmvn_file), "seg_dict")
            seg_dict = load_seg_dict(seg_path) if os.path.exists(seg_path) else None
        if hotword_list_or_file is None:
            return None

        def ids_of(words: List[str]) -> List[int]:
            return tokenizer.tokens2ids(seg_tokenize(words, seg_dict) if seg_dict is not None else words)

        if os.path.exists(hotword_list_or_file) and hotword_list_or_file.endswith(".txt"):
            with open(hotword_list_or_file, "r", encoding="utf-8") as f:
                out = [ids_of(line.strip().split()) for line in f.readlines()]
        elif hotword_list_or_file.startswith("http"):
            raise NotImplementedError("hotword lists by URL need a network; pass a local .txt file or a string")
        elif not hotword_list_or_file.endswith(".txt"):
            out = [ids_of(hw.strip().split()) for hw in hotword_list_or_file.strip().split()]
        else:
            return None
        out.append([self.sos])
        logging.info("hotword list: %d entries", len(out))
        return out

    def _hotword_embeddings(self, hw_list: Optional[List[List[int]]]) -> torch.Tensor:
        """[n_hotwords, D] on the device: h_n of the single-layer LSTM over each hotword's tokens (model.py:345-365; the
        recurrence is causal, so the padded batch's state at position len - 1 IS the packed sequence's h_n). hw_list None ->
        the single [sos] entry (:345-352)."""
        if hw_list is None:
            hw_list = [[1]]
        key = tuple(tuple(h) for h in hw_list)
        hit = self._hw_cache.get(key)
        if hit is not None:
            return hit
        table = (self.decoder.embed[0].weight if self.use_decoder_embedding else self.bias_embed.weight).detach()
        dev = table.device
        lens = [len(h) for h in hw_list]
        if min(lens) < 1:
            raise ValueError("empty hotword")
        L = max(lens)
        pad = torch.zeros(len(hw_list), L, dtype=torch.int32)
        for i, h in enumerate(hw_list):
            pad[i, : len(h)] = torch.tensor(h, dtype=torch.int32)
        x = ops.gather_rows(table.to(torch.float32), pad.view(-1).to(dev)).view(len(hw_list), L, -1)
        be = self.bias_encoder
        x = ops.lstm(x, be.weight_ih_l0.detach()[None], be.weight_hh_l0.detach()[None], be.bias_ih_l0.detach()[None],
                     be.bias_hh_l0.detach()[None])
        sel = x[torch.arange(len(hw_list), device=dev), torch.tensor(lens, device=dev) - 1].contiguous()
        self._hw_cache = {key: sel}
        return sel

    def load_state_dict(self, *args, **kwargs):
        self._hw_cache = {}
        return super().load_state_dict(*args, **kwargs)

    def _apply(self, fn, *args, **kwargs):
        self._hw_cache = {}
        return super()._apply(fn, *args, **kwargs)

    # ------------------------------------------------------------------------------------------- device pipeline
    def enqueue_features(self, speech: torch.Tensor, speech_lengths, return_intermediate: bool = False):
        enc, olens = self.encode(speech, speech_lengths, all_rows=return_intermediate)
        embeds, token_num, alphas, peaks = self.calc_predictor(enc, olens)
        tok = [int(round(v)) for v in token_num.tolist()]
        ids = None
        if max(tok) >= 1:
            hw = self._hotword_embeddings(self.hotword_list)
            ids, _ = self.decoder.greedy(enc, olens, embeds, tok, contextual_info=hw[None], clas_scale=self.clas_scale)
        pending = dict(tok=tok, ids=ids, B=enc.shape[0])
        if return_intermediate:
            pending["extra"] = dict(enc=enc, olens=olens, embeds=embeds, alphas=alphas, peaks=peaks)
        return pending

    clas_scale = 1.0

    def inference(self, data_in, data_lengths=None, key: list = None, tokenizer=None, frontend=None, **kwargs):
        self.hotword_list = self.generate_hotwords_list(kwargs.get("hotword", None), tokenizer=tokenizer, frontend=frontend)
        self.clas_scale = kwargs.get("clas_scale", 1.0)
        return super().inference(data_in, data_lengths=data_lengths, key=key, tokenizer=tokenizer, frontend=frontend, **kwargs)

    def inference_begin(self, data_in, data_lengths=None, key: list = None, tokenizer=None, frontend=None, **kwargs):
        self.hotword_list = self.generate_hotwords_list(kwargs.get("hotword", None), tokenizer=tokenizer, frontend=frontend)
        self.clas_scale = kwargs.get("clas_scale", 1.0)
        return super().inference_begin(data_in, data_lengths=data_lengths, key=key, tokenizer=tokenizer, frontend=frontend, **kwargs)
