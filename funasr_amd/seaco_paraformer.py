"""SeACo-Paraformer on gfx950: BiCifParaformer plus the hotword (semantic-aware contextual) decoder.

Host-side mirror of `SeacoParaformer` (funasr/models/seaco_paraformer/model.py:49-724, `model_classes["SeacoParaformer"]`,
the model behind the `paraformer-zh` alias): same constructor keywords and state_dict layout (bias_encoder.* = a 2-layer
LSTM, seaco_decoder.* = a ParaformerSANMDecoder with kernel 21 and no output layer, hotword_output_layer.*, and the
decoder's token table decoder.embed.0.weight that embeds the hotwords). `inference(..., hotword=None)` is BiCifParaformer's
path exactly like the reference (`_seaco_decode_with_ASF` returns the plain decoder scores without a hotword list,
:379-385); with `hotword="词一 词二"` or a .txt file (:583-690):

    hotword ids -> decoder.embed -> LSTM -> the state at each hotword's last token            (_hotword_representation)
    decoder(encoder_out, cif embeddings)            -> token ids + hidden states
    seaco_decoder(hotword states as memory, cif embeddings) + seaco_decoder(.., decoder hidden)  -> hotword_output_layer
    a position takes the bias decoder's token unless that token is NO_BIAS                    (_merge_res, seaco_weight 1)

The attention-score filter (ASF, :323-349) only runs for more than `nfilter` = 50 hotwords and is not built: longer lists
raise. All device work is enqueued before anything is read back; the merge of the two id arrays happens on the host.
"""
from __future__ import annotations

import logging
import os
import re
from typing import List, Optional

import torch

from . import ops
from .bicif_paraformer import BiCifParaformer
from .hip_module import Holder, linear
from .register import tables

_NFILTER = 50


def load_seg_dict(path: str) -> dict:
    seg = {}
    with open(path, "r", encoding="utf8") as f:
        for line in f:
            parts = line.strip().split()
            if parts:
                seg[parts[0]] = " ".join(parts[1:])
    return seg


def seg_tokenize(words: List[str], seg_dict: dict) -> List[str]:
    """hotword words -> model tokens through the model's seg_dict (:592-614): known words map to their pieces, unknown
    CJK / digit strings fall back to characters, everything else is <unk>"""
    pattern = re.compile(r"^[一-龥0-9]+$")
    out = ""
    for word in words:
        word = word.lower()
        if word in seg_dict:
            out += seg_dict[word] + " "
        elif pattern.match(word):
            for ch in word:
                out += (seg_dict[ch] if ch in seg_dict else "<unk>") + " "
        else:
            out += "<unk> "
    return out.strip().split()


@tables.register("model_classes", "SeacoParaformer")
class SeacoParaformer(BiCifParaformer):
    def __init__(self, *args, **kwargs):
        super().__init__(*args, **kwargs)
        self.inner_dim = kwargs.get("inner_dim", 256)
        self.bias_encoder_type = kwargs.get("bias_encoder_type", "lstm")
        if self.bias_encoder_type != "lstm" or kwargs.get("bias_encoder_bid", False):
            raise NotImplementedError("SeacoParaformer(HIP): the published uni-directional LSTM bias encoder is built")
        if self.inner_dim != self.encoder.output_size():
            raise NotImplementedError("SeacoParaformer(HIP): inner_dim must equal the encoder width (the hotword embeddings "
                                      "come from the decoder's token table)")
        D = self.inner_dim
        self.bias_encoder = Holder()
        for layer in range(2):
            for name, shape in (("weight_ih", (4 * D, D)), ("weight_hh", (4 * D, D)), ("bias_ih", (4 * D,)), ("bias_hh", (4 * D,))):
                self.bias_encoder.register_parameter(f"{name}_l{layer}", torch.nn.Parameter(torch.zeros(*shape), requires_grad=False))
        self.lstm_proj = None
        seaco_decoder = kwargs.get("seaco_decoder")
        if seaco_decoder is None:
            raise ValueError("SeacoParaformer needs `seaco_decoder` / `seaco_decoder_conf`")
        conf = dict(kwargs.get("seaco_decoder_conf") or {})
        self.seaco_decoder = tables.decoder_classes.get(seaco_decoder)(vocab_size=self.vocab_size, encoder_output_size=D, **conf)
        self.hotword_output_layer = linear(self.vocab_size, D)
        self.NO_BIAS = kwargs.get("NO_BIAS", 8377)
        self.predictor_name = kwargs.get("predictor")
        self.hotword_list = None
        self.nfilter = _NFILTER                              # `nfilter` of _seaco_decode_with_ASF (:277), 50 at its only call site
        self._hw_cache = {}

    # --------------------------------------------------------------------------------------------------- hotwords
    def generate_hotwords_list(self, hotword_list_or_file, tokenizer=None, frontend=None) -> Optional[List[List[int]]]:
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
        out.append([self.sos])                                          # the no-bias entry, always last (:640,:686)
        logging.info("hotword list: %d entries", len(out))
        return out

    def _hotword_representation(self, hw_list: List[List[int]]) -> torch.Tensor:
        """[n_hotwords, inner_dim] on the device: the LSTM state at each hotword's last token (:388-424). The recurrence is
        uni-directional, so running the padded batch and picking row len - 1 equals the reference's packed sequence."""
        key = tuple(tuple(h) for h in hw_list)
        hit = self._hw_cache.get(key)
        if hit is not None:
            return hit
        table = self.decoder.embed[0].weight.detach()
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
        for layer in range(2):
            x = ops.lstm(x, getattr(be, f"weight_ih_l{layer}").detach()[None], getattr(be, f"weight_hh_l{layer}").detach()[None],
                         getattr(be, f"bias_ih_l{layer}").detach()[None], getattr(be, f"bias_hh_l{layer}").detach()[None])
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
        hw_list = self.hotword_list
        if hw_list is None:
            return super().enqueue_features(speech, speech_lengths, return_intermediate)
        enc, olens = self.encode(speech, speech_lengths)
        embeds, token_num, alphas, peaks = self.calc_predictor(enc, olens)
        tok = [int(round(v)) for v in token_num.tolist()]
        B = enc.shape[0]
        ids = dha_ids = us_alphas = us_peaks = None
        if max(tok) >= 1:
            _, ids, dec_hidden, _ = self.decoder._run(enc, olens, embeds, tok, want_logits=False, want_ids=True, want_hidden=True)
            sel = self._hotword_representation(hw_list)
            ctx = sel[None].expand(B, -1, -1).contiguous()
            clen = [sel.shape[0]] * B
            nfilter, n_hot = self.nfilter, sel.shape[0]
            if 0 < nfilter < n_hot:
                # attention-score filter (:323-349): keep the nfilter hotwords the bias decoder's 6th block attends to most
                # (sequence 0's attention, summed over heads and token positions) plus the trailing no-bias entry
                scores = self.seaco_decoder.forward_asf6(ctx, clen, dec_hidden, tok)
                keep = torch.topk(scores.cpu(), min(nfilter, n_hot - 1))[1].tolist()
                keep.append(n_hot - 1)
                sel = sel[torch.tensor(keep, device=sel.device)].contiguous()
                ctx = sel[None].expand(B, -1, -1).contiguous()
                clen = [sel.shape[0]] * B
            cif_att, _ = self.seaco_decoder(ctx, clen, embeds, tok)
            dec_att, _ = self.seaco_decoder(ctx, clen, dec_hidden, tok)
            merged = (cif_att + dec_att).view(-1, cif_att.shape[-1])          # _merge (:193-200)
            w = self.hotword_output_layer
            dha_ids = ops.gemm_argmax(merged, w.weight.detach(), w.bias.detach()).view(B, -1)
            _, _, us_alphas, us_peaks = self.calc_predictor_timestamp(enc, olens, tok)
        pending = dict(tok=tok, ids=ids, dha_ids=dha_ids, B=B, extra=dict(us_alphas=us_alphas, us_peaks=us_peaks, olens=olens))
        if return_intermediate:
            pending["extra"].update(enc=enc, embeds=embeds, alphas=alphas, peaks=peaks)
        return pending

    def collect(self, pending: dict) -> dict:
        dha = pending.get("dha_ids")
        if dha is not None and pending["ids"] is not None:
            # _merge_res with seaco_weight 1 (:361-377): the bias decoder's token wins unless it says NO_BIAS. Both id arrays
            # are already arg-maxes, so the merge is an integer select (log_softmax does not move an arg-max)
            dec = pending["ids"]
            pending = dict(pending, ids=torch.where(dha == self.NO_BIAS, dec, dha.to(dec.dtype)))
        return super().collect(pending)

    # ---------------------------------------------------------------------------------------------- AutoModel API
    def _nothing_decoded(self, meta_data):
        return ([],)                                # seaco_paraformer/model.py:486-487: a one-tuple, i.e. no record at all

    def inference(self, data_in, data_lengths=None, key: list = None, tokenizer=None, frontend=None, **kwargs):
        self.hotword_list = self.generate_hotwords_list(kwargs.get("hotword", None), tokenizer=tokenizer, frontend=frontend)
        return super().inference(data_in, data_lengths=data_lengths, key=key, tokenizer=tokenizer, frontend=frontend, **kwargs)

    def inference_begin(self, data_in, data_lengths=None, key: list = None, tokenizer=None, frontend=None, **kwargs):
        self.hotword_list = self.generate_hotwords_list(kwargs.get("hotword", None), tokenizer=tokenizer, frontend=frontend)
        return super().inference_begin(data_in, data_lengths=data_lengths, key=key, tokenizer=tokenizer, frontend=frontend, **kwargs)
