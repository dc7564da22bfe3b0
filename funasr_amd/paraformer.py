"""Paraformer on gfx950: the model-level glue of the hot path.

Host-side mirror of `Paraformer` (funasr/models/paraformer/model.py:27-697, `model_classes["Paraformer"]`): it
builds encoder / predictor / decoder BY NAME through the registry exactly like :125-150, keeps the reference's
state_dict layout (encoder.*, predictor.*, decoder.*) and implements
`inference(data_in, data_lengths, key, tokenizer, frontend, **kwargs) -> (results, meta_data)` (:534-697, greedy path)
with the contract AutoModel relies on (funasr/auto/auto_model.py:812-829; executable spec tests/test_auto_model.py
in the reference). Unlike the reference the whole batch stays in HBM between the stages, the decoder's vocabulary
projection is fused with the arg-max, and there is exactly one device->host copy of token ids per batch.
"""
from __future__ import annotations

import os
import time
from typing import Dict, List, Optional

import torch
import torch.nn as nn

from .audio import batch_to_features, load_audio_list
import ctypes as C

from . import _lib
from .hip_module import HostCopyRing, StagedUpload, host_i32, stream_ptr
from .register import tables
from .timestamps import cif_token_spans
from .tokenizer import sentence_postprocess

# importing registers the classes under the reference's names
from . import cif_predictor as _cif_predictor  # noqa: F401
from . import paraformer_decoder as _paraformer_decoder  # noqa: F401
from . import sanm_encoder as _sanm_encoder  # noqa: F401
from . import wav_frontend as _wav_frontend  # noqa: F401


@tables.register("model_classes", "Paraformer")
class Paraformer(nn.Module):
    _always_timestamps = False      # BiCifParaformer / SeACo return token timestamps on every call
    # model.py:624-627 unwraps a list-of-lists `key` (what AutoModel hands over for list inputs with a key list) and repeats a
    # short key list; the BiCif / SeACo / Contextual classes of the reference do neither (their records then carry the list)
    _unwrap_key_lists = True

    def __init__(self, specaug: Optional[str] = None, specaug_conf: Optional[Dict] = None, normalize: str = None,
                 normalize_conf: Optional[Dict] = None, encoder: str = None, encoder_conf: Optional[Dict] = None,
                 decoder: str = None, decoder_conf: Optional[Dict] = None, ctc: str = None,
                 ctc_conf: Optional[Dict] = None, predictor: str = None, predictor_conf: Optional[Dict] = None,
                 ctc_weight: float = 0.5, input_size: int = 80, vocab_size: int = -1, ignore_id: int = -1,
                 blank_id: int = 0, sos: int = 1, eos: int = 2, lsm_weight: float = 0.0,
                 length_normalized_loss: bool = False, predictor_weight: float = 0.0, predictor_bias: int = 0,
                 sampling_ratio: float = 0.2, share_embedding: bool = False, use_1st_decoder_loss: bool = False,
                 **kwargs):
        super().__init__()
        enc_conf = dict(encoder_conf or {})
        enc_conf.pop("input_size", None)
        self.encoder = tables.encoder_classes.get(encoder)(input_size=input_size, **enc_conf)
        d = self.encoder.output_size()
        dec_conf = dict(decoder_conf or {})
        dec_conf.pop("vocab_size", None)
        dec_conf.pop("encoder_output_size", None)
        self.decoder = tables.decoder_classes.get(decoder)(vocab_size=vocab_size, encoder_output_size=d, **dec_conf)
        self.predictor = tables.predictor_classes.get(predictor)(**(predictor_conf or {}))
        self.blank_id, self.vocab_size, self.ignore_id = blank_id, vocab_size, ignore_id
        self.sos = sos if sos is not None else vocab_size - 1
        self.eos = eos if eos is not None else vocab_size - 1
        self.ctc, self.specaug, self.normalize = None, None, None
        if normalize is not None:                            # model.py:128-130: UtteranceMVN / GlobalMVN (funasr_amd/normalize.py)
            from . import normalize as _normalize  # noqa: F401  (registers normalize_classes)
            self.normalize = tables.normalize_classes.get(normalize)(**(normalize_conf or {}))
        if ctc_weight > 0.0:                                 # model.py:109-111: the CTC head exists only then
            from .ctc import CTC
            self.ctc = CTC(odim=vocab_size, encoder_output_size=d, **(ctc_conf or {}))
        self.ctc_weight = ctc_weight
        self.beam_search = None
        self.nbest = 1
        self._one_call = True                                # enqueue_features through pf_paraformer_forward where the model is the plain one
        if kwargs.get("precision"):                      # model_conf: {precision: f16x2 | fp32 | bf16x3 | bf16}
            self.set_precision(kwargs["precision"])

    # ------------------------------------------------------------------------------------------------ builders
    @classmethod
    def from_config(cls, cfg: dict) -> "Paraformer":
        """cfg in the layout of funasr_amd.synth.PARAFORMER_LARGE (keys as in models/paraformer/template.yaml)."""
        ec = dict(cfg["encoder"])
        input_size = ec.pop("input_size")
        dc = dict(cfg["decoder"])
        vocab = dc.pop("vocab_size")
        dc.pop("encoder_output_size", None)
        return cls(encoder="SANMEncoder", encoder_conf=dict(ec, input_layer="pe"), decoder="ParaformerSANMDecoder",
                   decoder_conf=dc, predictor="CifPredictorV2", predictor_conf=dict(cfg["predictor"]), ctc_weight=0.0,
                   input_size=input_size, vocab_size=vocab)

    def set_precision(self, mode=None):
        """"f16x2" (the default): fp32-class results, every GEMM / attention operand as two fp16 planes on the fp16 matrix
        cores (three products, fp32 accumulate) -- the measured mode, meets every fp32 parity bar. "fp32": exact-fp32 MFMA
        everywhere (the opt-out, 2.4x slower). "bf16x3": fp32-class results from three bf16 planes (six products). "bf16": bf16
        operands (fp32 accumulate / residual / LN / softmax / FSMN), bf16-class error. The CIF predictor is fp32 in every
        mode. None restores the default."""
        self.encoder.set_precision(mode)
        self.decoder.set_precision(mode)
        if getattr(self, "ctc", None) is not None and hasattr(self.ctc, "set_precision"):
            self.ctc.set_precision(mode)
        return self

    # ------------------------------------------------------------------------------------------- device pipeline
    def encode(self, speech: torch.Tensor, speech_lengths, all_rows: bool = False, **kwargs):
        """model.py:286-313. The reference's encoder computes every row of the padded batch; what reads its output here --
        CifPredictorV2 (the conv reaches r_order rows past the last valid frame and the tail weight sits on row len,
        cif_predictor.py:196-205,275-277), the decoder's cross-attention and the CTC head (rows < len) -- needs rows
        <= len + r_order - 1 only, so in the f16x2 mode only those are computed (the rest of `out` is zero). `all_rows`, the
        V3 predictor (its timestamp head runs over the whole padded tensor) and other encoders keep every row."""
        if hasattr(self.encoder, "set_row_packing"):
            v2_only = type(self.predictor).__name__ == "CifPredictorV2"
            self.encoder.set_row_packing(max(1, int(self.predictor.r_order)) if (v2_only and not all_rows) else self.encoder.ALL_ROWS)
        speech, speech_lengths = self._normalized(speech, speech_lengths)
        out, olens, _ = self.encoder(speech, speech_lengths)
        return out, olens

    def _normalized(self, speech: torch.Tensor, speech_lengths):
        """model.py:304-306: feature normalisation (Global-CMVN / Utterance-CMVN) in front of the encoder, in place like the reference"""
        if self.normalize is None:
            return speech, speech_lengths
        dev = self.encoder._device() if hasattr(self.encoder, "_device") else speech.device
        return self.normalize(speech.to(device=dev, dtype=torch.float32).contiguous(), speech_lengths)

    # what `inference` returns when no utterance of the batch predicts a token: the reference returns a BARE empty list there
    # (model.py:615-616; bicif_paraformer/model.py:342-343, contextual_paraformer/model.py:460-461), which AutoModel.inference
    # turns into ONE record {"text": ""} (auto_model.py:815-817); SeacoParaformer returns ([],) -- no record (its override)
    def _nothing_decoded(self, meta_data):
        return []

    def calc_predictor(self, encoder_out, encoder_out_lens):
        return self.predictor(encoder_out, None, None, ignore_id=self.ignore_id, lengths=encoder_out_lens)

    # ---- the one-call route (include/paraformer_hip.h pf_paraformer_forward): the plain offline model -- CifPredictorV2, the
    #      ParaformerSANMDecoder with its own output layer -- hands the whole chain to the library; subclasses with other predictors /
    #      decoders and callers that want the intermediate tensors keep the module-by-module chain below (bitwise the same ids)
    def _one_call_ok(self) -> bool:
        return (getattr(self, "_one_call", True) and os.environ.get("PF_ONE_CALL", "1") != "0" and type(self).__name__ in ("Paraformer", "ParaformerHip") and type(self.predictor).__name__ == "CifPredictorV2"
                and type(self.decoder).__name__ == "ParaformerSANMDecoder" and hasattr(self.encoder, "_apply_settings"))

    def _pipeline(self):
        lib, he = self.encoder._ensure_handle()
        _, hp = self.predictor._ensure_handle()
        _, hd = self.decoder._ensure_handle()
        key = (int(he or 0), int(hp or 0), int(hd or 0))
        cur = self.__dict__.get("_pipe")
        if cur is None or cur[0] != key:
            if cur is not None:
                lib.pf_paraformer_destroy(cur[1])
            h = _lib.check_handle(lib.pf_paraformer_create(he, hp, hd), "pf_paraformer_create")
            self.__dict__["_pipe"] = cur = (key, h)
        return lib, cur[1]

    def close(self):
        """Frees the pipeline object's device buffers (encoder outputs of two batches, embeds, ids). The module handles stay."""
        cur = self.__dict__.pop("_pipe", None)
        if cur is not None:
            try:
                _lib.load().pf_paraformer_destroy(cur[1])
            except Exception:
                pass

    def __del__(self):
        self.close()

    # the pipeline handle is a raw pointer into this process's library: never copied or pickled with the module
    def __getstate__(self):
        st = dict(self.__dict__)
        for k in ("_pipe", "_host_ring", "_upload", "_dec_stream"):
            st.pop(k, None)
        return st

    def begin_features(self, speech: torch.Tensor, speech_lengths):
        """Phase 1 of the offline forward (include/paraformer_hip.h pf_paraformer_begin): encoder + predictor scan ENQUEUED on the
        current HIP stream, no host synchronisation; returns a ticket for `finish_features`. A serving loop issues
        begin(batch i + 1) BEFORE finish(batch i): the CIF token count (the .item() of cif_predictor.py:311) still sizes the
        decoder exactly, but while the host reads it the GPU already holds the next batch's encoder."""
        if not self._one_call_ok():
            raise RuntimeError("begin_features: the split-phase forward exists for the plain offline model (CifPredictorV2 + ParaformerSANMDecoder)")
        lib, h = self._pipeline()
        enc_m = self.encoder
        enc_m.set_row_packing(max(1, int(self.predictor.r_order)))
        lib_e, he = enc_m._ensure_handle()
        enc_m._apply_settings(lib_e, he)
        self.decoder._apply_settings()
        dev = enc_m._handle_device
        xs = speech.to(device=dev, dtype=torch.float32).contiguous()
        if self.normalize is not None:
            xs, speech_lengths = self.normalize(xs, speech_lengths)
        B, T, Din = xs.shape
        if Din != enc_m._input_size:
            raise ValueError(f"expected feature dim {enc_m._input_size}, got {Din}")
        lens_c, _ = host_i32(speech_lengths, B)
        pe = enc_m._pe_table(T, dev)
        with torch.cuda.device(dev):
            t = lib.pf_paraformer_begin(h, xs.data_ptr(), lens_c, B, T, pe.data_ptr(), stream_ptr())
        if t < 0:
            _lib.check(t, "pf_paraformer_begin")
        return dict(ticket=t, B=B, T=T, dev=dev, keep=(xs, lens_c, pe))

    def finish_features(self, ticket: dict, stream: "torch.cuda.Stream" = None):
        """Phase 2 (pf_paraformer_finish): waits for the ticket's token counts only, enqueues embeds + decoder + fused arg-max and
        the ids' D2H copy; returns what `enqueue_features` returns (`collect()` brings the ids to the host). `stream`: another HIP
        stream for this phase -- batch i's decoder then runs BESIDE batch i + 1's encoder (the library orders the reuse of its
        buffers with an event); `pending["ready"]` is the event behind the phase's last kernel."""
        lib, h = self._pipeline()
        B, T, dev = ticket["B"], ticket["T"], ticket["dev"]
        tok_c = (C.c_int32 * B)()
        with torch.cuda.device(dev), torch.cuda.stream(stream if stream is not None else torch.cuda.current_stream(dev)):
            ids = torch.empty(B, T + 1, device=dev, dtype=torch.int32)      # a CIF fires at most once per frame (+ the tail)
            n = lib.pf_paraformer_finish(h, ticket["ticket"], ids.data_ptr(), T + 1, tok_c, None, None, stream_ptr())
            if n < 0:
                _lib.check(n, "pf_paraformer_finish")
            tok = [int(v) for v in tok_c]
            pending = dict(tok=tok, ids=ids if n >= 1 else None, B=B, keep=ticket["keep"])
            if n >= 1:
                pending["ids_host"] = (ids, self.__dict__.setdefault("_host_ring", HostCopyRing()).start(ids))
            if stream is not None:
                pending["ready"] = torch.cuda.Event()
                pending["ready"].record(stream)
        return pending

    def enqueue_features(self, speech: torch.Tensor, speech_lengths, return_intermediate: bool = False):
        """[B, T, 560] features -> everything up to the fused arg-max ENQUEUED on the current HIP stream. The only host wait
        inside is for the CIF token count (it sizes the decoder, like the .item() at cif_predictor.py:311); `begin_features` /
        `finish_features` are its two halves for loops that interleave batches. `collect()` brings the ids to the host; a serving
        loop enqueues batch i+1 before collecting batch i, so the GPU never waits for the host-side post-processing."""
        if not return_intermediate and self._one_call_ok():
            return self.finish_features(self.begin_features(speech, speech_lengths))
        return self.enqueue_finish(self.enqueue_begin(speech, speech_lengths, return_intermediate))

    # The module-by-module chain in two halves around its one host wait (the CIF token counts size the decoder, like the .item() at
    # cif_predictor.py:311): `enqueue_begin` puts encoder + predictor and the counts' D2H copy on the stream and returns at once
    # (pf_predictor_alphas_begin), `enqueue_finish` waits for the counts and enqueues the rest. A loop over batches calls
    # begin(i + 1) before finish(i) (AutoModel.inference through inference_begin / inference_launch). Subclasses that change the
    # chain override `enqueue_features` (and are then driven in two parts) or the halves themselves (BiCifParaformer).
    def enqueue_begin(self, speech: torch.Tensor, speech_lengths, return_intermediate: bool = False) -> dict:
        enc, olens = self.encode(speech, speech_lengths, all_rows=return_intermediate)
        if not hasattr(self.predictor, "forward_begin") or type(self).calc_predictor is not Paraformer.calc_predictor:
            return dict(enc=enc, olens=olens, want=return_intermediate, done=self.calc_predictor(enc, olens))   # (another predictor: its own wait)
        return dict(enc=enc, olens=olens, want=return_intermediate, pred=self.predictor.forward_begin(enc, olens))

    def enqueue_finish(self, half: dict) -> dict:
        enc, olens = half["enc"], half["olens"]
        embeds, token_num, alphas, peaks = half["done"] if "done" in half else self.predictor.forward_finish(half["pred"])
        tok = [int(round(v)) for v in token_num.tolist()]           # pre_token_length.round().long(), model.py:614
        ids = None
        if max(tok) >= 1:                                            # model.py:615-616
            ids, _ = self.decoder.greedy(enc, olens, embeds, tok)
        pending = dict(tok=tok, ids=ids, B=enc.shape[0])
        if ids is not None:
            # the batch's single D2H copy goes on the stream now: collect() then waits for THIS batch only
            pending["ids_host"] = (ids, self.__dict__.setdefault("_host_ring", HostCopyRing()).start(ids))
        if half["want"]:
            pending["extra"] = dict(enc=enc, olens=olens, embeds=embeds, alphas=alphas, peaks=peaks)
            if ids is not None:                                      # what the token timestamps read on the host (_token_timestamps)
                ring = self.__dict__.setdefault("_host_ring", HostCopyRing())
                pending["stamps_host"] = (ring.start(peaks.contiguous()), ring.start(alphas.contiguous()))
        return pending

    _split_enqueue_features = enqueue_features      # inference_begin uses the halves only while no subclass overrides the chain

    def collect(self, pending: dict) -> dict:
        tok, B = pending["tok"], pending["B"]
        raw: List[List[int]] = [[] for _ in range(B)]
        if pending["ids"] is not None:
            # the single D2H copy of the batch: started at enqueue time when this class enqueued these very ids
            early = pending.get("ids_host")
            ids_host = HostCopyRing.wait(early[1]) if early is not None and early[0] is pending["ids"] else pending["ids"].cpu()
            raw = [ids_host[b, : tok[b]].tolist() for b in range(B)]
        drop = (self.sos, self.eos, self.blank_id)
        out = dict(token_num=tok, raw_ids=raw, ids=[[t for t in r if t not in drop] for r in raw])
        out.update(pending.get("extra", {}))
        if "stamps_host" in pending:
            out["peaks_host"], out["alphas_host"] = (HostCopyRing.wait(h).clone() for h in pending["stamps_host"])
        return out

    def recognize_features(self, speech: torch.Tensor, speech_lengths, return_intermediate: bool = False):
        """[B, T, 560] features -> per-utterance token ids (sos/eos/blank removed), all on the current HIP stream."""
        return self.collect(self.enqueue_features(speech, speech_lengths, return_intermediate))

    # ------------------------------------------------------------------------------------------------ beam search
    def init_beam_search(self, **kwargs):
        """model.py:482-532: scorers ctc (weight decoding_ctc_weight) and length_bonus (weight penalty); `lm` / `ngram` have
        no scorer object in the reference either. pre-beam on the full score unless the model's own ctc_weight is 1."""
        from .beam_search import BeamSearchPara
        token_list = kwargs.get("token_list")
        self.beam_search = BeamSearchPara(beam_size=kwargs.get("beam_size", 2), vocab_size=len(token_list), sos=self.sos,
                                          eos=self.eos, ctc_weight=kwargs.get("decoding_ctc_weight", 0.0) if self.ctc is not None else 0.0,
                                          length_bonus_weight=kwargs.get("penalty", 0.0), blank=self.blank_id,
                                          pre_beam=self.ctc_weight != 1.0)

    def recognize_features_beam(self, speech: torch.Tensor, speech_lengths, maxlenratio: float = 0.0,
                                minlenratio: float = 0.0, return_intermediate: bool = False):
        """The beam-search route of `inference` (model.py:596-637): encoder, predictor, decoder LOGITS, then on the device the
        row-wise log-softmax of the decoder scores and of the CTC head; the per-hypothesis bookkeeping runs on the host
        (funasr_amd/beam_search.py) per utterance like the reference's. -> dict(nbest=[[Hypothesis]], token_num, ...)"""
        from . import ops
        enc, olens = self.encode(speech, speech_lengths, all_rows=return_intermediate)
        embeds, token_num, alphas, peaks = self.calc_predictor(enc, olens)
        tok = [int(round(v)) for v in token_num.tolist()]
        B = enc.shape[0]
        out = dict(token_num=tok, nbest=[[] for _ in range(B)])
        if return_intermediate:
            out.update(enc=enc, olens=olens, embeds=embeds, alphas=alphas, peaks=peaks)
        if max(tok) < 1:
            return out
        logits = self._decoder_logits(enc, olens, embeds, tok)
        am = ops.log_softmax(logits.contiguous(), inplace=True).cpu()          # one D2H copy of the batch's scores
        ctc_logp = self.ctc.log_softmax(enc).cpu().numpy() if (self.ctc is not None and self.beam_search.w_ctc != 0) else None
        for i in range(B):
            if tok[i] < 1:
                continue
            lp = ctc_logp[i, : int(olens[i])] if ctc_logp is not None else None
            out["nbest"][i] = self.beam_search(am[i, : tok[i]], lp, maxlenratio=maxlenratio, minlenratio=minlenratio)[: self.nbest]
        return out

    def _decoder_logits(self, enc, olens, embeds, tok) -> torch.Tensor:
        """the decoder scores the beam search ranks (cal_decoder_with_predictor, model.py:326-346); subclasses with other decoder
        inputs (ContextualParaformer: the hotword embeddings) override"""
        return self.decoder(enc, olens, embeds, torch.tensor(tok))[0]

    # ---------------------------------------------------------------------------------------------- AutoModel API
    def _wants_beam(self, kwargs) -> None:
        is_use_ctc = kwargs.get("decoding_ctc_weight", 0.0) > 0.00001 and self.ctc is not None        # model.py:554-562
        is_use_lm = kwargs.get("lm_weight", 0.0) > 0.00001 and kwargs.get("lm_file", None) is not None
        if self.beam_search is None and (is_use_lm or is_use_ctc):
            self.init_beam_search(**kwargs)
            self.nbest = kwargs.get("nbest", 1)

    def _prepare(self, data_in, data_lengths, frontend, kwargs, staged: bool = False):
        """model.py:576-595: waveforms / features -> (speech, speech_lengths, meta_data); `staged`: pinned upload on its own stream"""
        return batch_to_features(data_in, data_lengths, frontend, kwargs,
                                 uploader=self.__dict__.setdefault("_upload", StagedUpload()) if staged else None)

    def inference(self, data_in, data_lengths=None, key: list = None, tokenizer=None, frontend=None, **kwargs):
        self._wants_beam(kwargs)
        speech, speech_lengths, meta_data = self._prepare(data_in, data_lengths, frontend, kwargs, staged=True)
        want_stamps = self._always_timestamps or kwargs.get("pred_timestamp", False)
        if self.beam_search is not None and not self._always_timestamps:
            return self._inference_beam(speech, speech_lengths, key, tokenizer, want_stamps, meta_data, **kwargs)
        res = self.recognize_features(speech, speech_lengths, return_intermediate=want_stamps)
        return self._assemble(res, key, tokenizer, want_stamps, meta_data, kwargs)

    # ---- the same call in three parts, for AutoModel.inference's loop over batches (auto_model.py:790-840 runs them one after the
    #      other: load, features, forward, text -- the GPU idles through every host part). begin(i + 1) | launch(i) | end(i - 1):
    #      the host loads and uploads batch i + 1 and enqueues its encoder while batch i's decoder runs on a second stream and batch
    #      i - 1's ids become text. Records, their order and the ids are those of `inference` (tests/test_parity_gpu.py).
    def inference_begin(self, data_in, data_lengths=None, key: list = None, tokenizer=None, frontend=None, **kwargs):
        """-> a pending object for `inference_launch` / `inference_end`, or None when this call has to take `inference` itself (beam
        search, feature input, no GPU). The plain offline model without timestamps goes in three parts (pf_paraformer_begin /
        _finish: nothing waits for the host before `inference_launch`); timestamps and the other model classes (BiCif / SeACo /
        contextual: module-by-module chains that read the CIF token counts on the way) go in two -- everything enqueued here, text
        in `inference_end` -- which still puts the next batch's loading and upload and the previous batch's text beside GPU work."""
        self._wants_beam(kwargs)
        want_stamps = self._always_timestamps or kwargs.get("pred_timestamp", False)
        if (self.beam_search is not None or kwargs.get("data_type", "sound") == "fbank" or not str(kwargs.get("device", "")).startswith("cuda")
                or not torch.cuda.is_available()):
            return None
        speech, speech_lengths, meta_data = self._prepare(data_in, data_lengths, frontend, kwargs, staged=True)
        pending = dict(key=key, tokenizer=tokenizer, meta_data=meta_data, kwargs=kwargs, want_stamps=want_stamps)
        if self._one_call_ok() and not want_stamps:
            pending["ticket"] = self.begin_features(speech, speech_lengths)
        elif type(self).enqueue_features is getattr(type(self), "_split_enqueue_features", None):
            pending["half"] = self.enqueue_begin(speech, speech_lengths, want_stamps)        # the chain up to its host wait
        else:
            pending["fin"] = self.enqueue_features(speech, speech_lengths, return_intermediate=want_stamps)
        return pending

    def inference_launch(self, pending: dict) -> None:
        if "half" in pending:
            pending["fin"] = self.enqueue_finish(pending.pop("half"))
            return
        if "ticket" not in pending:
            return
        dev = pending["ticket"]["dev"]
        side = self.__dict__.get("_dec_stream")
        if side is None or side.device != dev:
            side = self.__dict__["_dec_stream"] = torch.cuda.Stream(device=dev)
            _lib.load().pf_set_concurrency_guard(1)
        pending["fin"] = self.finish_features(pending.pop("ticket"), stream=side)

    def inference_end(self, pending: dict):
        if "fin" not in pending:
            self.inference_launch(pending)
        res = self.collect(pending.pop("fin"))
        return self._assemble(res, pending["key"], pending["tokenizer"], pending["want_stamps"], pending["meta_data"], pending["kwargs"])

    def _assemble(self, res, key, tokenizer, want_stamps, meta_data, kwargs):
        """model.py:639-697: ids -> records"""
        B = len(res["ids"])
        if key is None:
            key = [f"utt_{i}" for i in range(B)]
        if self._unwrap_key_lists:
            if isinstance(key[0], (list, tuple)):
                key = key[0]
            if len(key) < B:
                key = list(key) * B
        if max(res["token_num"]) < 1:
            return self._nothing_decoded(meta_data)
        ibest_writer = None
        if kwargs.get("output_dir") is not None:                        # model.py:571-575
            if not hasattr(self, "writer"):
                from .datadir_writer import DatadirWriter
                self.writer = DatadirWriter(kwargs.get("output_dir"))
            ibest_writer = self.writer["1best_recog"]
        results = []
        for i in range(B):
            token_int = res["ids"][i]
            if tokenizer is not None:
                token = tokenizer.ids2tokens(token_int)
                text = tokenizer.tokens2text(token)
                if want_stamps:
                    stamps = self._token_timestamps(res, i, token, kwargs)
                    text, stamps = self._postprocess(tokenizer, token, text, stamps)
                    results.append({"key": key[i], "text": text, "timestamp": stamps})
                    if ibest_writer is not None and self._always_timestamps:     # bicif_paraformer/model.py:396
                        ibest_writer["timestamp"][key[i]] = stamps
                else:
                    if not hasattr(tokenizer, "bpemodel"):
                        text, _ = sentence_postprocess(token)
                    results.append({"key": key[i], "text": text})
                if ibest_writer is not None:                             # model.py:688-692
                    ibest_writer["token"][key[i]] = " ".join(token)
                    ibest_writer["text"][key[i]] = text
            else:
                results.append({"key": key[i], "token_int": token_int})
        return results, meta_data

    def _inference_beam(self, speech, speech_lengths, key, tokenizer, want_stamps, meta_data, **kwargs):
        """result assembly of model.py:629-697 for the n-best of the beam search (one result per hypothesis)"""
        res = self.recognize_features_beam(speech, speech_lengths, kwargs.get("maxlenratio", 0.0), kwargs.get("minlenratio", 0.0),
                                           return_intermediate=want_stamps)
        B = len(res["nbest"])
        if key is None:
            key = [f"utt_{i}" for i in range(B)]
        if self._unwrap_key_lists:
            if isinstance(key[0], (list, tuple)):
                key = key[0]
            if len(key) < B:
                key = list(key) * B
        if max(res["token_num"]) < 1:
            return self._nothing_decoded(meta_data)
        drop = (self.eos, self.sos, self.blank_id)
        results = []
        for i in range(B):
            for hyp in res["nbest"][i]:
                token_int = [t for t in hyp.yseq[1:-1] if t not in drop]
                if tokenizer is None:
                    results.append({"key": key[i], "token_int": token_int})        # model.py:693: no score in the record
                    continue
                token = tokenizer.ids2tokens(token_int)
                text = tokenizer.tokens2text(token)
                if want_stamps:
                    stamps = self._token_timestamps(res, i, token, kwargs)
                    text, stamps = self._postprocess(tokenizer, token, text, stamps)
                    results.append({"key": key[i], "text": text, "timestamp": stamps})
                else:
                    if not hasattr(tokenizer, "bpemodel"):
                        text, _ = sentence_postprocess(token)
                    results.append({"key": key[i], "text": text})
        return results, meta_data

    def _token_timestamps(self, res: dict, i: int, token, kwargs):
        # model.py:668-681. The reference hands (cif_peak, alphas) to the (us_alphas, us_peaks) parameters of
        # ts_prediction_lfr6_standard in THAT order; kept, so that the timestamps are the reference's.
        if "peaks_host" not in res:                       # one D2H copy per batch, not two per utterance
            res["peaks_host"], res["alphas_host"] = res["peaks"].cpu(), res["alphas"].cpu()
        return cif_token_spans(res["peaks_host"][i], res["alphas_host"][i], list(token),
                               vad_offset=kwargs.get("begin_time", 0), upsample_rate=1)

    def _postprocess(self, tokenizer, token, text, stamps):
        if not hasattr(tokenizer, "bpemodel"):
            text, stamps, _ = sentence_postprocess(token, stamps)
        return text, stamps

    def forward(self, *args, **kwargs):  # pragma: no cover
        raise NotImplementedError("training forward() is out of scope; use inference()/recognize_features()")
