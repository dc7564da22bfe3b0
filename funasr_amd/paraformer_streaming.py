"""Streaming (chunked online) Paraformer on gfx950.

Host-side mirrors of
  * `WavFrontendOnline`   funasr/frontends/wav_frontend.py:261-662 (`frontend_classes["WavFrontendOnline"]`): the
    stateful bookkeeping (left-over samples, LFR splice cache, :395-421,546-591) stays on the host, the arithmetic
    (log-mel, LFR gather, CMVN) runs in the kernels of csrc/frontend.hip through `pf_frontend_fbank` /
    `pf_frontend_lfr_cmvn`; the buffers live in HBM;
  * `SANMEncoderChunkOpt` funasr/models/scama/encoder.py:187-549 (`encoder_classes["SANMEncoderChunkOpt"]`): same
    parameters / state_dict keys as SANMEncoder; its `forward_chunk` state (overlap window, K/V look-back) is owned by
    the stream handle;
  * `ParaformerStreaming` funasr/models/paraformer_streaming/model.py:31-763 (`model_classes["ParaformerStreaming"]`):
    `init_cache`, `generate_chunk`, `inference(data_in, ..., cache=..., is_final=..., chunk_size=[0,10,5],
    encoder_chunk_look_back=4, decoder_chunk_look_back=1)` with the reference's chunk loop (9600-sample strides,
    left-over samples in cache["prev_samples"], tail chunk < 960 samples re-feeding the cached window, :705-752).
One chunk = one `pf_stream_step`: encoder window + K/V rings, CIF with carried remainder, decoder with FSMN /
cross-attention caches and the fused vocabulary arg-max, captured once in a hipGraph and replayed.
`StreamBatch` advances S independent streams in lock-step (an extension: the reference is batch-1 only).
"""
from __future__ import annotations

import logging

import ctypes as C
import time
from typing import List, Optional, Sequence

import numpy as np
import torch
import torch.nn as nn

from . import _lib
from .audio import load_audio_list
from .hip_module import host_i32, stream_ptr
from .paraformer import Paraformer
from .register import tables
from .sanm_encoder import SANMEncoder, sinusoidal_position_table
from .tokenizer import sentence_postprocess
from .wav_frontend import WavFrontend


# ================================================================================================== frontend
@tables.register("frontend_classes", "WavFrontendOnline")
class WavFrontendOnline(WavFrontend):
    """forward(input [1, n], input_lengths, cache=<dict>, is_final=bool) -> (feats [1, t, n_mels*lfr_m] | empty, lens)"""

    def __init__(self, *args, **kwargs):
        super().__init__(*args, **kwargs)
        # the online class stores `snip_edges` and never passes it to kaldi.fbank (wav_frontend.py:316,441-450): frames always lie
        # inside the samples seen so far (compute_frame_num, :393-399), whatever the option says; `window` IS passed (:449)
        self.snip_edges_option, self.snip_edges = self.snip_edges, True

    def init_cache(self, cache: dict = None):
        if cache is None:
            cache = {}
        cache["input_cache"] = None           # un-framed samples (device tensor), wav_frontend.py:414-421
        cache["lfr_splice_cache"] = None      # un-consumed fbank frames [k, n_mels] (device), :546-550,380
        return cache

    def _lfr_cmvn(self, lib, h, frames: torch.Tensor, is_final: bool):
        """WavFrontendOnline.apply_lfr (:349-380) row/splice arithmetic on the host, gather + CMVN on the device."""
        T = frames.shape[0]
        m, n = self.lfr_m, self.lfr_n
        t_lfr = int(np.ceil((T - (m - 1) // 2) / n))
        last_idx = (T - m) // n + 1
        rows = max(t_lfr if is_final else last_idx, 0)
        splice_idx = min(T - 1, rows * n)
        out = torch.empty(rows, self.output_size(), device=frames.device, dtype=torch.float32)
        if rows > 0:
            _lib.check(lib.pf_frontend_lfr_cmvn(h, frames.data_ptr(), T, rows, out.data_ptr(), stream_ptr()),
                       "pf_frontend_lfr_cmvn")
        return out, frames[splice_idx:].clone()

    def forward(self, input: torch.Tensor, input_lengths=None, **kwargs):
        is_final = kwargs.get("is_final", False)
        cache = kwargs.get("cache", {})
        if "input_cache" not in cache:
            self.init_cache(cache)
        if input.dim() == 2:
            if input.shape[0] != 1:
                raise ValueError("we support to extract feature online only when the batch size is equal to 1 now")
            input = input[0]
        dev = self._target_device(input)
        lib, h = self._ensure_handle(dev)
        x = input.to(device=dev, dtype=torch.float32)
        if cache["input_cache"] is not None:
            x = torch.cat((cache["input_cache"], x))
        x = x.contiguous()
        n = x.numel()
        win, hop = int(self.fs * self.frame_length * 0.001), int(self.fs * self.frame_shift * 0.001)
        frame_num = int((n - win) / hop + 1) if n >= win else 0
        cache["input_cache"] = x[frame_num * hop:].clone()
        empty = torch.empty(0)
        with torch.cuda.device(dev):
            if frame_num:
                fb = torch.empty(frame_num, self.n_mels, device=dev, dtype=torch.float32)
                _lib.check(lib.pf_frontend_fbank(h, x.data_ptr(), n, fb.data_ptr(), stream_ptr()), "pf_frontend_fbank")
                if cache["lfr_splice_cache"] is None:
                    cache["lfr_splice_cache"] = fb[0:1].repeat((self.lfr_m - 1) // 2, 1)
                if frame_num + cache["lfr_splice_cache"].shape[0] >= self.lfr_m:
                    frames = torch.cat((cache["lfr_splice_cache"], fb), 0).contiguous()
                    out, cache["lfr_splice_cache"] = self._lfr_cmvn(lib, h, frames, is_final)
                else:
                    cache["lfr_splice_cache"] = torch.cat((cache["lfr_splice_cache"], fb), 0)
                    return empty, torch.tensor([frame_num], dtype=torch.int32)
            else:
                if not is_final or cache["lfr_splice_cache"] is None:
                    return empty, torch.zeros(1, dtype=torch.int32)
                out, cache["lfr_splice_cache"] = self._lfr_cmvn(lib, h, cache["lfr_splice_cache"].contiguous(), True)
        return out[None], torch.tensor([out.shape[0]], dtype=torch.int32)


# =================================================================================================== encoder
@tables.register("encoder_classes", "SANMEncoderChunkOpt")
class SANMEncoderChunkOpt(SANMEncoder):
    def __init__(self, input_size: int, output_size: int = 256, attention_heads: int = 4, linear_units: int = 2048,
                 num_blocks: int = 6, input_layer: Optional[str] = "pe_online", chunk_size=(16,), stride=(10,),
                 pad_left=(0,), encoder_att_look_back_factor=(1,), decoder_att_look_back_factor=(1,), **kwargs):
        if input_layer not in ("pe", "pe_online"):
            raise NotImplementedError("SANMEncoderChunkOpt(HIP): input_layer must be 'pe' or 'pe_online'")
        super().__init__(input_size, output_size, attention_heads, linear_units, num_blocks, input_layer="pe", **kwargs)
        # training-time chunk masks (overlap_chunk_cls, scama/chunk_utilis.py) are out of scope; kept for config parity
        self.chunk_size, self.stride, self.pad_left = chunk_size, stride, pad_left
        self.encoder_att_look_back_factor = encoder_att_look_back_factor
        self.decoder_att_look_back_factor = decoder_att_look_back_factor

    def forward_chunk(self, *a, **k):
        raise RuntimeError("the chunk state lives in the stream handle: use ParaformerStreaming.inference / StreamBatch")


# ====================================================================================================== stream
class StreamBatch:
    """S independent streams advanced in lock-step over one (encoder, predictor, decoder) triple; all caches in HBM.

    `precision`: "fp32" = the step's GEMMs on the fp32 kernels (weight-streaming GEMM for a handful of rows: the latency path
    of one or a few streams); "f16x2" = on the fp16 matrix cores with two-plane operands and fp32 results (the offline
    default's arithmetic: the throughput path of many lock-step streams, `pf_stream_set_option("gemm_mode", 3)`). None picks
    "f16x2" from AUTO_F16X2_MIN_STREAMS streams on, else "fp32": measured step times fp32 / f16x2 at chunk [0, 10, 5]
    (profiles/r03w_bench_streaming.jsonl / r03s_bench_streaming_splitk.jsonl): S = 8 6.6 / 8.1 ms, 32 7.8 / 8.2, 64 10.7 / 8.5,
    128 17.8 / 9.7, 256 32.8 / 13.8 -- the two cross between 32 and 64 streams. The arithmetic is fixed per StreamBatch, so a
    stream's result never depends on how many chunks its neighbours bring."""

    AUTO_F16X2_MIN_STREAMS: Optional[int] = 48
    _announced: set = set()

    def __init__(self, model: "ParaformerStreaming", n_streams: int = 1, chunk_size: Sequence[int] = (0, 10, 5),
                 encoder_chunk_look_back: int = 4, decoder_chunk_look_back: int = 1, max_frames: int = None,
                 max_tokens: int = None, use_graph: bool = True, pe_rows: int = 8192, precision: Optional[str] = None):
        if len(chunk_size) != 3 or chunk_size[0] + chunk_size[2] == 0:
            # scama/encoder.py:480-494 keeps xs_pad[:, -(chunk_size[0] + chunk_size[2]):] as the overlap window: with both 0 that
            # slice is [-0:] = the whole window, i.e. the reference's window grows by every chunk and its CIF mask keeps decoding
            # the first chunk_size[1] frames (tests/test_oracle_streaming.py pins that on the reference's own session)
            raise ValueError("StreamBatch: chunk_size[0] + chunk_size[2] must be > 0 (the reference's overlap window is the whole "
                             "history when both are 0: a degenerate session that is not reproduced)")
        if precision is None:
            auto = self.AUTO_F16X2_MIN_STREAMS
            precision = "f16x2" if (auto is not None and n_streams >= auto) else "fp32"
            # the two arithmetics are both fp32-class but not bit-identical (CIF sums sit ~1e-5 from a threshold about once per
            # thousand clips): say which one a deployment got, once per (count, choice), so that results are
            # reproducible by pinning `precision=` (ParaformerStreaming.inference / AutoModel.generate: stream_precision="fp32" | "f16x2")
            if (n_streams, precision) not in StreamBatch._announced:
                StreamBatch._announced.add((n_streams, precision))
                logging.getLogger("funasr_amd").info("StreamBatch: %d stream(s) -> precision %r chosen by stream count "
                                                      "(pass precision= to pin it)", n_streams, precision)
        if precision not in ("fp32", "f16x2"):
            raise ValueError(f"StreamBatch: precision must be 'fp32' or 'f16x2', got {precision!r}")
        self.precision = precision
        self.model = model
        self.S, self.chunk_size = n_streams, list(chunk_size)
        # a step brings at most chunk_cur new frames plus the final flush of the look-ahead; CIF fires at most once per
        # weighted frame (the window minus its chunk_left zeroed frames) plus the carried remainder and the tail weight
        self.max_frames = max(max_frames or 0, chunk_size[1] + chunk_size[2] + 1)
        self.max_tokens = max_tokens if max_tokens is not None else chunk_size[2] + self.max_frames + 2
        lib, he = model.encoder._ensure_handle()
        _, hp = model.predictor._ensure_handle()
        _, hd = model.decoder._ensure_handle()
        self.dev = model.encoder._handle_device
        self.lib = lib
        self._parts = ((model.encoder, he), (model.predictor, hp), (model.decoder, hd))
        cfg = _lib.pf_stream_config(n_streams, chunk_size[0], chunk_size[1], chunk_size[2], encoder_chunk_look_back,
                                    decoder_chunk_look_back, self.max_frames, self.max_tokens, int(use_graph))
        with torch.cuda.device(self.dev):
            self._h = _lib.check_handle(lib.pf_stream_create(he, hp, hd, C.byref(cfg)), "pf_stream_create")
            pe = sinusoidal_position_table(pe_rows, model.encoder._input_size).contiguous()
            _lib.check(lib.pf_stream_set_pe(self._h, pe.data_ptr(), pe_rows), "pf_stream_set_pe")
            if precision == "f16x2":
                _lib.check(lib.pf_stream_set_option(self._h, b"gemm_mode", 3), "pf_stream_set_option")
        self.pe_rows = pe_rows
        self.start_idx = 0
        self._inflight = None
        self.keep = chunk_size[0] + chunk_size[2]

    def set_option(self, key: str, value: int):
        """pf_stream_set_option: "gemm_mode" 0 | 3; the fp32 step's launch fusions (A/B switches, defaults on): "ln_carry" 0 | 1 | 2
        (LayerNorms carried by the small-M GEMMs: never / steps of <= 32 rows / always), "fsmn_rides", "kv_batched"; "wide_k" (off)"""
        with torch.cuda.device(self.dev):
            _lib.check(self.lib.pf_stream_set_option(self._h, key.encode(), int(value)), "pf_stream_set_option")

    def reset(self):
        _lib.check(self.lib.pf_stream_reset(self._h, None), "pf_stream_reset")
        self.start_idx = 0
        self._inflight = None                            # (the C side drops a step in flight with the session it belonged to)

    def _grow_pe(self, need: int):
        if need <= self.pe_rows:
            return
        rows = max(need, 2 * self.pe_rows)
        pe = sinusoidal_position_table(rows, self.model.encoder._input_size).contiguous()
        _lib.check(self.lib.pf_stream_set_pe(self._h, pe.data_ptr(), rows), "pf_stream_set_pe")
        self.pe_rows = rows

    def step(self, feats: Optional[torch.Tensor], is_final: bool = False, tail_chunk: bool = False,
             return_enc: bool = False):
        """feats [S, n, 560] (device) or None for a tail chunk -> (list of raw id lists per stream, enc [S, W, 512]?)"""
        self.step_begin(feats, is_final, tail_chunk, return_enc)
        return self.step_end()

    def step_begin(self, feats: Optional[torch.Tensor], is_final: bool = False, tail_chunk: bool = False, return_enc: bool = False):
        """Enqueue the step on the handle's own HIP stream and return at once (pf_stream_step_begin); step_end() collects it.
        StreamBatches over DIFFERENT model objects may each have a step in flight: steps of a few dozen streams overlap on the chip."""
        n = 0 if tail_chunk else int(feats.shape[1])
        self._grow_pe(self.start_idx + n)
        for m, h in self._parts:
            if m._dirty:                       # weights changed on the live modules (load_state_dict, mark_dirty): push them
                if m._ensure_handle()[1] != h:
                    raise RuntimeError("StreamBatch: a module's handle was re-created (moved to another device?); build a new StreamBatch")
        W = self.keep + n
        enc = torch.empty(self.S, W, 512, device=self.dev, dtype=torch.float32) if return_enc else None
        f = None
        if not tail_chunk:
            f = feats.to(device=self.dev, dtype=torch.float32).contiguous()
            if f.shape[0] != self.S:
                raise ValueError(f"expected {self.S} streams, got {f.shape[0]}")
        with torch.cuda.device(self.dev):
            _lib.check(self.lib.pf_stream_step_begin(self._h, f.data_ptr() if f is not None else None, n, int(is_final),
                                                     int(tail_chunk), enc.data_ptr() if enc is not None else None,
                                                     stream_ptr()), "pf_stream_step_begin")
        self.start_idx += self.keep if tail_chunk else n
        self._inflight = (f, enc, return_enc)            # (the features stay alive until the step has read them)

    def step_end(self):
        if getattr(self, "_inflight", None) is None:
            raise RuntimeError("step_end(): no step in flight (call step_begin() first; reset() drops a pending step)")
        f, enc, return_enc = self._inflight
        self._inflight = None
        ids = (C.c_int32 * (self.S * self.max_tokens))()
        cnt = (C.c_int32 * self.S)()
        with torch.cuda.device(self.dev):
            _lib.check(self.lib.pf_stream_step_end(self._h, ids, cnt), "pf_stream_step_end")
        out = [[int(ids[s * self.max_tokens + k]) for k in range(int(cnt[s]))] for s in range(self.S)]
        return (out, enc) if return_enc else out

    def peek(self):
        a = (C.c_float * self.S)()
        h = (C.c_float * (self.S * 512))()
        si = C.c_int32(0)
        _lib.check(self.lib.pf_stream_peek(self._h, a, h, C.byref(si)), "pf_stream_peek")
        return dict(cif_alphas=list(a), cif_hidden=torch.tensor(list(h)).view(self.S, 512), start_idx=int(si.value))

    def close(self):
        h = self.__dict__.get("_h")
        if h:
            try:
                self.lib.pf_stream_destroy(h)
            except Exception:
                pass
            self.__dict__["_h"] = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


# ======================================================================================================= model
@tables.register("model_classes", "ParaformerStreaming")
class ParaformerStreaming(Paraformer):
    def __init__(self, *args, **kwargs):
        super().__init__(*args, **kwargs)
        self.scama_mask = None

    @classmethod
    def from_config(cls, cfg: dict) -> "ParaformerStreaming":
        ec = dict(cfg["encoder"])
        input_size = ec.pop("input_size")
        dc = dict(cfg["decoder"])
        vocab = dc.pop("vocab_size")
        dc.pop("encoder_output_size", None)
        return cls(encoder="SANMEncoderChunkOpt", encoder_conf=dict(ec, input_layer="pe_online"),
                   decoder="ParaformerSANMDecoder", decoder_conf=dc, predictor="CifPredictorV2",
                   predictor_conf=dict(cfg["predictor"]), ctc_weight=0.0, input_size=input_size, vocab_size=vocab)

    def init_cache(self, cache: dict = None, **kwargs):
        """paraformer_streaming/model.py:511-550: the tensors of the reference's cache dict live in the stream handle."""
        if cache is None:
            cache = {}
        old = cache.get("_stream")
        if old is not None:
            old.close()
        cache["_stream"] = StreamBatch(self, 1, kwargs.get("chunk_size", [0, 10, 5]),
                                       kwargs.get("encoder_chunk_look_back", 0), kwargs.get("decoder_chunk_look_back", 0),
                                       use_graph=kwargs.get("use_graph", True), precision=kwargs.get("stream_precision"))
        cache["encoder"] = {"tail_chunk": False, "chunk_size": kwargs.get("chunk_size", [0, 10, 5])}
        cache["decoder"] = {}
        cache["frontend"] = {}
        cache["prev_samples"] = torch.empty(0)
        return cache

    def generate_chunk(self, speech, speech_lengths=None, key: list = None, tokenizer=None, frontend=None, **kwargs):
        """:552-648, greedy. Returns the chunk's tokens (or ids when tokenizer is None)."""
        cache = kwargs.get("cache", {})
        sb: StreamBatch = cache["_stream"]
        tail = bool(cache["encoder"].get("tail_chunk", False))
        raw = sb.step(None if tail else speech, is_final=kwargs.get("is_final", False), tail_chunk=tail)[0]
        ids = [t for t in raw if t not in (self.sos, self.eos, self.blank_id)]
        return tokenizer.ids2tokens(ids) if tokenizer is not None else ids

    def inference_begin(self, *args, **kwargs):
        """the streaming model decodes chunk by chunk against a session cache: its calls cannot be taken apart and overlapped like
        the offline model's (Paraformer.inference_begin) -- AutoModel.inference then runs the plain loop"""
        return None

    def inference(self, data_in, data_lengths=None, key: list = None, tokenizer=None, frontend=None, cache: dict = None,
                  **kwargs):
        if kwargs.get("decoding_ctc_weight", 0.0) > 1e-5 or (kwargs.get("lm_weight", 0.0) > 1e-5 and kwargs.get("lm_file")):
            raise NotImplementedError("beam search with CTC/LM rescoring is outside the greedy hot path")
        if cache is None:
            cache = {}
        if len(cache) == 0:
            self.init_cache(cache, **kwargs)
        meta_data = {}
        chunk_size = kwargs.get("chunk_size", [0, 10, 5])
        stride = int(chunk_size[1] * 960)                                   # 600 ms (:688-689)
        t1 = time.perf_counter()
        is_final = bool(kwargs.get("is_final", False))
        first = data_in[0] if isinstance(data_in, (list, tuple)) and len(data_in) else data_in
        if isinstance(first, str) or hasattr(first, "read"):
            # a file (also inside the list AutoModel.inference always passes: load_audio_text_image_video forwards `cache`
            # through its list recursion, :692-701) is a whole utterance: run the look-ahead / tail chunk now
            is_final = True
        audio_list = load_audio_list(data_in, fs=frontend.fs, audio_fs=kwargs.get("fs", 16000))
        t2 = time.perf_counter()
        meta_data["load_data"] = f"{t2 - t1:0.3f}"
        assert len(audio_list) == 1, "batch_size must be set 1"
        audio = torch.cat((cache["prev_samples"], audio_list[0].cpu()))
        n = int(len(audio) // stride + int(is_final))
        m = int(len(audio) % stride * (1 - int(is_final)))
        tokens = []
        kw = {k: v for k, v in kwargs.items() if k not in ("is_final", "cache")}
        for i in range(n):
            fin = is_final and i == n - 1
            piece = audio[i * stride:(i + 1) * stride]
            if fin and len(piece) < 960:
                cache["encoder"]["tail_chunk"] = True
                speech, frames = None, cache["_stream"].keep
            else:
                speech, lens = frontend(piece[None], [len(piece)], cache=cache["frontend"], is_final=fin)
                frames = int(lens.sum())
            meta_data["extract_feat"] = f"{time.perf_counter() - t2:0.3f}"
            meta_data["batch_data_time"] = frames * frontend.frame_shift * frontend.lfr_n / 1000
            if speech is not None and speech.numel() == 0:
                continue
            tokens.extend(self.generate_chunk(speech, None, key=key, tokenizer=tokenizer, cache=cache, frontend=frontend,
                                              is_final=fin, **kw))
        cache["prev_samples"] = audio[-m:] if m > 0 else torch.empty(0)
        if is_final:
            self.init_cache(cache, **kwargs)
        if key is None:
            key = ["utt_0"]
        if tokenizer is not None:
            text, _ = sentence_postprocess(tokens)
            return [{"key": key[0], "text": text}], meta_data
        return [{"key": key[0], "token_int": tokens}], meta_data
