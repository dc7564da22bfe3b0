"""FSMN-VAD on gfx950: the network on the GPU, the decision logic on the host.

Host-side mirrors, registered under the reference's names:
  * `FSMN` (encoder_classes; funasr/models/fsmn_vad_streaming/encoder.py:288-378): holds the parameters under the
    reference's state_dict keys; `forward(feats, cache)` returns the softmax over the output pdfs like the reference,
    `silence_posterior(feats, cache, sil_pdf_ids)` the only thing the decision logic reads. Dense layers run on the
    fp32 GEMM kernels, the memory blocks / softmax / frame energies in csrc/vad.hip. The left-context cache of the memory
    blocks lives in HBM (`cache["fsmn_ctx"]`, [1, layers, (lorder-1)*lstride, proj_dim]).
  * `FsmnVADStreaming` (model_classes; funasr/models/fsmn_vad_streaming/model.py:367-1115): `inference()` with the
    reference's chunking (`chunk_size` ms per block, default 60 s) and its dynamic end-silence schedule for long
    recordings (:1005-1019); segments come from the decision logic of `vad_decision` (native host code, pinned to the
    reference's state machine). Output `[{"key": ..., "value": [[beg_ms, end_ms], ...]}]`, the contract
    `AutoModel.inference_with_vad` consumes.
For a whole recording the network runs ONCE over all frames (fbank / LFR(5,1) / CMVN through the offline frontend kernel,
then the FSMN): frame t of the score stream is fbank frame t and its energy is that of samples [160 t, 160 t + 400), the
span the reference's online frontend reports as `aligned_waveforms` (wav_frontend.py:606-640); only the decision logic
is fed block by block, which is where chunking is visible in the reference (per-block schedule, final frame).
"""
from __future__ import annotations

import time
from typing import Any, Dict, List, Optional

import torch

from . import _lib
from .audio import load_audio_list
from .hip_module import Holder, HipModule, HostCopyRing, ParamHolder, StagedUpload, linear, stream_ptr
from .register import tables
from .vad_decision import IN_SPEECH, NativeVadDecision, VadOptions

# (speech accumulated in ms, end-silence in ms): funasr/models/fsmn_vad_streaming/model.py:22-41
STREAMING_SILENCE_SCHEDULE = [(5000, 2000), (10000, 1500), (15000, 1000), (30000, 800), (45000, 400), (float("inf"), 100)]
DEFAULT_SILENCE_SCHEDULE = [(10000, 2000), (20000, 1000), (30000, 800), (40000, 600), (50000, 400), (60000, 200),
                            (float("inf"), 100)]


def _affine(out_f, in_f, bias=True):
    h = Holder()
    h.linear = linear(out_f, in_f, bias)
    return h


@tables.register("encoder_classes", "FSMN")
class FSMN(HipModule):
    _prefix = "pf_vad"

    def __init__(self, input_dim: int, input_affine_dim: int, fsmn_layers: int, linear_dim: int, proj_dim: int, lorder: int,
                 rorder: int, lstride: int, rstride: int, output_affine_dim: int, output_dim: int, use_softmax: bool = True,
                 **kwargs):
        super().__init__()
        if rorder != 0:
            raise NotImplementedError("FSMN(HIP): only the uni-directional memory (rorder 0) of fsmn-vad is built")
        if not use_softmax:
            raise NotImplementedError("FSMN(HIP): the VAD head is the softmax posterior")
        self.cfg = dict(input_dim=input_dim, input_affine_dim=input_affine_dim, fsmn_layers=fsmn_layers, linear_dim=linear_dim,
                        proj_dim=proj_dim, lorder=lorder, rorder=rorder, lstride=lstride, rstride=rstride,
                        output_affine_dim=output_affine_dim, output_dim=output_dim)
        self.input_dim, self.output_dim = input_dim, output_dim
        self.in_linear1 = _affine(input_affine_dim, input_dim)
        self.in_linear2 = _affine(linear_dim, input_affine_dim)
        blocks = []
        for _ in range(fsmn_layers):
            b = Holder()
            b.linear = _affine(proj_dim, linear_dim, bias=False)
            b.fsmn_block = Holder()
            b.fsmn_block.conv_left = ParamHolder((proj_dim, 1, lorder, 1))
            b.affine = _affine(linear_dim, proj_dim)
            blocks.append(b)
        self.fsmn = torch.nn.ModuleList(blocks)
        self.out_linear1 = _affine(output_affine_dim, linear_dim)
        self.out_linear2 = _affine(output_dim, output_affine_dim)

    def output_size(self) -> int:
        return self.output_dim

    def _make_config(self):
        c = self.cfg
        return _lib.pf_vad_config(c["input_dim"], c["input_affine_dim"], c["fsmn_layers"], c["linear_dim"], c["proj_dim"],
                                  c["lorder"], c["rorder"], c["lstride"], c["rstride"], c["output_affine_dim"], c["output_dim"])

    def _context(self, cache: Optional[dict], dev) -> Optional[torch.Tensor]:
        if cache is None:
            return None
        if "fsmn_ctx" not in cache:
            c = self.cfg
            cache["fsmn_ctx"] = torch.zeros(1, c["fsmn_layers"], (c["lorder"] - 1) * c["lstride"], c["proj_dim"], device=dev)
        return cache["fsmn_ctx"]

    def _run(self, feats: torch.Tensor, cache: Optional[dict], sil_pdf_ids, want_probs: bool):
        lib, h = self._ensure_handle()
        dev = self._handle_device
        x = feats.to(device=dev, dtype=torch.float32).contiguous()
        B, T, D = x.shape
        if D != self.input_dim:
            raise ValueError(f"FSMN: expected {self.input_dim}-dim features, got {D}")
        if cache is not None and B != 1:
            raise ValueError("FSMN: the streaming cache is per stream (batch 1), like the reference")
        ctx = self._context(cache, dev)
        p_sil = torch.empty(B, T, device=dev, dtype=torch.float32)
        probs = torch.empty(B, T, self.output_dim, device=dev, dtype=torch.float32) if want_probs else None
        ids = (_lib.C.c_int32 * len(sil_pdf_ids))(*[int(i) for i in sil_pdf_ids])
        with torch.cuda.device(dev):
            _lib.check(lib.pf_vad_forward(h, x.data_ptr(), B, T, ctx.data_ptr() if ctx is not None else None, ids,
                                          len(sil_pdf_ids), p_sil.data_ptr(), probs.data_ptr() if want_probs else None,
                                          1 if B * T <= 64 else 0, stream_ptr()), "pf_vad_forward")
        return p_sil, probs

    def forward(self, input: torch.Tensor, cache: Optional[Dict[str, torch.Tensor]] = None):
        return self._run(input, cache, [0], True)[1]

    def silence_posterior(self, feats: torch.Tensor, cache: Optional[dict] = None, sil_pdf_ids=(0,)) -> torch.Tensor:
        return self._run(feats, cache, list(sil_pdf_ids), False)[0]


def frame_decibel(wav: torch.Tensor, n_frames: int, frame_len: int = 400, frame_shift: int = 160) -> torch.Tensor:
    """10 log10(frame energy + 1e-6) of `n_frames` frames of a device waveform (ComputeDecibel, model.py:513-530)."""
    lib = _lib.load()
    if not wav.is_cuda:
        raise RuntimeError("frame_decibel: expected the waveform in GPU memory (there is no CPU path)")
    w = wav.to(torch.float32).contiguous().view(-1)
    if (n_frames - 1) * frame_shift + frame_len > w.numel():
        raise RuntimeError("VAD score frames and waveform samples are not aligned")
    out = torch.empty(n_frames, device=w.device, dtype=torch.float32)
    with torch.cuda.device(w.device):
        _lib.check(lib.pf_vad_frame_decibel(w.data_ptr(), w.numel(), n_frames, frame_len, frame_shift, out.data_ptr(), stream_ptr()),
                   "pf_vad_frame_decibel")
    return out


@tables.register("model_classes", "FsmnVADStreaming")
class FsmnVADStreaming(torch.nn.Module):
    def __init__(self, encoder: str = None, encoder_conf: Optional[Dict] = None, vad_post_args: Dict[str, Any] = None, **kwargs):
        super().__init__()
        self.vad_opts = VadOptions(**kwargs)
        enc_cls = tables.encoder_classes.get(encoder)
        if enc_cls is None:
            raise KeyError(f"encoder {encoder!r} is not registered: {sorted(tables.encoder_classes)}")
        self.encoder = enc_cls(**(encoder_conf or {}))
        self.encoder_conf = encoder_conf or {}

    # ----------------------------------------------------------------------------------------------------- cache
    def init_cache(self, cache: dict = None, **kwargs):
        cache = {} if cache is None else cache
        cache.clear()
        if kwargs.get("max_end_silence_time") is not None:
            self.vad_opts.max_end_silence_time = kwargs["max_end_silence_time"]
        cache["frontend"] = {}
        cache["encoder"] = {}
        cache["prev_samples"] = torch.empty(0)
        cache["decision"] = NativeVadDecision(self.vad_opts, speech_noise_thres=kwargs.get("speech_noise_thres"))
        cache["stats"] = cache["decision"]     # the reference's key: wrappers tune .speech_noise_thres / .max_end_sil_frame_cnt_thresh
        cache["frames_done"] = 0
        cache["wave"], cache["wave_start"] = None, 0   # waveform history behind the frame energies (streaming input)
        return cache

    # ------------------------------------------------------------------------------------------------- one block
    def _scores(self, frontend, wav: torch.Tensor, enc_cache: Optional[dict]):
        """whole waveform -> (silence posterior [T], decibel [T]) on the host; T = number of fbank frames"""
        dev = next(self.encoder.parameters()).device
        w = wav.to(dev)
        if hasattr(frontend, "init_cache"):                       # WavFrontendOnline: one final call flushes every frame
            feats, flens = frontend(w[None], [int(w.numel())], cache={}, is_final=True)
        else:
            feats, flens = frontend(w[None], [int(w.numel())])
        T = int(flens[0]) if feats.numel() else 0
        if T <= 0:
            import numpy as np
            return np.zeros(0, np.float32), np.zeros(0, np.float32)
        p_sil = self.encoder.silence_posterior(feats[:, :T], enc_cache, self.vad_opts.sil_pdf_ids)[0]
        o = self.vad_opts
        db = frame_decibel(w, T, int(o.frame_length_ms * o.sample_rate / 1000), int(o.frame_in_ms * o.sample_rate / 1000))
        return p_sil.cpu().numpy(), db.cpu().numpy()

    def _scores_enqueue(self, frontend, wav: torch.Tensor):
        """`_scores` without its wait: upload (pinned, on the upload stream), features, network and frame energies ENQUEUED, the two
        score vectors on their way to pinned host memory -> handles for `_scores_wait`, or None for a recording without a frame"""
        dev = next(self.encoder.parameters()).device
        if wav.device.type == "cpu":
            w = self.__dict__.setdefault("_upload", StagedUpload())([wav], dev)[0][0]
        else:
            w = wav.to(dev)
        if hasattr(frontend, "init_cache"):
            feats, flens = frontend(w[None], [int(w.numel())], cache={}, is_final=True)
        else:
            feats, flens = frontend(w[None], [int(w.numel())])
        T = int(flens[0]) if feats.numel() else 0
        if T <= 0:
            return None
        p_sil = self.encoder.silence_posterior(feats[:, :T], None, self.vad_opts.sil_pdf_ids)[0]
        o = self.vad_opts
        db = frame_decibel(w, T, int(o.frame_length_ms * o.sample_rate / 1000), int(o.frame_in_ms * o.sample_rate / 1000))
        ring = self.__dict__.setdefault("_host_ring", HostCopyRing())
        return ring.start(p_sil.contiguous()), ring.start(db.contiguous())

    @staticmethod
    def _scores_wait(handles):
        import numpy as np
        if handles is None:
            return np.zeros(0, np.float32), np.zeros(0, np.float32)
        return tuple(np.array(HostCopyRing.wait(h).numpy(), copy=True) for h in handles)

    def inference(self, data_in, data_lengths=None, key: list = None, tokenizer=None, frontend=None, cache: dict = None,
                  **kwargs):
        if frontend is None:
            raise ValueError("FsmnVADStreaming.inference needs the WavFrontend of the model directory (lfr_m 5, lfr_n 1)")
        if frontend.lfr_n != 1:
            raise ValueError("FsmnVADStreaming: one score per 10 ms frame (frontend lfr_n must be 1)")
        cache = {} if cache is None else cache
        if len(cache) == 0:
            self.init_cache(cache, **kwargs)
        meta: Dict[str, Any] = {}
        chunk_ms = kwargs.get("chunk_size", 60000)
        streaming_input = kwargs.get("is_streaming_input", False) if chunk_ms >= 15000 else kwargs.get("is_streaming_input", True)
        is_final = kwargs.get("is_final", False) if streaming_input else kwargs.get("is_final", True)
        t1 = time.perf_counter()
        audio = load_audio_list(data_in if isinstance(data_in, (list, tuple)) else [data_in], fs=frontend.fs,
                                audio_fs=kwargs.get("fs", 16000))
        if isinstance(data_in, (list, tuple)) and len(data_in) and isinstance(data_in[0], str):
            is_final, streaming_input = True, False                               # files are complete recordings
        meta["load_data"] = f"{time.perf_counter() - t1:0.3f}"
        k0 = key[0] if key else ""
        if len(audio) == 0 or audio[0].numel() == 0:
            return [{"key": k0, "value": []}], meta
        if len(audio) != 1:
            raise AssertionError("batch_size must be set 1")
        if streaming_input or not is_final or len(cache.get("prev_samples", ())) or cache.get("frames_done", 0):
            rest = {k: v for k, v in kwargs.items() if k not in ("is_final", "is_streaming_input", "chunk_size")}
            return self._inference_chunked(audio[0], k0, frontend, cache, chunk_ms, is_final, streaming_input, meta, **rest)
        wav = audio[0]
        t2 = time.perf_counter()
        p_sil, db = self._scores(frontend, wav, None)
        meta["extract_feat"] = f"{time.perf_counter() - t2:0.3f}"
        return self._decide(p_sil, db, int(wav.numel()), k0, frontend, cache, chunk_ms, meta, kwargs)

    # ---- the whole-recording call in two parts for AutoModel.inference's loop over recordings (paraformer.py inference_begin): the
    #      scores of recording i + 1 are enqueued before those of recording i are read and cut into segments on the host
    # per-process staging objects (HIP streams, pinned buffers) are never copied or pickled with the module
    def __getstate__(self):
        st = dict(self.__dict__)
        for k in ("_upload", "_host_ring"):
            st.pop(k, None)
        return st

    def inference_begin(self, data_in, data_lengths=None, key: list = None, tokenizer=None, frontend=None, cache: dict = None, **kwargs):
        if (cache or frontend is None or frontend.lfr_n != 1 or not torch.cuda.is_available()
                or next(self.encoder.parameters()).device.type != "cuda"):
            return None
        chunk_ms = kwargs.get("chunk_size", 60000)
        streaming_input = kwargs.get("is_streaming_input", False) if chunk_ms >= 15000 else kwargs.get("is_streaming_input", True)
        is_final = kwargs.get("is_final", False) if streaming_input else kwargs.get("is_final", True)
        t1 = time.perf_counter()
        audio = load_audio_list(data_in if isinstance(data_in, (list, tuple)) else [data_in], fs=frontend.fs, audio_fs=kwargs.get("fs", 16000))
        if isinstance(data_in, (list, tuple)) and len(data_in) and isinstance(data_in[0], str):
            is_final, streaming_input = True, False
        if len(audio) != 1 or audio[0].numel() == 0 or streaming_input or not is_final:
            return None                                            # (the plain call decides: empty input, batch > 1, streaming input)
        meta: Dict[str, Any] = {"load_data": f"{time.perf_counter() - t1:0.3f}"}
        t2 = time.perf_counter()
        handles = self._scores_enqueue(frontend, audio[0])
        meta["extract_feat"] = f"{time.perf_counter() - t2:0.3f}"
        return dict(handles=handles, n=int(audio[0].numel()), key=key[0] if key else "", frontend=frontend, chunk_ms=chunk_ms, meta=meta,
                    kwargs=kwargs)

    def inference_launch(self, pending: dict) -> None:
        return None

    def inference_end(self, pending: dict):
        p_sil, db = self._scores_wait(pending["handles"])
        return self._decide(p_sil, db, pending["n"], pending["key"], pending["frontend"], self.init_cache({}, **pending["kwargs"]),
                            pending["chunk_ms"], pending["meta"], pending["kwargs"])

    def _decide(self, p_sil, db, n_samples: int, k0: str, frontend, cache: dict, chunk_ms: int, meta: dict, kwargs: dict):
        """scores of a whole recording -> segments (the block loop of model.py:985-1087 over the frame counts its online frontend emits)"""
        meta["batch_data_time"] = len(p_sil) * frontend.frame_shift * frontend.lfr_n / 1000
        dec: NativeVadDecision = cache["decision"]
        dynamic = kwargs.get("dynamic_silence", kwargs.get("max_end_silence_time") is None)
        schedule = kwargs.get("silence_schedule", DEFAULT_SILENCE_SCHEDULE)
        to_sil = self.vad_opts.speech_to_sil_time_thres
        acc_ms, in_speech = 0, False
        # the reference cuts the SAMPLES into chunk_size blocks and lets its online frontend decide how many frames each
        # block emits; with lfr (5, 1) block b of S samples yields frames up to the last one whose LFR window is complete
        # (two frames of look-ahead), the final block the rest: the frame counts below reproduce that
        n_total = len(p_sil)
        stride = int(chunk_ms * frontend.fs / 1000)
        n_blocks = int(n_samples // stride + 1)
        hop = int(self.vad_opts.frame_in_ms * self.vad_opts.sample_rate / 1000)
        flen = int(self.vad_opts.frame_length_ms * self.vad_opts.sample_rate / 1000)
        segments: List[List[int]] = []
        done = 0
        for b in range(n_blocks):
            last = b == n_blocks - 1
            if dynamic:                                                          # model.py:1005-1019
                if dec.state == IN_SPEECH or in_speech:
                    acc_ms += chunk_ms
                    in_speech = True
                for limit_ms, silence_ms in schedule:
                    if acc_ms <= limit_ms:
                        dec.max_end_sil_ms = max(silence_ms - to_sil, 0)
                        dec.speech_noise_thres = 0.5
                        break
            seen = min((b + 1) * stride, n_samples)
            fb = max((seen - flen) // hop + 1, 0) if seen >= flen else 0        # fbank frames available so far
            upto = n_total if last else max(min(fb - 2, n_total), done)          # LFR(5,1): two frames of look-ahead
            if upto > done or last:
                got = dec.push(p_sil[done:upto], db[done:upto], is_final=last, streaming_events=False) if upto > done else []
                done = upto
                if got:
                    segments.extend(got)
                    if dynamic:
                        acc_ms, in_speech = 0, False
        self.init_cache(cache)                                                    # model.py:1086-1087
        if kwargs.get("output_dir") is not None:
            if not hasattr(self, "writer"):
                from .datadir_writer import DatadirWriter
                self.writer = DatadirWriter(kwargs.get("output_dir"))
            self.writer["1best_recog"]["text"][k0] = segments
        return [{"key": k0, "value": segments}], meta

    def _inference_chunked(self, samples: torch.Tensor, k0: str, frontend, cache: dict, chunk_ms: int, is_final: bool,
                           streaming_input: bool, meta: dict, **kwargs):
        """Streaming input (model.py:985-1087): the samples of this call (after what the previous call left over) go
        through the ONLINE frontend, the network (left context in HBM) and the decision logic `chunk_size` ms at a time;
        a started segment is reported as [beg, -1] and closed later by [-1, end]."""
        if not hasattr(frontend, "init_cache"):
            raise ValueError("streaming VAD input needs the online frontend (frontend: WavFrontendOnline in config.yaml)")
        dev = next(self.encoder.parameters()).device
        o = self.vad_opts
        hop, flen = int(o.frame_in_ms * o.sample_rate / 1000), int(o.frame_length_ms * o.sample_rate / 1000)
        audio = torch.cat((cache["prev_samples"].to(samples.dtype), samples.cpu())) if len(cache["prev_samples"]) else samples.cpu()
        stride = int(chunk_ms * frontend.fs / 1000)
        n = int(len(audio) // stride + int(is_final))
        m = int(len(audio) % stride * (1 - int(is_final)))
        dec: NativeVadDecision = cache["decision"]
        dynamic = kwargs.get("dynamic_silence", kwargs.get("max_end_silence_time") is None)
        schedule = kwargs.get("silence_schedule", DEFAULT_SILENCE_SCHEDULE)
        acc_ms, in_speech = cache.get("_dynamic_accumulated_ms", 0), cache.get("_dynamic_in_speech", False)
        segments: List[List[int]] = []
        t2 = time.perf_counter()
        for i in range(n):
            final_i = is_final and i == n - 1
            chunk = audio[i * stride: (i + 1) * stride].to(dev)
            if dynamic:
                if dec.state == IN_SPEECH or in_speech:
                    acc_ms += chunk_ms
                    in_speech = True
                for limit_ms, silence_ms in schedule:
                    if acc_ms <= limit_ms:
                        dec.max_end_sil_ms = max(silence_ms - o.speech_to_sil_time_thres, 0)
                        dec.speech_noise_thres = 0.5
                        break
                cache["_dynamic_accumulated_ms"], cache["_dynamic_in_speech"] = acc_ms, in_speech
            # waveform history for the frame energies: samples from absolute index `wave_start` on
            cache["wave"] = chunk if cache.get("wave") is None else torch.cat((cache["wave"], chunk))
            feats, flens = frontend(chunk[None], [int(chunk.numel())], cache=cache["frontend"], is_final=final_i)
            meta["batch_data_time"] = float(flens.sum().item()) * frontend.frame_shift * frontend.lfr_n / 1000
            if feats.numel() == 0 or feats.dim() != 3:
                continue
            T = feats.shape[1]
            first = cache["frames_done"]                                       # absolute index of the first new frame
            lo = first * hop - cache.get("wave_start", 0)
            db = frame_decibel(cache["wave"][lo:], T, flen, hop)
            p_sil = self.encoder.silence_posterior(feats, cache["encoder"], o.sil_pdf_ids)[0]
            got = dec.push(p_sil.cpu().numpy(), db.cpu().numpy(), is_final=final_i, streaming_events=streaming_input)
            cache["frames_done"] = first + T
            keep_from = cache["frames_done"] * hop - cache.get("wave_start", 0)   # the next frame starts here
            keep_from = min(max(keep_from, 0), cache["wave"].numel())
            cache["wave"] = cache["wave"][keep_from:]
            cache["wave_start"] = cache.get("wave_start", 0) + keep_from
            if got:
                segments.extend(got)
                if dynamic:
                    acc_ms, in_speech = 0, False
                    cache["_dynamic_accumulated_ms"], cache["_dynamic_in_speech"] = 0, False
        meta["extract_feat"] = f"{time.perf_counter() - t2:0.3f}"
        cache["prev_samples"] = audio[-m:] if m > 0 else torch.empty(0)
        if is_final:
            self.init_cache(cache)
        return [{"key": k0, "value": segments}], meta

    def forward(self, *a, **k):  # pragma: no cover
        raise NotImplementedError("use inference(); the network is FSMN.silence_posterior, the logic vad_decision.VadDecision")
