"""Speech / silence segmentation from frame scores: the decision logic of FSMN-VAD (host side, integer / threshold
logic on two per-frame arrays; no GPU work).

Restates what `FsmnVADStreaming.forward` does AFTER the network (funasr/models/fsmn_vad_streaming/model.py): frame
classification from the silence posterior and the frame energy (`GetFrameState` :761-823), the sliding-window
sil<->speech detector (`WindowDetector` :218-320), the start / end point state machine with look-back, look-ahead,
maximum segment length and end-silence timeout (`DetectOneFrame` :1158-1302 and the On*/Pop* callbacks :552-736),
dropping of history behind confirmed end points (`DropCachedFrames` :406-433, `ResetDetection` :435-456) and the two
reporting conventions (`forward` :861-905: complete `[beg, end]` pairs offline, `[beg, -1]` / `[-1, end]` events when
streaming). The waveform buffers of the reference only ever influence the decision through their LENGTHS, so this
version carries sample counts instead of samples.

Pinned to the reference by tests/golden/vad_decision.npz (oracle/make_golden_vad.py drives the reference's own class with
injected network scores, offline and chunked).
"""
from __future__ import annotations

import math
from typing import List, Optional, Sequence

NOT_STARTED, IN_SPEECH, ENDED = 1, 2, 3            # the three states of the start/end point machine
SIL, SPEECH = 0, 1


class VadOptions:
    """The post-processing knobs of the model's config.yaml (VADXOptions, model.py:71-173), same names and defaults."""

    def __init__(self, sample_rate: int = 16000, detect_mode: int = 1, snr_mode: int = 0, max_end_silence_time: int = 800,
                 max_start_silence_time: int = 3000, do_start_point_detection: bool = True,
                 do_end_point_detection: bool = True, window_size_ms: int = 200, sil_to_speech_time_thres: int = 150,
                 speech_to_sil_time_thres: int = 150, speech_2_noise_ratio: float = 1.0, do_extend: int = 1,
                 lookback_time_start_point: int = 200, lookahead_time_end_point: int = 100,
                 max_single_segment_time: int = 60000, nn_eval_block_size: int = 8, dcd_block_size: int = 4,
                 snr_thres: float = -100.0, noise_frame_num_used_for_snr: int = 100, decibel_thres: float = -100.0,
                 speech_noise_thres: float = 0.6, fe_prior_thres: float = 1e-4, silence_pdf_num: int = 1,
                 sil_pdf_ids: Sequence[int] = (0,), speech_noise_thresh_low: float = -0.1,
                 speech_noise_thresh_high: float = 0.3, output_frame_probs: bool = False, frame_in_ms: int = 10,
                 frame_length_ms: int = 25, **kwargs):
        self.__dict__.update({k: v for k, v in locals().items() if k not in ("self", "kwargs")})
        self.sil_pdf_ids = list(sil_pdf_ids)


class _Window:
    """Majority vote over the last `size` frame labels with hysteresis (WindowDetector, model.py:218-320)."""

    def __init__(self, opts: VadOptions):
        self.size = int(opts.window_size_ms / opts.frame_in_ms)
        self.to_speech = int(opts.sil_to_speech_time_thres / opts.frame_in_ms)
        self.to_sil = int(opts.speech_to_sil_time_thres / opts.frame_in_ms)
        self.reset()

    def reset(self):
        self.ring, self.pos, self.total, self.speaking = [0] * self.size, 0, 0, False

    def push(self, label: int) -> str:
        self.total += label - self.ring[self.pos]
        self.ring[self.pos] = label
        self.pos = (self.pos + 1) % self.size
        if not self.speaking and self.total >= self.to_speech:
            self.speaking = True
            return "sil>speech"
        if self.speaking and self.total <= self.to_sil:
            self.speaking = False
            return "speech>sil"
        return "speech" if self.speaking else "sil"


class _Segment:
    __slots__ = ("start_ms", "end_ms", "has_start", "has_end")

    def __init__(self, start_ms: int):
        self.start_ms = self.end_ms = start_ms
        self.has_start = self.has_end = False


def _alias_max_end_sil(cls):
    """the reference's name for max_end_sil_ms: `cache["stats"].max_end_sil_frame_cnt_thresh` (model.py:929-936)"""
    cls.max_end_sil_frame_cnt_thresh = property(lambda self: self.max_end_sil_ms,
                                                lambda self, v: setattr(self, "max_end_sil_ms", v))
    return cls


@_alias_max_end_sil
class VadDecision:
    """State of one audio stream. `push()` takes the network's silence scores and the frame energies of the next block
    of frames and returns the segments that became reportable."""

    def __init__(self, opts: Optional[VadOptions] = None, speech_noise_thres: Optional[float] = None, **kwargs):
        self.o = opts if opts is not None else VadOptions(**kwargs)
        o = self.o
        self.shift = int(o.frame_in_ms * o.sample_rate / 1000)              # samples per frame hop
        self.flen = int(o.frame_length_ms * o.sample_rate / 1000)           # samples per analysis frame
        self.win = _Window(o)
        self.max_end_sil_ms = o.max_end_silence_time - o.speech_to_sil_time_thres       # Stats(...), model.py:929-936
        self.speech_noise_thres = o.speech_noise_thres if speech_noise_thres is None else speech_noise_thres
        self.state = NOT_STARTED
        self.n_frames = 0                  # frames seen so far (absolute index of the next frame)
        self.dropped = 0                   # absolute index of the first frame whose score / energy is still kept
        self.sil_score: List[float] = []   # per kept frame
        self.decibel: List[float] = []
        self.noise_db = -100.0
        self.buf_start = 0                 # first frame not yet handed to an output segment ("data_buf_start_frame")
        self.kept_samples = 0              # len(data_buf_all)
        self.buf_off = 0                   # offset of data_buf inside data_buf_all, in samples
        self.first_block = True
        self.last_speech = 0
        self.last_silence = -1
        self.sil_run = 0
        self.start_frame = self.end_frame = -1
        self.n_ends = 0
        self.segments: List[_Segment] = []
        self.reported = 0
        self.next_is_new = True            # "next_seg": the next streaming event opens a segment

    # ------------------------------------------------------------------ sample-count bookkeeping of the waveform buffers
    def _buf_len(self) -> int:
        return max(self.kept_samples - self.buf_off, 0)

    def _reslice(self):
        self.buf_off = max(self.buf_start - self.dropped, 0) * self.shift

    def _take_samples(self, n_frames: int):
        if self.first_block:
            self.kept_samples = (n_frames - 1) * self.shift + self.flen
            self.first_block = False
        else:
            self.kept_samples += n_frames * self.shift
        self._reslice()

    def _drop_before(self, frame: int):
        if frame > self.n_frames:
            raise RuntimeError(f"Cannot drop through frame {frame}; only {self.n_frames} frames exist")
        k = frame - self.dropped
        if k <= 0:
            return
        self.kept_samples = max(self.kept_samples - k * self.shift, 0)
        del self.decibel[:k]
        del self.sil_score[:k]
        self.dropped = frame
        self._reslice()

    def _skip_to(self, frame: int):
        while self.buf_start < frame:
            if self._buf_len() < self.shift:
                raise RuntimeError("VAD: waveform history exhausted while advancing the segment buffer")
            self.buf_start += 1
            self.buf_off = (self.buf_start - self.dropped) * self.shift

    # ------------------------------------------------------------------------------------------ segment bookkeeping
    def _emit(self, frame: int, count: int, opens: bool, closes: bool):
        self._skip_to(frame)
        if not self.segments or opens:
            self.segments.append(_Segment(frame * self.o.frame_in_ms))
        seg = self.segments[-1]
        self.buf_start += count
        seg.end_ms = (frame + count) * self.o.frame_in_ms
        seg.has_start = seg.has_start or opens
        seg.has_end = seg.has_end or closes

    def _speech(self, frame: int):
        self.last_speech = frame
        self._emit(frame, 1, False, False)

    def _silence(self, frame: int):
        self.last_silence = frame
        if self.state == NOT_STARTED:
            self._skip_to(frame)

    def _begin(self, frame: int, fake: bool = False):
        if self.start_frame == -1:
            self.start_frame = frame
        if not fake and self.state == NOT_STARTED:
            self._emit(self.start_frame, 1, True, False)

    def _finish(self, frame: int, fake: bool):
        for t in range(self.last_speech + 1, frame):
            self._speech(t)
        if self.end_frame == -1:
            self.end_frame = frame
        if not fake:
            self._emit(self.end_frame, 1, False, True)
        self.n_ends += 1

    def _restart(self):
        self.sil_run, self.last_speech, self.last_silence = 0, 0, -1
        self.start_frame = self.end_frame = -1
        self.state = NOT_STARTED
        self.win.reset()
        if self.segments:
            assert self.segments[-1].has_end
            self._drop_before(int(self.segments[-1].end_ms / self.o.frame_in_ms))

    # -------------------------------------------------------------------------------------------- per-frame decision
    def _label(self, t: int) -> int:
        """speech / silence of kept frame t from its silence posterior and energy (GetFrameState)."""
        o = self.o
        if t >= len(self.decibel):
            return SIL
        db = self.decibel[t]
        snr = db - self.noise_db
        if db < o.decibel_thres:
            return SIL
        p_sil = self.sil_score[t]
        noise_prob = math.log(p_sil) * o.speech_2_noise_ratio
        speech_prob = math.log(1.0 - p_sil)
        if math.exp(speech_prob) >= math.exp(noise_prob) + self.speech_noise_thres:
            return SPEECH if (snr >= o.snr_thres and db >= o.decibel_thres) else SIL
        if self.noise_db < -99.9:
            self.noise_db = db
        else:
            self.noise_db = (db + self.noise_db * (o.noise_frame_num_used_for_snr - 1)) / o.noise_frame_num_used_for_snr
        return SIL

    def _latency(self) -> int:
        n = self.win.size
        if self.o.do_extend:
            n += int(self.o.lookback_time_start_point / self.o.frame_in_ms)
        return n

    def _step(self, label: int, frame: int, final: bool):
        o, ms = self.o, self.o.frame_in_ms
        if label == SPEECH and not (math.fabs(1.0) > o.fe_prior_thres):
            label = SIL
        change = self.win.push(label)
        too_long = lambda: frame - self.start_frame + 1 > o.max_single_segment_time / ms   # noqa: E731

        def continue_or_close():
            if too_long():
                self._finish(frame, False)
                self.state = ENDED
            elif not final:
                self._speech(frame)
            else:
                self._finish(frame, False)
                self.state = ENDED

        if change == "sil>speech":
            self.sil_run = 0
            if self.state == NOT_STARTED:
                first = max(self.buf_start, frame - self._latency())
                self._begin(first)
                self.state = IN_SPEECH
                for t in range(first + 1, frame + 1):
                    self._speech(t)
            elif self.state == IN_SPEECH:
                for t in range(self.last_speech + 1, frame):
                    self._speech(t)
                continue_or_close()
        elif change in ("speech>sil", "speech"):
            self.sil_run = 0
            if self.state == IN_SPEECH:
                continue_or_close()
        else:                                                     # silence continues
            self.sil_run += 1
            if self.state == NOT_STARTED:
                timed_out = o.detect_mode == 0 and self.sil_run * ms > o.max_start_silence_time
                if timed_out or (final and self.n_ends == 0):
                    for t in range(self.last_silence + 1, frame):
                        self._silence(t)
                    self._begin(0, fake=True)
                    self._finish(0, fake=True)
                    self.state = ENDED
                elif frame >= self._latency():
                    self._silence(frame - self._latency())
            elif self.state == IN_SPEECH:
                if self.sil_run * ms >= self.max_end_sil_ms:
                    back = int(self.max_end_sil_ms / ms)
                    if o.do_extend:
                        back = max(0, back - int(o.lookahead_time_end_point / ms) - 1)
                    self._finish(frame - back, False)
                    self.state = ENDED
                elif too_long():
                    self._finish(frame, False)
                    self.state = ENDED
                elif o.do_extend and not final:
                    if self.sil_run <= int(o.lookahead_time_end_point / ms):
                        self._speech(frame)
                elif final:
                    self._finish(frame, False)
                    self.state = ENDED
        if self.state == ENDED and o.detect_mode == 1:
            self._restart()

    # ---------------------------------------------------------------------------------------------------- interface
    def push(self, sil_scores: Sequence[float], decibels: Sequence[float], is_final: bool = False,
             streaming_events: bool = False) -> List[List[int]]:
        """sil_scores[t]: summed posterior of the silence pdfs of frame t; decibels[t]: 10 log10(frame energy + 1e-6).
        -> [[beg_ms, end_ms], ...]; with streaming_events a started segment is reported as [beg, -1] and closed later
        by [-1, end]."""
        n = len(sil_scores)
        if n == 0:
            return []
        if len(decibels) != n:
            raise RuntimeError(f"VAD score frames and energies are not aligned: {n} vs {len(decibels)}")
        self._take_samples(n)
        self.decibel.extend(float(d) for d in decibels)
        self.sil_score.extend(float(s) for s in sil_scores)
        self.n_frames += n
        if self.state != ENDED:
            for back in range(n - 1, -1, -1):
                frame = self.n_frames - 1 - back
                self._step(self._label(frame - self.dropped), frame, is_final and back == 0)
        self._drop_before(self.buf_start)
        out: List[List[int]] = []
        for i in range(self.reported, len(self.segments)):
            seg = self.segments[i]
            if streaming_events:
                if not seg.has_start or (not self.next_is_new and not seg.has_end):
                    continue
                beg = seg.start_ms if self.next_is_new else -1
                if seg.has_end:
                    end, self.next_is_new = seg.end_ms, True
                    self.reported += 1
                else:
                    end, self.next_is_new = -1, False
                out.append([beg, end])
            else:
                if not is_final and not (seg.has_start and seg.has_end):
                    continue
                out.append([seg.start_ms, seg.end_ms])
                self.reported += 1
        return out


class NativeVadDecision:
    """The same logic in native host code (csrc/vad_decision.hip, C ABI `pf_vad_decision_*`): ~100x the frame rate of the
    Python loop above, which stays as the readable restatement. Same interface, same golden vectors."""

    def __init__(self, opts: Optional[VadOptions] = None, speech_noise_thres: Optional[float] = None, **kwargs):
        import ctypes as C
        from . import _lib
        self.o = opts if opts is not None else VadOptions(**kwargs)
        o = self.o
        if len(o.sil_pdf_ids) != o.silence_pdf_num:
            raise ValueError("VadOptions: len(sil_pdf_ids) must equal silence_pdf_num")
        self._C, self._lib = C, _lib.load()
        c = _lib.pf_vad_options(o.sample_rate, o.detect_mode, o.max_end_silence_time, o.max_start_silence_time, o.window_size_ms,
                                o.sil_to_speech_time_thres, o.speech_to_sil_time_thres, int(o.do_extend),
                                o.lookback_time_start_point, o.lookahead_time_end_point, o.max_single_segment_time,
                                o.noise_frame_num_used_for_snr, o.frame_in_ms, o.frame_length_ms, float(o.speech_2_noise_ratio),
                                float(o.snr_thres), float(o.decibel_thres),
                                float(o.speech_noise_thres if speech_noise_thres is None else speech_noise_thres),
                                float(o.fe_prior_thres))
        self._h = _lib.check_handle(self._lib.pf_vad_decision_create(C.byref(c)), "pf_vad_decision_create")
        self._max_end_sil_ms = float(o.max_end_silence_time - o.speech_to_sil_time_thres)
        self._speech_noise_thres = float(c.speech_noise_thres)
        self._out = (C.c_int32 * 512)()

    def __del__(self):
        h = self.__dict__.get("_h")
        if h is not None:
            try:
                self._lib.pf_vad_decision_destroy(h)
            except Exception:
                pass
            self._h = None

    @property
    def state(self) -> int:
        return int(self._lib.pf_vad_decision_state(self._h))

    def _sync(self):
        self._lib.pf_vad_decision_set_thresholds(self._h, self._max_end_sil_ms, self._speech_noise_thres)

    max_end_sil_ms = property(lambda self: self._max_end_sil_ms,
                              lambda self, v: (setattr(self, "_max_end_sil_ms", float(v)), self._sync())[1])
    speech_noise_thres = property(lambda self: self._speech_noise_thres,
                                  lambda self, v: (setattr(self, "_speech_noise_thres", float(v)), self._sync())[1])
    # the reference's name for the same quantity (`cache["stats"].max_end_sil_frame_cnt_thresh`, written by
    # DynamicStreamingVAD, funasr/models/fsmn_vad_streaming/dynamic_vad.py:96-104)
    max_end_sil_frame_cnt_thresh = max_end_sil_ms

    def push(self, sil_scores, decibels, is_final: bool = False, streaming_events: bool = False) -> List[List[int]]:
        import numpy as np
        from . import _lib
        p = np.ascontiguousarray(sil_scores, dtype=np.float32)
        d = np.ascontiguousarray(decibels, dtype=np.float32)
        if p.shape != d.shape or p.ndim != 1:
            raise RuntimeError(f"VAD score frames and energies are not aligned: {p.shape} vs {d.shape}")
        if p.size == 0:
            return []
        cap = len(self._out) // 2
        m = self._lib.pf_vad_decision_push(self._h, p.ctypes.data, d.ctypes.data, int(p.size), int(is_final),
                                           int(streaming_events), self._out, cap)
        _lib.check(0 if m >= 0 else m, "pf_vad_decision_push")
        if m > cap:
            raise RuntimeError(f"pf_vad_decision_push: {m} segments in one block (capacity {cap})")
        return [[int(self._out[2 * i]), int(self._out[2 * i + 1])] for i in range(m)]
