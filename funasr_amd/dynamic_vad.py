"""Streaming VAD with a silence threshold that follows the length of the running segment (host-side wrapper).

Mirror of `DynamicStreamingVAD` (funasr/models/fsmn_vad_streaming/dynamic_vad.py:37-230): chunks are fed to a streaming
fsmn-vad `AutoModel`; before every chunk the end-of-speech silence the VAD waits for is re-chosen from a schedule
[(segment length so far in ms, silence in ms), ...] -- short utterances wait long (no chopping), long ones are cut quickly --
by writing `speech_noise_thres` and `max_end_sil_frame_cnt_thresh` of the VAD's `cache["stats"]`. The VAD's streaming
events ([beg, -1], [-1, end], [beg, end]) are folded into finished [beg_ms, end_ms] segments. Works with the reference's
model object or with funasr_amd's (whose decision state answers to the same two attribute names).
"""
from __future__ import annotations

from typing import List, Optional, Sequence, Tuple

import numpy as np
import torch

DEFAULT_SILENCE_SCHEDULE = [(5000, 2000), (10000, 1500), (15000, 1000), (30000, 800), (45000, 400), (float("inf"), 100)]


class DynamicStreamingVAD:
    def __init__(self, vad_model, chunk_size_ms: int = 60, speech_noise_thres: float = 0.5, speech_to_sil_thres_ms: int = 150,
                 silence_schedule: Optional[Sequence[Tuple[float, int]]] = None, sample_rate: int = 16000):
        self.model = vad_model
        self.chunk_size_ms, self.sample_rate = chunk_size_ms, sample_rate
        self.speech_noise_thres, self.speech_to_sil_thres_ms = speech_noise_thres, speech_to_sil_thres_ms
        self.silence_schedule = list(silence_schedule) if silence_schedule is not None else DEFAULT_SILENCE_SCHEDULE
        self.reset()

    def reset(self) -> None:
        self.cache: dict = {}
        self.confirmed_segments: List[List[int]] = []
        self.current_speech_start: Optional[int] = None
        self.accumulated_since_cut_ms = 0

    # --------------------------------------------------------------------------------------------------- state
    @property
    def is_speaking(self) -> bool:
        return self.current_speech_start is not None

    @property
    def current_duration_ms(self) -> int:
        return self.accumulated_since_cut_ms

    @property
    def current_threshold_ms(self) -> int:
        for limit_ms, silence_ms in self.silence_schedule:
            if self.accumulated_since_cut_ms <= limit_ms:
                return silence_ms
        return self.silence_schedule[-1][1]

    # -------------------------------------------------------------------------------------------------- feeding
    def feed(self, audio_chunk: torch.Tensor, is_final: bool = False) -> List[List[int]]:
        """any number of samples in, the segments that ENDED inside them out"""
        if audio_chunk.dim() > 1:
            audio_chunk = audio_chunk.squeeze()
        self.accumulated_since_cut_ms += int(len(audio_chunk) * 1000 / self.sample_rate)
        first_call = {}
        stats = self.cache.get("stats")
        if stats is not None:
            stats.speech_noise_thres = self.speech_noise_thres
            stats.max_end_sil_frame_cnt_thresh = max(self.current_threshold_ms - self.speech_to_sil_thres_ms, 0)
        else:                                    # the VAD builds its state from these on its first call
            first_call = dict(max_end_silence_time=self.current_threshold_ms, speech_noise_thres=self.speech_noise_thres)
        res = self.model.generate(input=[audio_chunk], cache=self.cache, is_final=is_final, chunk_size=self.chunk_size_ms,
                                  dynamic_silence=False, **first_call)
        done: List[List[int]] = []
        for beg, end in res[0].get("value", []):
            if beg >= 0 and end == -1:
                self.current_speech_start = beg
                continue
            if beg == -1 and end >= 0:
                seg = [self.current_speech_start if self.current_speech_start is not None else 0, end]
            elif beg >= 0 and end >= 0:
                seg = [beg, end]
            else:
                continue
            self.confirmed_segments.append(seg)
            done.append(seg)
            self.current_speech_start = None
            self.accumulated_since_cut_ms = 0
        return done

    def finalize(self) -> List[List[int]]:
        """10 ms of silence with is_final: closes a segment that is still open"""
        return self.feed(torch.zeros(int(self.sample_rate * 0.01), dtype=torch.float32), is_final=True)

    def process(self, audio) -> List[List[int]]:
        """a whole recording through the same chunk loop"""
        self.reset()
        if isinstance(audio, np.ndarray):
            audio = torch.from_numpy(audio).float()
        if audio.dim() > 1:
            audio = audio.squeeze()
        step = int(self.sample_rate * self.chunk_size_ms / 1000)
        out: List[List[int]] = []
        for i in range(0, len(audio), step):
            out.extend(self.feed(audio[i: i + step], is_final=i + step >= len(audio)))
        return out
