"""WavFrontend on gfx950: batched fbank + LFR + CMVN in two kernels (csrc/frontend.hip).

Host-side mirror of funasr/frontends/wav_frontend.py:89-196 (`WavFrontend`, registered as "WavFrontend" /
"wav_frontend" in `frontend_classes`), same constructor keywords, `output_size()`, and
`forward(input [B, n], input_lengths [B]) -> (feats [B, T, n_mels*lfr_m], feats_lens [B])`.
Differences, on purpose: the whole batch is one launch instead of a Python loop over utterances and the result stays in
HBM. `dither` (the reference default is 1.0: Gaussian noise per frame sample from torch.randn, wav_frontend.py:106,171-181)
is drawn inside the fbank kernel from a counter-based generator keyed by `dither_seed`: the same seed reproduces the same
feature sequence; parity with the reference's noise is statistical by nature. `dither=0` (this class's default, and what
the reference's own C++ runtime pins, runtime/onnxruntime/src/paraformer.cpp:24) is the deterministic parity mode.
"""
from __future__ import annotations

import ctypes as C

import numpy as np
import os

import torch
import torch.nn as nn

from . import _lib
from .hip_module import host_i32, stream_ptr
from .register import tables


def load_cmvn(cmvn_file: str) -> torch.Tensor:
    """Kaldi-nnet am.mvn -> [2, dim] (row 0 <AddShift>, row 1 <Rescale>); format per wav_frontend.py:15-43."""
    shift, scale = None, None
    with open(cmvn_file, "r", encoding="utf-8") as f:
        lines = f.readlines()
    for i, line in enumerate(lines):
        tok = line.split()
        if tok and tok[0] in ("<AddShift>", "<Rescale>") and i + 1 < len(lines):
            nxt = lines[i + 1].split()
            if nxt and nxt[0] == "<LearnRateCoef>":
                vals = np.array(nxt[3:len(nxt) - 1]).astype(np.float32)
                if tok[0] == "<AddShift>":
                    shift = vals
                else:
                    scale = vals
    if shift is None or scale is None or shift.shape != scale.shape:
        raise ValueError(f"{cmvn_file}: could not find matching <AddShift>/<Rescale> rows")
    return torch.from_numpy(np.stack([shift, scale]))


WINDOW_TYPES = ("hamming", "hanning", "povey", "rectangular", "blackman")


def kaldi_window(window_type: str, window_size: int, blackman_coeff: float = 0.42) -> torch.Tensor:
    """The analysis window as torchaudio.compliance.kaldi._feature_window_function builds it in float32 (what the reference's
    kaldi.fbank(window_type=self.window) evaluates, wav_frontend.py:171-181); Kaldi's definitions are
    kaldi-native-fbank/csrc/feature-window.cc:25-55."""
    import math

    if window_type == "hamming":
        return torch.hamming_window(window_size, periodic=False, alpha=0.54, beta=0.46, dtype=torch.float32)
    if window_type == "hanning":
        return torch.hann_window(window_size, periodic=False, dtype=torch.float32)
    if window_type == "povey":
        return torch.hann_window(window_size, periodic=False, dtype=torch.float32).pow(0.85)
    if window_type == "rectangular":
        return torch.ones(window_size, dtype=torch.float32)
    if window_type == "blackman":
        a = 2 * math.pi / (window_size - 1)
        i = torch.arange(window_size, dtype=torch.float32)
        return (blackman_coeff - 0.5 * torch.cos(a * i) + (0.5 - blackman_coeff) * torch.cos(2 * a * i)).to(torch.float32)
    raise ValueError(f"Invalid window type {window_type!r} (one of {WINDOW_TYPES})")


def kaldi_tables(n_mels: int, window_size: int, fs: float, low_freq: float = 20.0, high_freq: float = 0.0,
                 window_type: str = "hamming"):
    """Window and dense mel matrix in the float32 arithmetic torchaudio.compliance.kaldi uses, so that the device
    frontend matches the reference's Python path to float32 round-off (the library's built-in tables follow
    kaldi-native-fbank's float64-cosine / float-scalar construction instead)."""
    import math

    padded = 1 << (window_size - 1).bit_length()
    window = kaldi_window(window_type, window_size)
    nyq = 0.5 * fs
    hf = high_freq + nyq if high_freq <= 0.0 else high_freq
    mlo = 1127.0 * math.log(1.0 + low_freq / 700.0)
    mhi = 1127.0 * math.log(1.0 + hf / 700.0)
    delta = (mhi - mlo) / (n_mels + 1)
    b = torch.arange(n_mels, dtype=torch.float32).unsqueeze(1)
    left, center, right = mlo + b * delta, mlo + (b + 1.0) * delta, mlo + (b + 2.0) * delta
    mel = (1127.0 * (1.0 + ((fs / padded) * torch.arange(padded // 2, dtype=torch.float32)) / 700.0).log()).unsqueeze(0)
    bins = torch.max(torch.zeros(1), torch.min((mel - left) / (center - left), (right - mel) / (right - center)))
    return window.contiguous(), torch.nn.functional.pad(bins, (0, 1)).contiguous()


def mel_on_the_512_grid(mel: torch.Tensor) -> torch.Tensor:
    """[n_mels, P/2 + 1] weights over the bins of a P-point FFT (P = 2 .. 512) as a dense [n_mels, 257] matrix over the bins of the
    kernel's 512-point FFT: the zero-padded 512-point transform of a frame holds the P-point one at every (512 / P)-th bin exactly
    (X_512[k 512/P] = X_P[k]), so a shorter window's mel energies are the same sums with the weights placed at those bins."""
    P = 2 * (mel.shape[1] - 1)
    if P == 512:
        return mel.contiguous()
    out = torch.zeros(mel.shape[0], 257, dtype=mel.dtype)
    out[:, :: 512 // P] = mel
    return out


@tables.register("frontend_classes", "wav_frontend")
@tables.register("frontend_classes", "WavFrontend")
class WavFrontend(nn.Module):
    def __init__(self, cmvn_file: str = None, fs: int = 16000, window: str = "hamming", n_mels: int = 80,
                 frame_length: int = 25, frame_shift: int = 10, filter_length_min: int = -1,
                 filter_length_max: int = -1, lfr_m: int = 1, lfr_n: int = 1, dither: float = 0.0,
                 snip_edges: bool = True, upsacle_samples: bool = True, cmvn: torch.Tensor = None,
                 device=None, dither_seed: int = None, verify: bool = None, window_samples: int = None, **kwargs):
        super().__init__()
        # window_samples: the analysis window in samples instead of frame_length ms -- the per-clip window of clips shorter than
        # frame_length (wav_frontend.py:176); used by this class for its own short-clip children
        self._win_override = None if window_samples is None else int(window_samples)
        self._short = {}
        # verify: the fbank kernel evaluates every frame twice and repeats until two runs agree (pf_frontend_set_verify): for a GPU
        # this process SHARES with another one. None = on exactly when the launcher says so (PF_FRONTEND_VERIFY=1)
        self.verify = bool(int(os.environ.get("PF_FRONTEND_VERIFY", "0"))) if verify is None else bool(verify)
        if window not in WINDOW_TYPES:
            raise ValueError(f"Invalid window type {window!r} (one of {WINDOW_TYPES})")
        if float(dither) < 0.0:
            raise ValueError("dither must be >= 0")
        self.dither_seed = int(torch.initial_seed() if dither_seed is None else dither_seed) & 0xFFFFFFFFFFFFFFFF
        self.fs, self.window, self.n_mels = fs, window, n_mels
        self.frame_length, self.frame_shift = frame_length, frame_shift
        self.filter_length_min, self.filter_length_max = filter_length_min, filter_length_max
        self.lfr_m, self.lfr_n = lfr_m, lfr_n
        self.cmvn_file, self.dither, self.snip_edges, self.upsacle_samples = cmvn_file, dither, snip_edges, upsacle_samples
        self.cmvn = cmvn if cmvn is not None else (None if cmvn_file is None else load_cmvn(cmvn_file))
        self._device_arg = device
        self._handle = None
        self._handle_device = None

    def output_size(self) -> int:
        return self.n_mels * self.lfr_m

    def faults(self) -> int:
        """disagreements the cross-check (verify=True) has seen on this handle (pf_frontend_faults); 0 without a handle"""
        if self._handle is None:
            return 0
        n = C.c_uint32(0)
        _lib.check(_lib.load().pf_frontend_faults(self._handle, C.byref(n)), "pf_frontend_faults")
        return int(n.value)

    def fault_log(self):
        """the first 16 disagreements: (cycles of the first evaluation, cycles of the second, frame index, retry) each"""
        if self._handle is None:
            return []
        buf = (C.c_uint32 * 64)()
        _lib.check(_lib.load().pf_frontend_fault_log(self._handle, buf), "pf_frontend_fault_log")
        return [tuple(int(buf[4 * k + i]) for i in range(4)) for k in range(min(16, self.faults()))]

    # ------------------------------------------------------------------------------------------------ internals
    def _ensure_handle(self, dev: torch.device):
        lib = _lib.load()
        if self._handle is not None and self._handle_device == dev:
            return lib, self._handle
        if dev.type != "cuda":
            raise RuntimeError("WavFrontend runs only on an AMD GPU through libparaformer_hip.so (no CPU fallback)")
        self.close()
        win = self._win_override or int(self.fs * self.frame_length * 0.001)
        cfg = _lib.pf_frontend_config(self.fs, win, int(self.fs * self.frame_shift * 0.001), self.n_mels, self.lfr_m,
                                      self.lfr_n, 20.0, 0.0, 0.97, float(1 << 15) if self.upsacle_samples else 1.0)
        with torch.cuda.device(dev):
            h = _lib.check_handle(lib.pf_frontend_create(C.byref(cfg)), "pf_frontend_create")
            w, mel = kaldi_tables(self.n_mels, win, float(self.fs), window_type=self.window)
            mel = mel_on_the_512_grid(mel)
            _lib.check(lib.pf_frontend_set_tables(h, w.data_ptr(), mel.data_ptr()), "pf_frontend_set_tables")
            if not self.snip_edges:
                _lib.check(lib.pf_frontend_set_snip_edges(h, 0), "pf_frontend_set_snip_edges")
            if float(self.dither) != 0.0:
                _lib.check(lib.pf_frontend_set_dither(h, float(self.dither), self.dither_seed), "pf_frontend_set_dither")
            if self.verify:
                _lib.check(lib.pf_frontend_set_verify(h, 1), "pf_frontend_set_verify")
            if self.cmvn is not None:
                c = self.cmvn.to(torch.float32).contiguous().cpu()
                dim = self.output_size()
                _lib.check(lib.pf_frontend_set_cmvn(h, c[0, :dim].contiguous().data_ptr(),
                                                    c[1, :dim].contiguous().data_ptr(), dim), "pf_frontend_set_cmvn")
        self._handle, self._handle_device = h, dev
        return lib, h

    def close(self):
        h = self.__dict__.get("_handle")
        if h is not None:
            try:
                _lib.load().pf_frontend_destroy(h)
            except Exception:
                pass
            self.__dict__["_handle"] = None      # not nn.Module.__setattr__: it is unusable at interpreter exit

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __getstate__(self):
        # device handles (this object's and the short-clip children's) are not part of a pickle or a deep copy: the copy makes its own
        st = dict(self.__dict__)
        st["_handle"], st["_handle_device"], st["_short"] = None, None, {}
        return st

    def _target_device(self, x: torch.Tensor) -> torch.device:
        if x.is_cuda:
            return x.device
        if self._device_arg is not None:
            return torch.device(self._device_arg)
        if not torch.cuda.is_available():
            raise RuntimeError("WavFrontend: no GPU visible; the HIP frontend is the only implementation")
        return torch.device("cuda", torch.cuda.current_device())

    def num_fbank_frames(self, n_samples: int) -> int:
        """Kaldi's NumFrames (feature-window.cc:66-90): whole windows inside the waveform, or with snip_edges=False one frame per
        shift, centred, the ends mirrored"""
        win, hop = self.window_samples(n_samples), int(self.fs * self.frame_shift * 0.001)
        if not self.snip_edges:
            return (n_samples + hop // 2) // hop
        return 0 if n_samples < win else 1 + (n_samples - win) // hop

    def window_samples(self, n_samples: int) -> int:
        """The analysis window of a clip of n_samples: `frame_length=min(self.frame_length, waveform_length / self.fs * 1000)`
        (wav_frontend.py:176; waveform_length is a 0-dim integer tensor there, so the right-hand side is float32 tensor arithmetic and
        torchaudio's int(sample_frequency * frame_length * 0.001) truncates a float32 product) -- a clip shorter than frame_length
        ms is analysed with one window of its own length, and the FFT size follows."""
        win = int(self.fs * self.frame_length * 0.001)
        if self._win_override is not None:
            return self._win_override
        if n_samples >= win:
            return win
        ms = torch.tensor(int(n_samples)) / self.fs * 1000
        return win if not bool(ms < self.frame_length) else int(float(self.fs) * ms * 0.001)

    def num_frames(self, n_samples: int) -> int:
        return (self.num_fbank_frames(n_samples) + self.lfr_n - 1) // self.lfr_n

    def forward(self, input: torch.Tensor, input_lengths, return_fbank: bool = False, **kwargs):
        if input.dim() == 1:
            input = input[None, :]
        dev = self._target_device(input)
        lib, h = self._ensure_handle(dev)
        wav = input.to(device=dev, dtype=torch.float32).contiguous()
        B, n_max = wav.shape
        lens_c, lens = host_i32(input_lengths, B)
        win = self._win_override or int(self.fs * self.frame_length * 0.001)
        if any(self.window_samples(n) != win for n in lens):
            return self._forward_with_short_clips(wav, lens, win, return_fbank)
        T = max(self.num_frames(n) for n in lens)
        if T <= 0:
            raise ValueError("WavFrontend: every utterance must hold at least one frame")
        feats = torch.empty(B, T, self.output_size(), device=dev, dtype=torch.float32)
        out_lens = (C.c_int32 * B)()
        fb = None
        if return_fbank:
            tfb = max(self.num_fbank_frames(n) for n in lens)
            fb = torch.zeros(B, tfb, self.n_mels, device=dev, dtype=torch.float32)
        with torch.cuda.device(dev):
            _lib.check(lib.pf_frontend_forward(h, wav.data_ptr(), wav.stride(0), lens_c, B, feats.data_ptr(), T,
                                               out_lens, fb.data_ptr() if fb is not None else None, stream_ptr()),
                       "pf_frontend_forward")
        feats_lens = torch.tensor(list(out_lens), dtype=torch.int32)
        if return_fbank:
            return feats, feats_lens, fb
        return feats, feats_lens

    def _forward_with_short_clips(self, wav: torch.Tensor, lens, win: int, return_fbank: bool):
        """A batch that holds clips shorter than frame_length ms (a VAD segment of a few milliseconds): the reference's loop gives
        each of them its own window (`window_samples`); here the other clips go through the batched launch and every short one through
        a child frontend built for its window length (cached per length), and the rows are put together like pad_sequence does."""
        B = wav.shape[0]
        dev = wav.device
        for n in lens:
            if self.window_samples(n) < 2:                          # torchaudio: assert 2 <= window_size
                raise ValueError(f"WavFrontend: a clip of {n} sample(s) has no analysis window (window size must be >= 2)")
        T = max(self.num_frames(n) for n in lens)
        if T <= 0:
            raise ValueError("WavFrontend: every utterance must hold at least one frame")
        feats = torch.zeros(B, T, self.output_size(), device=dev, dtype=torch.float32)
        flens = [0] * B
        fb = torch.zeros(B, max(self.num_fbank_frames(n) for n in lens), self.n_mels, device=dev, dtype=torch.float32) if return_fbank else None
        groups = {}
        for i, n in enumerate(lens):
            groups.setdefault(self.window_samples(n), []).append(i)
        for ws, idx in groups.items():
            fe = self if ws == win else self._short_clip_frontend(ws)
            sub_lens = [lens[i] for i in idx]
            sub = wav[idx][:, : max(sub_lens)].contiguous()
            out = fe.forward(sub, sub_lens, return_fbank=return_fbank)
            for j, i in enumerate(idx):
                t = int(out[1][j])
                feats[i, :t] = out[0][j, :t]
                flens[i] = t
                if return_fbank:
                    tf = fe.num_fbank_frames(lens[i])
                    fb[i, :tf] = out[2][j, :tf]
        feats_lens = torch.tensor(flens, dtype=torch.int32)
        if return_fbank:
            return feats, feats_lens, fb
        return feats, feats_lens

    def _short_clip_frontend(self, ws: int) -> "WavFrontend":
        fe = self._short.get(ws)
        if fe is None:
            if len(self._short) >= 32:                              # one handle per distinct length: keep the set small
                self._short.pop(next(iter(self._short))).close()
            fe = self._short[ws] = WavFrontend(cmvn=self.cmvn, fs=self.fs, window=self.window, n_mels=self.n_mels, frame_length=self.frame_length,
                                               frame_shift=self.frame_shift, lfr_m=self.lfr_m, lfr_n=self.lfr_n, dither=self.dither,
                                               snip_edges=self.snip_edges, upsacle_samples=self.upsacle_samples, device=self._device_arg,
                                               dither_seed=self.dither_seed, verify=self.verify, window_samples=ws)
        return fe
