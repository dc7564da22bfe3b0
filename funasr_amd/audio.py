"""Host-side input handling: file / array / tensor / bytes -> mono float32 waveforms in [-1, 1] at the frontend rate.

Mirrors the subset of `load_audio_text_image_video` the ASR path uses (funasr/utils/load_utils.py:48-179): local WAV
paths (decoded with the stdlib `wave` module instead of torchaudio/soundfile/ffmpeg), numpy arrays, tensors, raw
16-bit PCM bytes, and lists of those; a missing file raises FileNotFoundError like :95-112. A sample-rate mismatch is
resampled on the host with a polyphase FIR (scipy) where the reference uses torchaudio.transforms.Resample (:176-178).
"""
from __future__ import annotations

import io
import os
import wave
from typing import List, Sequence, Union

import numpy as np
import torch


def _decode_wav(src) -> tuple[torch.Tensor, int]:
    with wave.open(src, "rb") as f:
        fs, ch, width, n = f.getframerate(), f.getnchannels(), f.getsampwidth(), f.getnframes()
        raw = f.readframes(n)
    if width == 2:
        x = np.frombuffer(raw, dtype="<i2").astype(np.float32) / 32768.0
    elif width == 4:
        x = np.frombuffer(raw, dtype="<i4").astype(np.float32) / 2147483648.0
    elif width == 1:
        x = (np.frombuffer(raw, dtype=np.uint8).astype(np.float32) - 128.0) / 128.0
    else:
        raise ValueError(f"unsupported PCM sample width {width}")
    if ch > 1:
        x = x.reshape(-1, ch).mean(axis=1)      # reduce_channels (load_utils.py:126-127)
    return torch.from_numpy(np.ascontiguousarray(x)), fs


def load_audio(item, fs: int = 16000, audio_fs: int = 16000) -> torch.Tensor:
    if isinstance(item, str):
        if not os.path.exists(item):
            raise FileNotFoundError(f"Audio file not found: {item!r}. Pass a valid local file path, numpy array, "
                                    f"torch.Tensor, or bytes.")
        x, audio_fs = _decode_wav(item)
    elif isinstance(item, (bytes, bytearray)):
        if item[:4] == b"RIFF":
            x, audio_fs = _decode_wav(io.BytesIO(bytes(item)))
        else:                                      # headerless 16-bit PCM (funasr/utils/load_utils.py:load_bytes)
            x = torch.from_numpy(np.frombuffer(bytes(item), dtype="<i2").astype(np.float32) / 32768.0)
    elif isinstance(item, np.ndarray):
        x = torch.from_numpy(item)
    elif isinstance(item, torch.Tensor):
        x = item
    elif hasattr(item, "read"):
        x, audio_fs = _decode_wav(item)
    else:
        raise TypeError(f"unsupported audio input type {type(item)}")
    if x.dim() > 1:
        x = x.reshape(-1, x.shape[-1]).mean(0) if x.shape[0] <= 8 else x.reshape(-1)
    if audio_fs != fs:
        x = resample(x.to(torch.float32), audio_fs, fs)
    if x.dtype in (torch.int16,):
        x = x.to(torch.float32) / 32768.0
    return x.to(torch.float32)


def resample(x: torch.Tensor, src_fs: int, dst_fs: int) -> torch.Tensor:
    """Host-side band-limited resampling (the reference calls torchaudio.transforms.Resample, load_utils.py:176-178;
    torchaudio is not a dependency here): polyphase FIR from scipy. Not bit-identical to torchaudio's kernel -- audio that
    is already at the frontend rate (the hot path) never passes through here."""
    from math import gcd

    from scipy.signal import resample_poly

    g = gcd(int(src_fs), int(dst_fs))
    y = resample_poly(x.numpy().astype(np.float64), int(dst_fs) // g, int(src_fs) // g)
    return torch.from_numpy(np.ascontiguousarray(y.astype(np.float32)))


def load_audio_list(data_in, fs: int = 16000, audio_fs: int = 16000) -> List[torch.Tensor]:
    if isinstance(data_in, (list, tuple)):
        return [load_audio(d, fs, audio_fs) for d in data_in]
    if isinstance(data_in, torch.Tensor) and data_in.dim() == 2:
        return [row for row in data_in]
    return [load_audio(data_in, fs, audio_fs)]
