"""Host-side input handling: file / array / tensor / bytes -> mono float32 waveforms in [-1, 1] at the frontend rate.

Mirrors the subset of `load_audio_text_image_video` the ASR path uses (funasr/utils/load_utils.py:48-179): local WAV
paths (decoded with the stdlib `wave` module instead of torchaudio/soundfile/ffmpeg), numpy arrays, tensors, bytes, and
lists of those; a missing file raises FileNotFoundError like :95-112. A sample-rate mismatch is resampled on the host with
a polyphase FIR (scipy) where the reference uses torchaudio.transforms.Resample (:176-178).
Bytes follow `load_bytes` (:306-341): a recognised container (`is_audio_container`, restating `_is_audio_container`
:272-303 incl. the MPEG frame-chain test that keeps raw PCM starting with a sync-like sample from being mistaken for MP3) is
DECODED -- RIFF / RIFX WAVE here; compressed formats need torchaudio / soundfile / ffmpeg, which this package does not
depend on, and raise the reference's "complete supported audio file" error instead of being read as samples -- everything
else is headerless little-endian int16 PCM.
"""
from __future__ import annotations

import io
import os
import wave
from typing import List

import numpy as np
import torch


def _decode_wav_fast(path: str):
    """The canonical RIFF layout -- 'RIFF' size 'WAVE' 'fmt ' 16 <PCM, channels, rate, ..., bits> 'data' size samples, 44 header
    bytes -- read with one file read and no chunk walking (a corpus of short clips spends as long in the stdlib's chunk parser as
    in reading the samples). -> (raw bytes, rate, channels, sample width) or None for anything else (`wave` then decides)."""
    import struct
    with open(path, "rb") as f:
        data = f.read()
    if len(data) < 44 or data[:4] != b"RIFF" or data[8:16] != b"WAVEfmt " or data[36:40] != b"data":
        return None
    fmt_size, tag, ch, fs, _, align, bits = struct.unpack_from("<IHHIIHH", data, 16)
    n = struct.unpack_from("<I", data, 40)[0]
    if fmt_size != 16 or tag != 1 or ch < 1 or bits not in (8, 16, 32) or align != ch * bits // 8:
        return None
    n = min(n, len(data) - 44) // align * align                      # like wave.readframes: whole frames that are really there
    return data[44: 44 + n], fs, ch, bits // 8


def _decode_wav(src) -> tuple[torch.Tensor, int]:
    fast = _decode_wav_fast(src) if isinstance(src, str) else None
    if fast is not None:
        raw, fs, ch, width = fast
        return _pcm_to_float(raw, ch, width), fs
    with wave.open(src, "rb") as f:
        fs, ch, width, n = f.getframerate(), f.getnchannels(), f.getsampwidth(), f.getnframes()
        raw = f.readframes(n)
    return _pcm_to_float(raw, ch, width), fs


def _pcm_to_float(raw: bytes, ch: int, width: int) -> torch.Tensor:
    # (scaled in place by the power of two: bit for bit the division, one pass and one allocation less per clip)
    if width == 2:
        x = np.frombuffer(raw, dtype="<i2").astype(np.float32)
        x *= np.float32(2.0 ** -15)
    elif width == 4:
        x = np.frombuffer(raw, dtype="<i4").astype(np.float32)
        x *= np.float32(2.0 ** -31)
    elif width == 1:
        x = np.frombuffer(raw, dtype=np.uint8).astype(np.float32)
        x -= np.float32(128.0)
        x *= np.float32(2.0 ** -7)
    else:
        raise ValueError(f"unsupported PCM sample width {width}")
    if ch > 1:
        x = x.reshape(-1, ch).mean(axis=1)      # reduce_channels (load_utils.py:126-127)
    return torch.from_numpy(x if x.flags.c_contiguous else np.ascontiguousarray(x))


# ------------------------------------------------------------------------------------------- container sniffing
_MPEG1_KBPS = {3: (32, 64, 96, 128, 160, 192, 224, 256, 288, 320, 352, 384, 416, 448),     # layer I
               2: (32, 48, 56, 64, 80, 96, 112, 128, 160, 192, 224, 256, 320, 384),        # layer II
               1: (32, 40, 48, 56, 64, 80, 96, 112, 128, 160, 192, 224, 256, 320)}         # layer III
_MPEG2_KBPS = {3: (32, 48, 56, 64, 80, 96, 112, 128, 144, 160, 176, 192, 224, 256),
               2: (8, 16, 24, 32, 40, 48, 56, 64, 80, 96, 112, 128, 144, 160),
               1: (8, 16, 24, 32, 40, 48, 56, 64, 80, 96, 112, 128, 144, 160)}
_MPEG_HZ = {3: (44100, 48000, 32000), 2: (22050, 24000, 16000), 0: (11025, 12000, 8000)}


def _mpeg_header(data: bytes, off: int):
    """(version id, layer id, bitrate index, sample-rate index, padding bit) of a plausible MPEG audio frame header at `off`"""
    if off < 0 or off + 4 > len(data):
        return None
    word = int.from_bytes(data[off: off + 4], "big")
    if word >> 21 != 0x7FF:
        return None
    version, layer, bitrate, rate, pad = (word >> 19) & 3, (word >> 17) & 3, (word >> 12) & 15, (word >> 10) & 3, (word >> 9) & 1
    if version == 1 or layer == 0 or bitrate == 15 or rate == 3:
        return None
    return version, layer, bitrate, rate, pad


def _mpeg_frame_bytes(data: bytes, off: int) -> int:
    """length of a fixed-bitrate frame, 0 for no header / free format"""
    h = _mpeg_header(data, off)
    if h is None or h[2] == 0:
        return 0
    version, layer, bitrate, rate, pad = h
    bps = (_MPEG1_KBPS if version == 3 else _MPEG2_KBPS)[layer][bitrate - 1] * 1000
    hz = _MPEG_HZ[version][rate]
    if layer == 3:
        return (12 * bps // hz + pad) * 4
    return (144 if version == 3 or layer == 2 else 72) * bps // hz + pad


def _mpeg_frame_chain(data: bytes) -> bool:
    """two consecutive fixed-bitrate frames, or three equally spaced free-format headers of the same stream type"""
    first = _mpeg_header(data, 0)
    if first is None:
        return False
    if first[2] != 0:
        n = _mpeg_frame_bytes(data, 0)
        return n > 0 and _mpeg_frame_bytes(data, n) > 0
    kind = (first[0], first[1], first[3])
    slot = 4 if first[1] == 3 else 1
    for second in range(24, min(len(data) - 3, 8192)):
        h2 = _mpeg_header(data, second)
        if h2 is None or (h2[0], h2[1], h2[3]) != kind or h2[2] != 0:
            continue
        third = second + (second - first[4] * slot) + h2[4] * slot
        h3 = _mpeg_header(data, third)
        if h3 is not None and (h3[0], h3[1], h3[3]) == kind and h3[2] == 0:
            return True
    return False


def is_audio_container(data: bytes) -> bool:
    if len(data) < 4:
        return False
    if len(data) >= 12 and data[:4] in (b"RIFF", b"RIFX", b"RF64", b"BW64") and data[8:12] == b"WAVE":
        return True
    if data[:3] == b"ID3" or (data[0] == 0xFF and data[1] & 0xE0 == 0xE0 and _mpeg_frame_chain(data)):
        return True
    if data[:4] in (b"OggS", b"fLaC", b"\x1a\x45\xdf\xa3"):
        return True
    return len(data) >= 8 and data[4:8] == b"ftyp"


def _decode_rifx(data: bytes) -> tuple[torch.Tensor, int]:
    """big-endian WAVE (RIFX): PCM chunks parsed by hand, the stdlib `wave` module only reads RIFF"""
    import struct
    pos, fmt, pcm = 12, None, None
    while pos + 8 <= len(data):
        cid, size = data[pos: pos + 4], struct.unpack(">I", data[pos + 4: pos + 8])[0]
        body = data[pos + 8: pos + 8 + size]
        if cid == b"fmt ":
            fmt = struct.unpack(">HHIIHH", body[:16])
        elif cid == b"data":
            pcm = body
        pos += 8 + size + (size & 1)
    if fmt is None or pcm is None or fmt[0] != 1:
        raise ValueError("RIFX: no PCM fmt / data chunk")
    _, ch, fs, _, _, bits = fmt
    if bits == 16:
        x = np.frombuffer(pcm[: len(pcm) // 2 * 2], dtype=">i2").astype(np.float32) / 32768.0
    elif bits == 32:
        x = np.frombuffer(pcm[: len(pcm) // 4 * 4], dtype=">i4").astype(np.float32) / 2147483648.0
    elif bits == 8:
        x = (np.frombuffer(pcm, dtype=np.uint8).astype(np.float32) - 128.0) / 128.0
    else:
        raise ValueError(f"unsupported PCM sample width {bits} bits")
    if ch > 1:
        x = x.reshape(-1, ch).mean(axis=1)
    return torch.from_numpy(np.ascontiguousarray(x)), fs


def _decode_container(data: bytes) -> tuple[torch.Tensor, int]:
    try:
        if data[:4] == b"RIFF":
            return _decode_wav(io.BytesIO(data))
        if data[:4] == b"RIFX":
            return _decode_rifx(data)
        raise ValueError("no decoder for this container in funasr_amd (stdlib WAVE only)")
    except Exception as exc:  # noqa: BLE001
        raise RuntimeError("Failed to decode container-formatted audio bytes. Verify that the input is a complete supported "
                           "audio file and that torchaudio, soundfile, or ffmpeg is available.") from exc


def load_audio(item, fs: int = 16000, audio_fs: int = 16000) -> torch.Tensor:
    if isinstance(item, str):
        if not os.path.exists(item):
            raise FileNotFoundError(f"Audio file not found: {item!r}. Pass a valid local file path, numpy array, "
                                    f"torch.Tensor, or bytes.")
        x, audio_fs = _decode_wav(item)
    elif isinstance(item, (bytes, bytearray)):
        data = bytes(item)
        if is_audio_container(data):
            x, audio_fs = _decode_container(data)
        else:                                      # headerless 16-bit PCM (funasr/utils/load_utils.py:329-341)
            x = torch.from_numpy(np.frombuffer(data[: len(data) // 2 * 2], dtype="<i2").astype(np.float32) / 32768.0)
    elif isinstance(item, np.ndarray):
        x = torch.from_numpy(item)
    elif isinstance(item, torch.Tensor):
        x = item
    elif hasattr(item, "read"):
        x, audio_fs = _decode_wav(item)
    else:
        raise TypeError(f"unsupported audio input type {type(item)}")
    if x.dtype in (torch.int16,):                  # integer PCM -> [-1, 1) BEFORE anything casts it to float
        x = x.to(torch.float32) / 32768.0
    if x.dim() > 1:
        # [channels, n] or [n, channels]: the channel axis is the small one; anything else is ambiguous
        x = x.reshape(-1, x.shape[-1]) if x.dim() > 2 else x
        if x.shape[0] <= 8:
            x = x.to(torch.float32).mean(0)
        elif x.shape[1] <= 8:
            x = x.to(torch.float32).mean(1)
        else:
            raise ValueError(f"audio array of shape {tuple(x.shape)}: neither axis looks like a channel axis (<= 8)")
    if audio_fs != fs:
        x = resample(x.to(torch.float32), audio_fs, fs)
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


def batch_to_features(data_in, data_lengths, frontend, kwargs, uploader=None):
    """The head of every model's `inference` (funasr/models/paraformer/model.py:576-595, sense_voice/model.py:948-975): waveforms
    (paths / arrays / tensors / bytes) or ready features -> (speech, speech_lengths, meta_data). `uploader`
    (hip_module.StagedUpload): the padded batch goes to the device through pinned memory on an upload stream -- the form of
    `.to(device)` for loops that overlap batches."""
    import time
    meta_data = {}
    device = kwargs.get("device", None)
    if isinstance(data_in, torch.Tensor) and kwargs.get("data_type", "sound") == "fbank":
        speech, speech_lengths = data_in, data_lengths
        if speech.dim() < 3:
            speech = speech[None]
        if speech_lengths is None:
            speech_lengths = [speech.shape[1]] * speech.shape[0]
        return speech, speech_lengths, meta_data
    t1 = time.perf_counter()
    audio = load_audio_list(data_in, fs=frontend.fs, audio_fs=kwargs.get("fs", 16000))
    t2 = time.perf_counter()
    meta_data["load_data"] = f"{t2 - t1:0.3f}"
    if (uploader is not None and device is not None and str(device).startswith("cuda") and torch.cuda.is_available()
            and all(a.device.type == "cpu" for a in audio)):
        wav, lens = uploader(audio, device)
    else:
        lens = [int(a.shape[0]) for a in audio]
        wav = torch.nn.utils.rnn.pad_sequence(audio, batch_first=True)      # load_utils.py:413
        if device is not None:
            wav = wav.to(device)
    speech, speech_lengths = frontend(wav, lens)
    t3 = time.perf_counter()
    meta_data["extract_feat"] = f"{t3 - t2:0.3f}"
    meta_data["batch_data_time"] = (int(speech_lengths.sum().item()) * frontend.frame_shift * frontend.lfr_n / 1000)
    return speech, speech_lengths, meta_data


def peek_num_samples(item, fs: int = 16000):
    """How many samples `load_audio(item, fs)` will return, WITHOUT decoding: canonical RIFF header of a path, length of a 1-D array
    or tensor; None where only decoding can tell (other containers, bytes, multi-channel arrays). For planning batches by length."""
    import struct
    if isinstance(item, str):
        try:
            with open(item, "rb") as f:
                h = f.read(44)
        except OSError:
            return None
        if len(h) < 44 or h[:4] != b"RIFF" or h[8:16] != b"WAVEfmt " or h[36:40] != b"data":
            return None
        fmt_size, tag, ch, rate, _, align, bits = struct.unpack_from("<IHHIIHH", h, 16)
        if fmt_size != 16 or tag != 1 or ch < 1 or align != ch * bits // 8 or align == 0 or rate <= 0:
            return None
        try:
            n = min(struct.unpack_from("<I", h, 40)[0], max(os.path.getsize(item) - 44, 0)) // align
        except OSError:
            return None
        return n if rate == fs else int(round(n * fs / rate))
    if isinstance(item, (np.ndarray, torch.Tensor)) and item.ndim == 1:
        return int(item.shape[0])
    return None
