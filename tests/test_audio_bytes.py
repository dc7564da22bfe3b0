"""Bytes input (`load_bytes`, funasr/utils/load_utils.py:272-341): the scenarios of the reference's
tests/test_load_audio_bytes.py that do not need a compressed-audio decoder, replayed against funasr_amd.audio."""
import io
import struct
import wave

import numpy as np
import pytest

from funasr_amd.audio import is_audio_container, load_audio


def _sine_pcm(fs, duration=0.1):
    t = np.arange(round(fs * duration), dtype=np.float64) / fs
    return np.round(np.sin(2 * np.pi * 440 * t) * 12000).astype(np.int16)


def _wav_bytes(samples, fs):
    out = io.BytesIO()
    with wave.open(out, "wb") as f:
        f.setnchannels(1)
        f.setsampwidth(2)
        f.setframerate(fs)
        f.writeframes(samples.tobytes())
    return out.getvalue()


def _rifx_bytes(samples, fs):
    pcm = samples.astype(">i2").tobytes()
    return (b"RIFX" + struct.pack(">I", 36 + len(pcm)) + b"WAVEfmt " + struct.pack(">IHHIIHH", 16, 1, 1, fs, fs * 2, 2, 16)
            + b"data" + struct.pack(">I", len(pcm)) + pcm)


def _mp3_like(frames=3, bitrate_index=9, free_format=False):
    """structurally valid MPEG-1 layer III frame headers (44.1 kHz) at the right spacing, zero payload"""
    if free_format:
        header, length = bytes([0xFF, 0xFB, 0x00, 0x00]), 200
    else:
        header = bytes([0xFF, 0xFB, bitrate_index << 4, 0x00])
        length = 144 * (32, 40, 48, 56, 64, 80, 96, 112, 128, 160, 192, 224, 256, 320)[bitrate_index - 1] * 1000 // 44100
    return b"".join(header + bytes(length - 4) for _ in range(frames))


def test_wav_containers_are_decoded_not_read_as_samples():
    s = _sine_pcm(16000)
    x = load_audio(_wav_bytes(s, 16000)).numpy()
    assert x.dtype == np.float32 and np.allclose(x, s.astype(np.float32) / 32768.0, atol=1e-6)
    y = load_audio(_wav_bytes(_sine_pcm(8000), 8000)).numpy()                  # 8 kHz container -> 16 kHz
    assert len(y) == 1600 and np.isfinite(y).all() and float(np.abs(y).max()) > 0.1
    be = np.array([-32768, -1000, 0, 1000, 32767], dtype=np.int16)
    assert np.array_equal(load_audio(_rifx_bytes(be, 16000)).numpy(), be.astype(np.float32) / 32768.0)
    for marker in (b"RF64", b"BW64"):                                           # recognised as containers (64-bit WAVE)
        assert is_audio_container(marker + b"\xff\xff\xff\xffWAVEplaceholder")


def test_raw_pcm_is_preserved_even_when_it_looks_like_a_header():
    for raw in (np.array([-32768, -12345, 0, 12345, 32767], dtype=np.int16).tobytes(),
                b"\xff\xfb\x00\x00\x39\x30\xc7\xcf",                            # first sample looks like an MPEG sync word
                b"RIFF\x00\x00\x00\x00NOPE\x00\x00\x00\x00"):                   # RIFF, but not WAVE
        assert not is_audio_container(raw)
        assert np.array_equal(load_audio(raw).numpy(), np.frombuffer(raw, dtype=np.int16).astype(np.float32) / 32768.0)
    raw = bytearray(np.arange(160, dtype=np.int16).tobytes())                  # sync-like words at inconsistent distances
    for off in (0, 100, 210):
        raw[off: off + 4] = b"\xff\xfb\x00\x00"
    raw = bytes(raw)
    assert np.array_equal(load_audio(raw).numpy(), np.frombuffer(raw, dtype=np.int16).astype(np.float32) / 32768.0)


def test_compressed_containers_never_fall_back_to_raw_pcm():
    for blob in (_mp3_like(), _mp3_like(free_format=True), b"ID3\x03\x00" + bytes(40), b"OggS" + bytes(40), b"fLaC" + bytes(40),
                 bytes(4) + b"ftypisom" + bytes(20), b"\x1a\x45\xdf\xa3" + bytes(20), b"RIFF\x10\x00\x00\x00WAVEbroken"):
        assert is_audio_container(blob)
        with pytest.raises(RuntimeError, match="complete supported audio file"):
            load_audio(blob)
    assert not is_audio_container(_mp3_like(frames=1)) and not is_audio_container(b"\xff\xfb")


def test_int16_arrays_are_scaled_before_resampling_and_channels_are_averaged():
    """integer PCM must come out in [-1, 1] whether or not it is resampled on the way; a 2-D array is [channels, n] or
    [n, channels] (the small axis is the channel axis), anything else is refused"""
    import numpy as np
    import torch
    from funasr_amd.audio import load_audio
    t = np.arange(8000) / 8000.0
    pcm = (np.sin(2 * np.pi * 220 * t) * 10000).astype(np.int16)
    same = load_audio(pcm, fs=8000, audio_fs=8000)
    up = load_audio(pcm, fs=16000, audio_fs=8000)
    assert abs(float(same.abs().max()) - 10000 / 32768.0) < 1e-3
    assert up.numel() == 16000 and abs(float(up.abs().max()) - 10000 / 32768.0) < 2e-2
    st = np.stack([pcm, pcm // 2], 0)                                    # [channels, n]
    a = load_audio(st, fs=8000, audio_fs=8000)
    b = load_audio(np.ascontiguousarray(st.T), fs=8000, audio_fs=8000)   # [n, channels]
    assert a.shape == (8000,) and torch.allclose(a, b) and abs(float(a.abs().max()) - 7500 / 32768.0) < 1e-3
    import pytest
    with pytest.raises(ValueError):
        load_audio(np.zeros((100, 100), np.float32))


def test_canonical_wav_fast_reader_equals_the_wave_module(tmp_path):
    """funasr_amd/audio.py _decode_wav_fast (one read, no chunk walking) against the stdlib path it short-cuts: mono / stereo,
    8 / 16 / 32 bit, a data chunk that announces more than the file holds, and the layouts it must hand back to `wave`"""
    import struct
    import wave

    import numpy as np
    import torch

    from funasr_amd import audio

    def via_wave(path):
        with wave.open(path, "rb") as f:
            return audio._pcm_to_float(f.readframes(f.getnframes()), f.getnchannels(), f.getsampwidth()), f.getframerate()

    rng = np.random.default_rng(3)
    k = 0
    for ch in (1, 2):
        for width, dtype in ((1, np.uint8), (2, "<i2"), (4, "<i4")):
            for n in (0, 1, 1601):
                p = str(tmp_path / f"a{k}.wav")
                k += 1
                info = np.iinfo(dtype)
                pcm = rng.integers(info.min, info.max, size=n * ch, endpoint=True).astype(dtype)
                with wave.open(p, "wb") as f:
                    f.setnchannels(ch)
                    f.setsampwidth(width)
                    f.setframerate(8000 if n == 1 else 16000)
                    f.writeframes(pcm.tobytes())
                assert audio._decode_wav_fast(p) is not None
                x, fs = audio._decode_wav(p)
                y, fs2 = via_wave(p)
                assert fs == fs2 and x.dtype == torch.float32 and torch.equal(x, y), (ch, width, n)
    # truncated file: the data chunk announces 4000 bytes, 1001 are there -> whole frames only, like wave.readframes
    p = str(tmp_path / "cut.wav")
    body = rng.integers(-3000, 3000, size=2000).astype("<i2").tobytes()
    hdr = b"RIFF" + struct.pack("<I", 36 + 4000) + b"WAVEfmt " + struct.pack("<IHHIIHH", 16, 1, 1, 16000, 32000, 2, 16) + b"data" + struct.pack("<I", 4000)
    open(p, "wb").write(hdr + body[:1001])
    assert audio._decode_wav(p)[0].numel() == 500
    # a LIST chunk before the data, a 24-bit file, an extensible fmt chunk: not the canonical layout -> the stdlib decides
    p2 = str(tmp_path / "list.wav")
    open(p2, "wb").write(b"RIFF" + struct.pack("<I", 36 + 12 + 8 + 200) + b"WAVEfmt " + struct.pack("<IHHIIHH", 16, 1, 1, 16000, 32000, 2, 16)
                         + b"LIST" + struct.pack("<I", 4) + b"INFO" + b"data" + struct.pack("<I", 200) + body[:200])
    assert audio._decode_wav_fast(p2) is None
    x, fs = audio._decode_wav(p2)
    assert fs == 16000 and x.numel() == 100 and torch.equal(x, torch.from_numpy(np.frombuffer(body[:200], dtype="<i2").astype(np.float32) / 32768.0))
    p3 = str(tmp_path / "w24.wav")
    with wave.open(p3, "wb") as f:
        f.setnchannels(1)
        f.setsampwidth(3)
        f.setframerate(16000)
        f.writeframes(b"\x00" * 30)
    assert audio._decode_wav_fast(p3) is None
    # the header peek used for planning batches by length agrees with what decoding returns
    for q in sorted(tmp_path.glob("a*.wav")):
        x, fs = audio._decode_wav(str(q))
        assert audio.peek_num_samples(str(q), fs) == x.numel()
    assert audio.peek_num_samples(p2, 16000) is None and audio.peek_num_samples(torch.zeros(7), 16000) == 7
    assert audio.peek_num_samples(torch.zeros(2, 7), 16000) is None and audio.peek_num_samples(b"1234", 16000) is None
