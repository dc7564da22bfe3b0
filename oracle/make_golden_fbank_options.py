"""Golden fbank outputs for WavFrontend's `window` / `snip_edges` options (funasr/frontends/wav_frontend.py:171-181:
kaldi.fbank(window_type=self.window, snip_edges=self.snip_edges)). TEST INFRASTRUCTURE ONLY.

torchaudio is absent from this image, so -- as for the default options in make_golden.py -- the reference side is the
kaldi-native-fbank the reference vendors (runtime/onnxruntime/third_party/kaldi-native-fbank), compiled in place by
oracle/Makefile into oracle/_ref/libknf_ref.so and driven through oracle/knf_shim.cc:knf_fbank_opts. Inputs are the two PCM
clips of tests/golden/frontend.npz (real speech from the reference's asr_example.wav and a synthetic clip).

    python oracle/make_golden_fbank_options.py        # writes tests/golden/fbank_options.npz
"""
import ctypes
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
GOLD = os.path.join(os.path.dirname(HERE), "tests", "golden")
WINDOWS = ("hamming", "hanning", "povey", "rectangular", "blackman")


def knf_fbank_opts(wave_scaled: np.ndarray, window_type: str, snip_edges: bool, n_mels: int = 80, frame_length_ms: float = 25.0) -> np.ndarray:
    lib = ctypes.CDLL(os.path.join(HERE, "_ref", "libknf_ref.so"))
    lib.knf_fbank_opts.restype = ctypes.c_int
    lib.knf_fbank_opts.argtypes = [ctypes.c_void_p, ctypes.c_int64, ctypes.c_int, ctypes.c_float, ctypes.c_int, ctypes.c_float,
                                   ctypes.c_char_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_int64]
    w = np.ascontiguousarray(wave_scaled, dtype=np.float32)
    max_frames = len(w) // 160 + 2
    out = np.zeros((max_frames, n_mels), dtype=np.float32)
    n = lib.knf_fbank_opts(w.ctypes.data, len(w), n_mels, frame_length_ms, 10, 16000.0, window_type.encode(), int(snip_edges),
                           out.ctypes.data, max_frames)
    return out[:n]


def main():
    g = np.load(os.path.join(GOLD, "frontend.npz"))
    arrs = {}
    for k in ("a", "b"):
        wave = g[f"pcm_{k}"].astype(np.float32) / 32768.0 * 32768.0
        for w in WINDOWS:
            for snip in (True, False):
                if w == "hamming" and snip:
                    continue                                   # the default: tests/golden/frontend.npz
                arrs[f"{k}_{w}_{'snip' if snip else 'nosnip'}"] = knf_fbank_opts(wave, w, snip)
    # short inputs of snip_edges = False: fewer samples than a window (frames mirror several times)
    arrs["short_pcm"] = g["pcm_a"][3000:3000 + 250]
    arrs["short_hamming_nosnip"] = knf_fbank_opts(arrs["short_pcm"].astype(np.float32), "hamming", False)
    # clips shorter than one 25-ms window: WavFrontend analyses them with ONE window of their own length (wav_frontend.py:176,
    # frame_length = n / 16 ms; the FFT size follows: 512 down to 64 points here)
    for n in (399, 300, 257, 256, 200, 129, 100, 64, 33):
        pcm = g["pcm_a"][5000:5000 + n]
        arrs[f"tiny_{n}_pcm"] = pcm
        arrs[f"tiny_{n}_hamming_snip"] = knf_fbank_opts(pcm.astype(np.float32), "hamming", True, frame_length_ms=n / 16.0)
        assert arrs[f"tiny_{n}_hamming_snip"].shape == (1, 80), arrs[f"tiny_{n}_hamming_snip"].shape
    arrs["tiny_300_povey_snip"] = knf_fbank_opts(arrs["tiny_300_pcm"].astype(np.float32), "povey", True, frame_length_ms=300 / 16.0)
    path = os.path.join(GOLD, "fbank_options.npz")
    np.savez_compressed(path, **arrs)
    print(f"wrote {path} ({os.path.getsize(path) / 1024:.1f} KB): " + ", ".join(f"{k}{v.shape}" for k, v in arrs.items()))


if __name__ == "__main__":
    main()
