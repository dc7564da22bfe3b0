"""The HOST half of the product's WavFrontendOnline (funasr_amd/paraformer_streaming.py: left-over samples, LFR splice cache,
row / splice arithmetic, the final flush with and without new frames) on random piece sequences, with the two device calls
(`pf_frontend_fbank`, `pf_frontend_lfr_cmvn`) replaced by stand-ins that compute the same arithmetic with the CPU oracle
through the raw pointers the product passes -- so the bookkeeping runs without a GPU. Expected values:
`streaming_oracle.frontend_step`, itself identical to the REFERENCE's WavFrontendOnline.forward on 300 random sessions
(oracle/fuzz_streaming_frontend_vs_reference.py). The GPU tests (tests/test_streaming_gpu.py) cover the kernels on the
9600-sample strides of ParaformerStreaming.inference only; this covers piece lengths from a few samples to several chunks."""
import contextlib
import ctypes as C
import os

import numpy as np
import pytest
import torch

from funasr_amd import synth
from oracle import paraformer_oracle as O
from oracle import streaming_oracle as S

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _view(ptr, n):
    return np.ctypeslib.as_array((C.c_float * n).from_address(ptr))


class _FakeLib:
    """pf_frontend_fbank / pf_frontend_lfr_cmvn with the oracle's arithmetic on host memory"""

    def __init__(self, cmvn):
        self.cmvn = cmvn

    def pf_frontend_fbank(self, h, wav_ptr, n, fb_ptr, stream):
        x = torch.from_numpy(_view(wav_ptr, n).copy())
        fb = O.kaldi_fbank(x * (1 << 15), 80, 25.0, 10.0, 16000.0)
        _view(fb_ptr, fb.numel())[:] = fb.reshape(-1).numpy()
        return 0

    def pf_frontend_lfr_cmvn(self, h, frames_ptr, T, rows, out_ptr, stream):
        frames = torch.from_numpy(_view(frames_ptr, T * 80).copy()).view(T, 80)
        idx = (torch.arange(rows)[:, None] * 6 + torch.arange(7)[None, :]).clamp(max=T - 1)
        out = O.apply_cmvn(frames[idx].reshape(rows, 560), self.cmvn)
        _view(out_ptr, rows * 560)[:] = out.reshape(-1).numpy()
        return 0


@pytest.fixture()
def frontend(monkeypatch):
    from funasr_amd import paraformer_streaming as PS
    cmvn = O.load_cmvn(os.path.join(GOLD, "am.mvn"))
    fe = PS.WavFrontendOnline(cmvn_file=os.path.join(GOLD, "am.mvn"), lfr_m=7, lfr_n=6, dither=0.0)
    fake = _FakeLib(cmvn)
    monkeypatch.setattr(fe, "_target_device", lambda x: torch.device("cpu"))
    monkeypatch.setattr(fe, "_ensure_handle", lambda dev: (fake, 0))
    monkeypatch.setattr(PS, "stream_ptr", lambda: 0)
    monkeypatch.setattr(torch.cuda, "device", lambda dev: contextlib.nullcontext())
    return fe, cmvn


def test_online_frontend_bookkeeping_on_random_piece_sequences(frontend):
    fe, cmvn = frontend
    g = torch.Generator().manual_seed(11)
    calls = rows = empty = 0
    for si in range(60):
        n_pieces = int(torch.randint(1, 8, (1,), generator=g))
        kind = si % 4
        lens = []
        for _ in range(n_pieces):
            if kind == 0:
                lens.append(int(torch.randint(1, 700, (1,), generator=g)))          # mostly too short for a frame / an LFR row
            elif kind == 1:
                lens.append(int(torch.randint(300, 4000, (1,), generator=g)))
            elif kind == 2:
                lens.append(960 * int(torch.randint(1, 21, (1,), generator=g)))     # ParaformerStreaming.inference strides
            else:
                lens.append(int(torch.randint(1, 20000, (1,), generator=g)))
        wav = synth.speech_like(sum(lens) + 1, seed=si)[:sum(lens)]
        cache, oc = {}, S.frontend_init()
        off = 0
        for pi, n in enumerate(lens):
            piece = wav[off:off + n]
            off += n
            fin = pi == n_pieces - 1
            got, lens_out = fe(piece[None].clone(), None, cache=cache, is_final=fin)
            want = S.frontend_step(piece.clone(), oc, cmvn, fin)
            calls += 1
            if want.shape[0] == 0:
                empty += 1
                assert got.numel() == 0, (si, pi, lens, fin)
                continue
            rows += want.shape[0]
            assert tuple(got.shape) == (1, want.shape[0], 560) and int(lens_out[0]) == want.shape[0], (si, pi, lens, fin)
            assert torch.equal(got[0], want), (si, pi, lens, fin, (got[0] - want).abs().max().item())
            # the carried state agrees too: left-over samples and splice frames
            assert torch.equal(cache["input_cache"], oc["input_cache"]), (si, pi)
            assert torch.equal(cache["lfr_splice_cache"], oc["lfr_splice_cache"]), (si, pi)
    assert calls > 150 and rows > 1000 and empty > 20          # the sweep reached both the productive and the starved branches


def test_online_frontend_rejects_batches(frontend):
    fe, _ = frontend
    with pytest.raises(ValueError, match="batch size"):
        fe(torch.zeros(2, 1600), None, cache={}, is_final=False)
