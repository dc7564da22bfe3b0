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


def test_streaming_inference_chunk_loop_on_random_call_splits(frontend, monkeypatch):
    """ParaformerStreaming.inference (product mirror) cut into random calls: which pieces reach the frontend, when the look-ahead
    is flushed, when the < 960-sample tail chunk re-feeds the cached window, what is carried between calls -- against the
    oracle's streaming_inference (the reference's loop, paraformer_streaming/model.py:688-752) with the network stubbed out on
    both sides (the recorded decisions and features must agree)."""
    from funasr_amd.paraformer_streaming import ParaformerStreaming
    fe, cmvn = frontend
    g = torch.Generator().manual_seed(23)
    cfg = synth.tiny(synth.PARAFORMER_LARGE, enc_blocks=1, dec_blocks=1, vocab=30)
    model = ParaformerStreaming.from_config(cfg)

    class FakeStream:
        def __init__(self, keep):
            self.keep = keep

        def close(self):
            pass

    for si in range(40):
        chunk = [int(torch.randint(0, 2, (1,), generator=g)) * 5, int(torch.randint(4, 15, (1,), generator=g)), 0]
        chunk[2] = int(torch.randint(1, chunk[1] // 2 + 2, (1,), generator=g))
        stride = chunk[1] * 960
        n_total = int(torch.randint(stride // 2, 5 * stride, (1,), generator=g))
        if si % 5 == 0:
            n_total = (n_total // stride + 1) * stride + int(torch.randint(0, 959, (1,), generator=g))     # a tail shorter than 960 samples
        wav = synth.speech_like(n_total + 1, seed=100 + si)[:n_total]
        cuts = sorted(set(int(v) for v in torch.randint(1, n_total, (int(torch.randint(0, 4, (1,), generator=g)),), generator=g)))
        bounds = [0] + cuts + [n_total]
        got, want = [], []

        def init_cache(cache=None, **kw):
            cache = {} if cache is None else cache
            cache["_stream"] = FakeStream(chunk[0] + chunk[2])
            cache["encoder"] = {"tail_chunk": False, "chunk_size": chunk}
            cache["decoder"], cache["frontend"], cache["prev_samples"] = {}, {}, torch.empty(0)
            return cache

        def fake_generate(speech, speech_lengths=None, **kw):
            c = kw["cache"]
            got.append((None if c["encoder"]["tail_chunk"] else speech[0].clone(), bool(kw.get("is_final")), bool(c["encoder"]["tail_chunk"])))
            return []

        monkeypatch.setattr(model, "init_cache", init_cache)
        monkeypatch.setattr(model, "generate_chunk", fake_generate)

        def oracle_generate(feats, st, sd, cfg_, is_final, trace=None):
            want.append((None if st["tail_chunk"] else feats[0].clone(), bool(is_final), bool(st["tail_chunk"])))
            return []

        monkeypatch.setattr(S, "generate_chunk", oracle_generate)
        st = S.model_init(cfg, tuple(chunk), 1, 1)
        cache = {}
        for ci in range(len(bounds) - 1):
            piece = wav[bounds[ci]:bounds[ci + 1]]
            fin = ci == len(bounds) - 2
            model.inference([piece], key=["u"], tokenizer=None, frontend=fe, cache=cache, is_final=fin, chunk_size=chunk,
                            encoder_chunk_look_back=1, decoder_chunk_look_back=1)
            S.streaming_inference(piece, st, {}, cfg, cmvn, fin)
            if not fin:
                assert torch.equal(cache["prev_samples"], st["prev_samples"]), (si, ci)
        assert len(got) == len(want), (si, chunk, bounds, len(got), len(want))
        for k, (a, b) in enumerate(zip(got, want)):
            assert a[1:] == b[1:], (si, k, a[1:], b[1:])
            assert (a[0] is None and b[0] is None) or torch.equal(a[0], b[0]), (si, k)
