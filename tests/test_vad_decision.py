"""FSMN-VAD decision logic against golden vectors from the reference's own FsmnVADStreaming.forward driven with injected
network scores (oracle/make_golden_vad.py): same segments / streaming events, block by block, in every scenario."""
import json
import os

import numpy as np

import pytest

from funasr_amd.vad_decision import NativeVadDecision, VadDecision, VadOptions

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "vad_decision.npz")


@pytest.mark.parametrize("impl", [VadDecision, NativeVadDecision], ids=["python", "native"])
def test_segments_equal_reference_block_by_block(impl):
    g = np.load(GOLD, allow_pickle=False)
    cases = json.loads(str(g["meta"]))["cases"]
    p_all, db_all = g["p_sil"], g["decibel"]
    modes, n_events, off = set(), 0, 0
    for ci, c in enumerate(cases):
        p_sil, decibel = p_all[off: off + c["n"]].tolist(), db_all[off: off + c["n"]].tolist()
        off += c["n"]
        dec = impl(VadOptions(**c["options"]))
        f0 = 0
        modes.add(c["mode"])
        for bi, (b, want) in enumerate(zip(c["blocks"], c["segments_per_block"])):
            got = dec.push(p_sil[f0: f0 + b], decibel[f0: f0 + b], is_final=bi == len(c["blocks"]) - 1,
                           streaming_events=c["mode"] == "stream_events")
            assert got == want, (ci, c["mode"], c["options"], bi, got, want)
            n_events += len(got)
            f0 += b
    assert len(cases) == 72 and modes == {"offline_one", "offline_blocks", "stream_events"} and n_events > 150



def test_native_equals_python_on_random_streams_with_schedule_changes():
    rng = np.random.default_rng(5)
    for trial in range(40):
        opts = VadOptions(do_extend=int(rng.integers(0, 2)), max_end_silence_time=int(rng.choice([300, 800, 2000])),
                          max_single_segment_time=int(rng.choice([2000, 60000])), decibel_thres=float(rng.choice([-100.0, -30.0])))
        a, b = VadDecision(opts), NativeVadDecision(opts)
        events = bool(trial % 2)
        n_blocks = int(rng.integers(1, 12))
        for bi in range(n_blocks):
            n = int(rng.integers(1, 400))
            speechy = rng.random(n) < (0.7 if (bi // 2) % 2 else 0.2)
            p = np.where(speechy, rng.uniform(0.01, 0.3, n), rng.uniform(0.6, 0.99, n)).astype(np.float32)
            db = np.where(speechy, rng.uniform(-25, -5, n), rng.uniform(-70, -35, n)).astype(np.float32)
            if rng.random() < 0.3:                                     # the per-block schedule of FsmnVADStreaming.inference
                v = float(rng.choice([50, 250, 850, 1850]))
                a.max_end_sil_ms = b.max_end_sil_ms = v
                a.speech_noise_thres = b.speech_noise_thres = 0.5
            final = bi == n_blocks - 1
            assert a.push(p.tolist(), db.tolist(), final, events) == b.push(p, db, final, events), (trial, bi)
            assert a.state == b.state
