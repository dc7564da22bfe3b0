"""FSMN-VAD decision logic against golden vectors from the reference's own FsmnVADStreaming.forward driven with injected
network scores (oracle/make_golden_vad.py): same segments / streaming events, block by block, in every scenario."""
import json
import os

import numpy as np

from funasr_amd.vad_decision import VadDecision, VadOptions

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "vad_decision.npz")


def test_segments_equal_reference_block_by_block():
    g = np.load(GOLD, allow_pickle=False)
    cases = json.loads(str(g["meta"]))["cases"]
    p_all, db_all = g["p_sil"], g["decibel"]
    modes, n_events, off = set(), 0, 0
    for ci, c in enumerate(cases):
        p_sil, decibel = p_all[off: off + c["n"]].tolist(), db_all[off: off + c["n"]].tolist()
        off += c["n"]
        dec = VadDecision(VadOptions(**c["options"]))
        f0 = 0
        modes.add(c["mode"])
        for bi, (b, want) in enumerate(zip(c["blocks"], c["segments_per_block"])):
            got = dec.push(p_sil[f0: f0 + b], decibel[f0: f0 + b], is_final=bi == len(c["blocks"]) - 1,
                           streaming_events=c["mode"] == "stream_events")
            assert got == want, (ci, c["mode"], c["options"], bi, got, want)
            n_events += len(got)
            f0 += b
    assert len(cases) == 72 and modes == {"offline_one", "offline_blocks", "stream_events"} and n_events > 150
