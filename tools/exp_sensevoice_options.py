#!/usr/bin/env python3
"""SenseVoiceSmall 128 x 10 s (M = 22 528 rows): encoder schedule options A/B, interleaved in one process.
usage: exp_sensevoice_options.py [steps]"""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from funasr_amd import synth
from funasr_amd.sense_voice import SenseVoiceSmall
from funasr_amd.wav_frontend import WavFrontend

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 6
dev = torch.device("cuda:0")
sh, sc = synth.synthetic_cmvn(560)
cfg = synth.SENSEVOICE_SMALL
m = SenseVoiceSmall.from_config(cfg); m.load_state_dict(synth.sensevoice_state_dict(cfg, seed=0), strict=False); m = m.to(dev); m.set_precision("f16x2")
fe = WavFrontend(cmvn=torch.stack([sh, sc]), lfr_m=7, lfr_n=6, dither=0.0, device=dev)
B, n = 128, 160000
wav = torch.stack([synth.speech_like(n, seed=500 + i) for i in range(B)]).to(dev)
lens = [n] * B

def run(k):
    pend = None
    for _ in range(k):
        f, fl = fe(wav, lens)
        nxt = m.enqueue_features(f, fl, "auto", "woitn")
        if pend is not None:
            m.collect(pend)
        pend = nxt
    return m.collect(pend)

DEFAULTS = {"w2_row": 2, "w2_tile": 7, "gemm_tile": 0, "row_sched": 0}
VARIANTS = [("default", {}), ("w_2 as tile GEMM (w4) + LayerNorm launch", {"w2_row": 0}), ("w_2 row form forced", {"w2_row": 1}),
            ("w_2 tile, eight-wave 256x256", {"w2_row": 0, "w2_tile": 0}), ("w_1 / QKV on the four-wave kernel", {"gemm_tile": 7}),
            ("w_2 tile + four-wave w_1 / QKV", {"w2_row": 0, "gemm_tile": 7})]
ref, out = None, {}
for rep in range(2):
    for name, opts in VARIANTS:
        for k, v in {**DEFAULTS, **opts}.items():
            m.encoder.set_option(k, v)
        run(2); torch.cuda.synchronize()
        t0 = time.perf_counter(); r = run(steps); torch.cuda.synchronize(); dt = time.perf_counter() - t0
        if ref is None:
            ref = r["ids"]
        out.setdefault(name, []).append({"audio_s_per_s": round(B * 10.0 * steps / dt, 1), "ms": round(dt / steps * 1e3, 2), "ids_equal_default": r["ids"] == ref})
print(json.dumps(out, indent=1))
