#!/usr/bin/env python3
"""Does a second model replica on its own HIP stream fill the CUs a kernel's last round leaves idle?
SenseVoiceSmall at 128 x 10 s runs M = 22 528 rows = 88 x 256: every power-of-two tiling of its N = 512 / 1536 projections fills
11/16 of the chip. R replicas (own handles and workspaces, own streams), each software-pipelined like bench.py's loop, batches dealt
round-robin; ids compared with the one-replica run.   usage: exp_two_streams.py [sensevoice|paraformer] [steps]"""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from funasr_amd import synth
from funasr_amd.wav_frontend import WavFrontend

which = sys.argv[1] if len(sys.argv) > 1 else "sensevoice"
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 8
dev = torch.device("cuda:0")
sh, sc = synth.synthetic_cmvn(560)
cmvn = torch.stack([sh, sc])
if which == "sensevoice":
    from funasr_amd.sense_voice import SenseVoiceSmall
    cfg = synth.SENSEVOICE_SMALL
    sd = synth.sensevoice_state_dict(cfg, seed=0)
    B, secs = 128, 10.0
    def make():
        m = SenseVoiceSmall.from_config(cfg); m.load_state_dict(sd, strict=False); m = m.to(dev); m.set_precision("f16x2"); return m
    call = lambda m, f, fl: m.enqueue_features(f, fl, "auto", "woitn")
else:
    from funasr_amd.paraformer import Paraformer
    cfg = synth.PARAFORMER_LARGE
    sd = synth.paraformer_state_dict(cfg, seed=0, cif_bias=synth.BENCH_CIF_BIAS)
    B, secs = 64, 30.0
    def make():
        m = Paraformer.from_config(cfg); m.load_state_dict(sd, strict=False); m = m.to(dev); m.set_precision("f16x2"); return m
    call = lambda m, f, fl: m.enqueue_features(f, fl)
n = int(secs * 16000)
wav = torch.stack([synth.speech_like(n, seed=500 + i) for i in range(B)]).to(dev)
lens = [n] * B
R = 2
models = [make() for _ in range(R)]
fes = [WavFrontend(cmvn=cmvn, lfr_m=7, lfr_n=6, dither=0.0, device=dev) for _ in range(R)]
streams = [torch.cuda.Stream(device=dev) for _ in range(R)]
torch.cuda.synchronize()

def enqueue(r):
    with torch.cuda.stream(streams[r]):
        f, fl = fes[r](wav, lens)
        return call(models[r], f, fl)

def run(k, reps):
    """k batches over `reps` replicas; every replica keeps two batches in flight on its stream (bench.py's loop per replica)"""
    pend = [[] for _ in range(reps)]
    last = None
    for i in range(k):
        r = i % reps
        pend[r].append(enqueue(r))
        if len(pend[r]) > 1:
            with torch.cuda.stream(streams[r]):
                last = models[r].collect(pend[r].pop(0))
    for r in range(reps):
        for p in pend[r]:
            with torch.cuda.stream(streams[r]):
                last = models[r].collect(p)
    return last

out = {"workload": which, "batch": B, "clip_s": secs, "steps": steps}
ref = None
for rep in range(2):
    for reps in (1, 2):
        run(2 * reps, reps); torch.cuda.synchronize()
        t0 = time.perf_counter(); res = run(steps, reps); torch.cuda.synchronize(); dt = time.perf_counter() - t0
        if ref is None:
            ref = res["ids"]
        out.setdefault(f"replicas_{reps}", []).append({"audio_s_per_s": round(B * secs * steps / dt, 1), "ms_per_batch": round(dt / steps * 1e3, 2),
                                                       "ids_equal_first_run": res["ids"] == ref})
print(json.dumps(out))
