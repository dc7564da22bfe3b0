#!/usr/bin/env python3
"""Experiment: one full batch on one stream vs two half batches on two streams (two host threads, two model handles with
the same weights) -- does overlapping MFMA-bound and HBM-bound kernels of independent sub-batches raise throughput?"""
import os, sys, threading, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from funasr_amd import synth
from funasr_amd.paraformer import Paraformer
from funasr_amd.wav_frontend import WavFrontend

dev = torch.device("cuda:0")
cfg = synth.PARAFORMER_LARGE
sd = synth.paraformer_state_dict(cfg, seed=0, cif_bias=synth.BENCH_CIF_BIAS)
shift, scale = synth.synthetic_cmvn(560)
B, n = 64, 480000
base = [synth.speech_like(n, seed=i) for i in range(8)]
wav = torch.stack([base[i % 8].roll(137 * (i // 8)) for i in range(B)]).to(dev)
mode = sys.argv[1] if len(sys.argv) > 1 else "f16x2"

def make():
    m = Paraformer.from_config(cfg); m.load_state_dict(sd, strict=False); m = m.to(dev); m.set_precision(mode)
    fe = WavFrontend(cmvn=torch.stack([shift, scale]), lfr_m=7, lfr_n=6, dither=0.0, device=dev)
    return m, fe

def run(m, fe, w, steps):
    lens = [n] * w.shape[0]
    pend = m.enqueue_features(*fe(w, lens))
    for _ in range(steps - 1):
        nxt = m.enqueue_features(*fe(w, lens))
        r = m.collect(pend); pend = nxt
    return m.collect(pend)

K = 6
m0, f0 = make()
run(m0, f0, wav, 2); torch.cuda.synchronize()
t0 = time.perf_counter(); r_full = run(m0, f0, wav, K); torch.cuda.synchronize(); t_full = (time.perf_counter() - t0) / K
print(f"one stream, batch 64: {t_full*1e3:.1f} ms/step  {B*30/t_full:.0f} audio-s/s", flush=True)

for nsplit in (2, 4):
    models = [make() for _ in range(nsplit)]
    streams = [torch.cuda.Stream() for _ in range(nsplit)]
    hb = B // nsplit
    res = [None] * nsplit
    def worker(i, steps):
        with torch.cuda.stream(streams[i]):
            res[i] = run(models[i][0], models[i][1], wav[i * hb:(i + 1) * hb], steps)
            streams[i].synchronize()
    for steps in (2, K):
        th = [threading.Thread(target=worker, args=(i, steps)) for i in range(nsplit)]
        torch.cuda.synchronize(); t0 = time.perf_counter()
        [t.start() for t in th]; [t.join() for t in th]
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / steps
    ids = sum((r["raw_ids"] for r in res), [])
    print(f"{nsplit} streams x batch {hb}: {dt*1e3:.1f} ms/step  {B*30/dt:.0f} audio-s/s  ids equal full-batch: {ids == r_full['raw_ids']}", flush=True)
    del models
