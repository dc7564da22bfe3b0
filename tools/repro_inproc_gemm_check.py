#!/usr/bin/env python3
"""Companion of tools/repro_inproc.py: while the frontend runs on the main stream, is the OUTPUT OF THE 128 x 128 f16x2 GEMM on the side
stream still the bits it returns alone? (Tells a disturbance specific to fbank_kernel from one that hits everything on a shared CU.)"""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from funasr_amd import ops, synth
from funasr_amd.wav_frontend import WavFrontend

dev = torch.device("cuda:0")
secs = float(sys.argv[1]) if len(sys.argv) > 1 else 8.0
g = torch.Generator().manual_seed(0)
sh, sc = synth.synthetic_cmvn(560)
fe = WavFrontend(cmvn=torch.stack([sh, sc]), lfr_m=7, lfr_n=6, dither=0.0, device=dev, verify=True)
wav = synth.speech_like(235000, seed=7).to(dev)[None]
a = ops.split2(torch.randn(1024, 512, generator=g).to(dev), 8)
w = ops.split2((torch.randn(2048, 512, generator=g) * 512 ** -0.5).to(dev), 12)
b = torch.zeros(2048, device=dev)
ref = ops.gemm_f16x2(a, w, b, scale_exp=20, tile=3).clone()
side = torch.cuda.Stream(device=dev)
outs = [torch.empty_like(ref) for _ in range(40)]
torch.cuda.synchronize()
t0 = time.time()
n = bad = 0
while time.time() - t0 < secs:
    with torch.cuda.stream(side):
        for o in outs:
            ops.gemm_f16x2(a, w, b, scale_exp=20, tile=3, out=o)
    for _ in range(4):
        fe(wav, [235000])
    torch.cuda.synchronize()
    for o in outs:
        n += 1
        bad += 0 if torch.equal(o, ref) else 1
print(json.dumps({"gemm_outputs_checked": n, "gemm_outputs_different": bad, "frontend_cross_check_disagreements": fe.faults()}))
