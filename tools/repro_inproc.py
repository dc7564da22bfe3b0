#!/usr/bin/env python3
"""The frontend cross-check's fault counter with this library's 128 x 128 f16x2 GEMM running on a SECOND HIP STREAM OF THE SAME
PROCESS (tools/repro_frontend.py + tools/gemm_load.py do it with a second process). usage: repro_inproc.py SECONDS KIND"""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from funasr_amd import ops, synth, _lib
from funasr_amd.wav_frontend import WavFrontend

dev = torch.device("cuda:0")
secs = float(sys.argv[1]) if len(sys.argv) > 1 else 20.0
kind = sys.argv[2] if len(sys.argv) > 2 else "tile128"
g = torch.Generator().manual_seed(0)
sh, sc = synth.synthetic_cmvn(560)
fe = WavFrontend(cmvn=torch.stack([sh, sc]), lfr_m=7, lfr_n=6, dither=0.0, device=dev, verify=True)
wav = synth.speech_like(235000, seed=7).to(dev)[None]
ref = fe(wav, [235000])[0].clone()
side = torch.cuda.Stream(device=dev)
with torch.cuda.stream(side):
    M = 1024
    if kind.startswith("tile128"):
        a = ops.split2(torch.randn(M, 512, generator=g).to(dev), 8)
        w = ops.split2((torch.randn(2048, 512, generator=g) * 512 ** -0.5).to(dev), 12)
        b = torch.zeros(2048, device=dev)
        tile = {"tile128": 3, "tile128_no_epilogue": 0x82, "tile128_no_dma": 0x83}[kind]
        run = lambda: ops.gemm_f16x2(a, w, b, scale_exp=20, tile=tile)
    else:
        a = torch.randn(M, 512, generator=g).to(dev)
        w = (torch.randn(2048, 512, generator=g) * 512 ** -0.5).to(dev)
        b = torch.zeros(2048, device=dev)
        run = lambda: ops.gemm(a, w, b)
torch.cuda.synchronize()
t0 = time.time()
it = bad = 0
while time.time() - t0 < secs:
    with torch.cuda.stream(side):
        for _ in range(40):
            run()
    for _ in range(4):
        f = fe(wav, [235000])[0]
        it += 1
        if not torch.equal(f, ref):
            bad += 1
torch.cuda.synchronize()
print(json.dumps({"same_process_second_stream": kind, "frontend_calls": it, "mismatching_outputs": bad, "cross_check_disagreements": fe.faults(),
                  "log": fe.fault_log()[:4]}))
