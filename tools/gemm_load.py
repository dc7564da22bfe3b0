#!/usr/bin/env python3
"""Background load for tools/repro_frontend.py: this library's large-LDS f16x2 GEMM (LDS-DMA staged, 256 x 256 blocks) in a loop
for SECONDS (argv[1]) seconds, in its own process."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from funasr_amd import ops

dev = torch.device("cuda:0")
secs = float(sys.argv[1]) if len(sys.argv) > 1 else 30.0
g = torch.Generator().manual_seed(0)
a = ops.split2(torch.randn(32768, 512, generator=g).to(dev), 8)
w = ops.split2((torch.randn(2048, 512, generator=g) * 512 ** -0.5).to(dev), 12)
b = torch.zeros(2048, device=dev)
torch.cuda.synchronize()
t0 = time.time()
n = 0
while time.time() - t0 < secs:
    for _ in range(50):
        ops.gemm_f16x2(a, w, b, scale_exp=20, tile=2, relu=True, out_planes=True, out_scale_exp=9)
    torch.cuda.synchronize()
    n += 50
print(f"gemm_load: {n} launches in {time.time() - t0:.1f} s")
