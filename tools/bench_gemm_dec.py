#!/usr/bin/env python3
"""gemm_f16x2 on the decoder's shapes (M = 64 clips x ~230 token rows) for both block shapes."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from funasr_amd import ops
dev = torch.device("cuda:0")
for M in (14720, 11008):
    for name, N, K in (("w_1", 2048, 512), ("w_2", 512, 2048), ("q/o", 512, 512), ("vocab", 8404, 512)):
        a = torch.randn(M, K, device=dev); w = torch.randn(N, K, device=dev) * K ** -0.5; b = torch.randn(N, device=dev)
        a2, w2 = ops.split2(a, 8), ops.split2(w, 12)
        row = {}
        for t in (0, 1, 2):
            if N % 256 and t == 2: continue
            if N % 4: continue
            ms = min(ops.gemm_f16x2(a2, w2, b, scale_exp=20, tile=t, time_iters=20)[1] for _ in range(3))
            row[f"tile{t}_us"] = round(ms * 1e3, 1)
        fl = 2.0 * M * N * K
        print(M, name, row, "TF-eq(auto) %.0f" % (fl / row["tile0_us"] / 1e6), flush=True)
