#!/usr/bin/env python3
"""Kernel durations of the small-M GEMM's forms at the streaming step's shapes (run under rocprofv3 --kernel-trace; the shapes are
told apart by their output width N, which is unique per case): plain vs LayerNorm-on-fetch, 15 vs 23 rows, K = 512 vs 2048."""
import math
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from funasr_amd import ops

dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(0)
cases = []      # (tag, M, N, K, ln)
for M in (15, 23):
    for K in (512, 2048):
        for ln in (False, True):
            cases.append((M, 512 + 16 * len(cases), K, ln))
for M, N, K, ln in cases:
    a = torch.randn(M, K, generator=g).to(dev)
    w = (torch.randn(N, K, generator=g) / math.sqrt(K)).to(dev)
    b = torch.randn(N, generator=g).to(dev)
    st = torch.rand(M, K // 16, 2, generator=g).to(dev) + 16.0
    gam, bet = torch.ones(K, device=dev), torch.zeros(K, device=dev)
    flush = torch.empty(64 << 20, device=dev)                       # 256 MB: the weights leave the caches between launches
    for _ in range(30):
        flush.zero_()
        if ln:
            ops.gemm_small_m_ln(a, w, b, stats_in=st, ln=(gam, bet, 1e-5), want_stats=True)
        else:
            ops.gemm_small_m_ln(a, w, b, want_stats=True)
    print(f"CASE N={N} M={M} K={K} ln={ln}")
torch.cuda.synchronize()
