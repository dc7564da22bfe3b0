#!/usr/bin/env python3
"""Driver for a rocprofv3 --pmc pass over single launches of the f16x2 GEMM shapes of one encoder block at M = 32768
(tools/pmc_gemm.sh): each variant is launched a few times; the counters are read per kernel name from the CSV."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from funasr_amd import ops

dev = torch.device("cuda:0")
M = 32768
g = torch.Generator().manual_seed(0)
for N, K, kw in ((2048, 512, dict(relu=True, out_planes=True, out_scale_exp=9)), (512, 2048, dict(resid=True)), (1536, 512, {})):
    a = torch.randn(M, K, generator=g).to(dev); w = (torch.randn(N, K, generator=g) * K ** -0.5).to(dev); b = torch.randn(N, generator=g).to(dev)
    a2, w2 = ops.split2(a, 8), ops.split2(w, 12)
    kw = dict(kw)
    if kw.pop("resid", False):
        kw["add2"] = torch.randn(M, N, generator=g).to(dev)
    for tile in (2, 6):
        for _ in range(4):
            ops.gemm_f16x2(a2, w2, b, scale_exp=20, tile=tile, **kw)
    if N == 512:
        gamma = torch.ones(512, device=dev); beta = torch.zeros(512, device=dev)
        for _ in range(4):
            ops.gemm_f16x2_row(a2, w2, b, add2=kw["add2"], scale_exp=20, ln=(gamma, beta, 1e-12), out_scale_exp=7)
torch.cuda.synchronize()
