#!/usr/bin/env python3
"""Ablation of gemm_f16x2_kernel (wide tile) on the QKV / ffn shapes: full kernel vs no-global-stores vs no-epilogue vs no-DMA."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from funasr_amd import ops
dev = torch.device("cuda:0")
M = 32768
if len(sys.argv) > 1 and sys.argv[1] == "resid":
    # the out-projection / w_2 forms with residual operands (modes 3 and 2): which tile shape hides the epilogue's residual reads best
    for name, N, K, nres in (("out+2res", 512, 512, 2), ("out+1res", 512, 512, 1), ("out", 512, 512, 0), ("ffn2+1res", 512, 2048, 1)):
        a = torch.randn(M, K, device=dev); w = torch.randn(N, K, device=dev) * K ** -0.5; b = torch.randn(N, device=dev)
        r1 = torch.randn(M, N, device=dev); r2 = torch.randn(M, N, device=dev)
        a2, w2 = ops.split2(a, 8), ops.split2(w, 12)
        kw = dict(add1=r1 if nres == 2 else None, add2=r2 if nres >= 1 else None)
        row = {}
        ref = ops.gemm_f16x2(a2, w2, b, scale_exp=20, tile=2, **kw)
        for label, tile in (("wide", 2), ("narrow", 1), ("small2wg", 3), ("wide_sched0", 2 + 256)):
            out = ops.gemm_f16x2(a2, w2, b, scale_exp=20, tile=tile, **kw)
            ms = min(ops.gemm_f16x2(a2, w2, b, scale_exp=20, tile=tile, time_iters=20, **kw)[1] for _ in range(3))
            row[label] = (round(ms * 1e3, 1), bool(torch.equal(out, ref)))
        print(name, row, flush=True)
    sys.exit(0)
for name, N, K in (("qkv", 1536, 512), ("ffn1", 2048, 512), ("ffn2", 512, 2048), ("out", 512, 512)):
    a = torch.randn(M, K, device=dev); w = torch.randn(N, K, device=dev) * K ** -0.5; b = torch.randn(N, device=dev)
    a2, w2 = ops.split2(a, 8), ops.split2(w, 12)
    row = {}
    same4 = bool(torch.equal(ops.gemm_f16x2(a2, w2, b, scale_exp=20, tile=4), ops.gemm_f16x2(a2, w2, b, scale_exp=20, tile=2)))
    same3 = bool(torch.equal(ops.gemm_f16x2(a2[:, :4096].contiguous(), w2, b, scale_exp=20, tile=3), ops.gemm_f16x2(a2[:, :4096].contiguous(), w2, b, scale_exp=20, tile=1)))
    for label, tile in (("full", 2), ("nostore", 2 + 16), ("noepi", 2 + 32), ("nodma", 2 + 48), ("l2store", 2 + 64), ("persist", 4), ("narrow", 1), ("small2wg", 3), ("sched1", 2 + 256), ("sched2", 2 + 512)):
        ms = min(ops.gemm_f16x2(a2, w2, b, scale_exp=20, tile=tile, time_iters=20)[1] for _ in range(3))
        row[label] = round(ms * 1e3, 1)
    fl = 2.0 * M * N * K
    print(name, "tile3==tile1:", same3, "persist==tile2:", same4, row, "TF-eq full %.0f noepi %.0f" % (fl / row["full"] / 1e6, fl / row["noepi"] / 1e6), flush=True)
