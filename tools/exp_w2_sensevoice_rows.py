#!/usr/bin/env python
"""w_2 (N = 512, K = 2048, + residual, fp32 out) at SenseVoice's row count M = 22 528 (176 blocks of 256 x 256 on 256 CUs: one under-filled
round) on every block shape of the product library, stand-alone, random planes: is a finer shape faster there?"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from funasr_amd import ops  # noqa: E402

dev = torch.device("cuda:0")
g = torch.Generator(device=dev).manual_seed(7)
for M in (22528, 32768):
    N, K = 512, 2048
    a2 = ops.split2(torch.randn(M, K, device=dev, generator=g), 8)
    w2 = ops.split2(torch.randn(N, K, device=dev, generator=g) * K ** -0.5, 12)
    b = torch.randn(N, device=dev, generator=g)
    r = torch.randn(M, N, device=dev, generator=g)
    row = {"M": M}
    ref = None
    for label, tile in (("auto", 0), ("256x256 eight waves", 2), ("128x128 two workgroups per CU", 3), ("256x256 four waves", 7), ("persistent 256x128", 10)):
        try:
            out, ms = ops.gemm_f16x2(a2, w2, b, scale_exp=20, tile=tile, add2=r, time_iters=30)
            ms = min([ms] + [ops.gemm_f16x2(a2, w2, b, scale_exp=20, tile=tile, add2=r, time_iters=30)[1] for _ in range(2)])
            ref = out if ref is None else ref
            row[label] = {"us": round(ms * 1e3, 1), "bitwise_equal_to_auto": bool(torch.equal(out, ref))}
        except Exception as e:  # noqa: BLE001
            row[label] = repr(e)[:100]
    print(json.dumps(row), flush=True)
