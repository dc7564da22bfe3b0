#!/usr/bin/env python3
"""Round 5 experiment: how much of a tile GEMM's time is the chip-wide store burst of aligned epilogues? Half of the first round's
workgroups start late by `relu` x s_sleep(127) (measurement build ABL 5 of gemm_f16x2_kernel, tile 2 + 5 * 16), which de-phases the
two halves of the chip for the remaining rounds. total(delay) - total(0) = delay - gain."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from funasr_amd import ops
dev = torch.device("cuda:0")
M = 32768
for name, N, K in (("w1", 2048, 512), ("qkv", 1536, 512), ("w2", 512, 2048)):
    g = torch.Generator(device=dev).manual_seed(7)
    a = torch.randn(M, K, device=dev, generator=g); w = torch.randn(N, K, device=dev, generator=g) * K ** -0.5
    b = torch.randn(N, device=dev, generator=g)
    a2, w2 = ops.split2(a, 8), ops.split2(w, 12)
    row = {"shape": name}
    base = min(ops.gemm_f16x2(a2, w2, b, scale_exp=20, tile=2, time_iters=30)[1] for _ in range(3)) * 1e3
    row["tile2_us"] = round(base, 1)
    row["tile8_sched0_us"] = round(min(ops.gemm_f16x2(a2, w2, b, scale_exp=20, tile=8, time_iters=30)[1] for _ in range(3)) * 1e3, 1)
    row["tile2_again_us"] = round(min(ops.gemm_f16x2(a2, w2, b, scale_exp=20, tile=2, time_iters=30)[1] for _ in range(3)) * 1e3, 1)
    row["bits_equal"] = bool(torch.equal(ops.gemm_f16x2(a2, w2, b, scale_exp=20, tile=8), ops.gemm_f16x2(a2, w2, b, scale_exp=20, tile=2)))
    if name == "w1":
        for t in (2, 8, 7):
            row[f"planes_tile{t}_us"] = round(min(ops.gemm_f16x2(a2, w2, b, scale_exp=20, tile=t, relu=True, out_planes=True, out_scale_exp=9, time_iters=30)[1] for _ in range(3)) * 1e3, 1)
    if name == "qkv":
        for t in (2, 8, 7):
            row[f"qkvform_tile{t}_us"] = round(min(ops.gemm_f16x2_qkv(a2, w2, b, 512, 20, 8.0, 16.0, 32.0, tile=t, time_iters=30)["ms"] for _ in range(3)) * 1e3, 1)
    for d in (0, 4, 8):
        # relu doubles as the delay count here (the outputs are not looked at); relu = 0 is the undelayed measurement build
        t = min(ops.gemm_f16x2(a2, w2, b, scale_exp=20, tile=2 + 5 * 16, relu=d, time_iters=30)[1] for _ in range(3)) * 1e3
        row[f"delay_{d}"] = round(t, 1)
    print(json.dumps(row), flush=True)
