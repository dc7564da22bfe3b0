#!/usr/bin/env python3
"""GEMM micro-benchmark on the hot-path shapes (M = 64 x 500 rows): fp32 MFMA vs bf16-operand MFMA, TFLOP/s."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from funasr_amd import ops

dev = torch.device("cuda:0")
M = 32000
for name, N, K in (("qkv", 1536, 512), ("out", 512, 512), ("ffn1", 2048, 512), ("ffn2", 512, 2048), ("dec_kv", 1024, 512),
                   ("square4k", 4096, 4096)):
    m = 4096 if name == "square4k" else M
    a = torch.randn(m, K, device=dev)
    w = torch.randn(N, K, device=dev)
    b = torch.randn(N, device=dev)
    out = torch.empty(m, N, device=dev)
    ab, wb = ops.cast_bf16(a), ops.cast_bf16(w)
    ob = torch.empty(m, N, device=dev, dtype=torch.bfloat16)
    fl = 2.0 * m * N * K
    ms = min(ops.gemm_time_ms(a, w, b, out, 20) for _ in range(3))
    ms16 = min(ops.gemm_bf16_time_ms(ab, wb, b, ob, 20) for _ in range(3))
    a3, w3 = ops.split3(a), ops.split3(w)
    ms3 = min(ops.gemm_split3(a3, w3, b, time_iters=20)[1] for _ in range(3))
    ms3p = min(ops.gemm_split3(a3, w3, b, relu=True, out_planes=True, time_iters=20)[1] for _ in range(3))
    print(f"{name:9s} M={m} N={N} K={K}: f32 {ms*1e3:7.1f} us {fl/ms/1e9:6.1f} TF | bf16 {ms16*1e3:7.1f} us {fl/ms16/1e9:7.1f} TF"
          f" | bf16x3 {ms3*1e3:7.1f} us {fl/ms3/1e9:6.1f} TF-equiv ({6*fl/ms3/1e9:6.0f} raw), plane-out {fl/ms3p/1e9:6.1f}", flush=True)
