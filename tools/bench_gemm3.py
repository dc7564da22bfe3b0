#!/usr/bin/env python3
"""bf16x3 split GEMM micro-benchmark on the encoder shapes (M = 64 x 500 rows), TFLOP/s of fp32-equivalent work."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from funasr_amd import ops

from funasr_amd import _lib
dev = torch.device("cuda:0")
M = 32000
if True:
  tot = 0.0
  for name, N, K, planes in (("qkv", 1536, 512, False), ("out", 512, 512, False), ("ffn1", 2048, 512, True), ("ffn2", 512, 2048, False)):
    a = torch.randn(M, K, device=dev); w = torch.randn(N, K, device=dev); b = torch.randn(N, device=dev)
    a3, w3 = ops.split3(a), ops.split3(w)
    fl = 2.0 * M * N * K
    ref = (a.double() @ w.double().T + b.double())
    out = ops.gemm_split3(a3, w3, b)
    err = ((out.double() - ref).abs().max() / ref.abs().max()).item()
    ms = min(ops.gemm_split3(a3, w3, b, relu=planes, out_planes=planes, time_iters=20)[1] for _ in range(3))
    tot += ms
    print(f"{name:5s} N={N} K={K}: {ms*1e3:7.1f} us {fl/ms/1e9:6.1f} TF-equiv  rel err {err:.1e}", flush=True)
  print(f"layer total {tot*1e3:.1f} us  (x50 = {tot*50:.1f} ms)")
