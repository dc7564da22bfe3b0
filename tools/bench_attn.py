#!/usr/bin/env python3
"""Self-attention micro-benchmark at the encoder shape (64 x 500 frames, 4 heads x 128): fp32 MFMA vs bf16x3."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from funasr_amd import ops

dev = torch.device("cuda:0")
B, T, H, dk = 64, 500, 4, 128
Tq = int(sys.argv[1]) if len(sys.argv) > 1 else T          # 120 = the decoder's cross-attention (queries = tokens)
qkv = torch.randn(B, T, 3 * H * dk, device=dev)
q, k, v = qkv[:, :Tq, :512].contiguous(), qkv[:, :, 512:1024], qkv[:, :, 1024:]
lens = torch.full((B,), T, dtype=torch.int32, device=dev)
fl = 4.0 * B * Tq * T * H * dk
ref = ops.attention(q, k, v, lens, H, dk ** -0.5)
torch.cuda.synchronize()
a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
a.record()
for _ in range(20):
    ops.attention(q, k, v, lens, H, dk ** -0.5)
b.record(); torch.cuda.synchronize()
ms = a.elapsed_time(b) / 20
out, ms3 = ops.attention_split3(q, k, v, lens, H, dk ** -0.5, time_iters=20)
print(f"fp32 MFMA {ms*1e3:7.1f} us {fl/ms/1e9:6.1f} TF | bf16x3 {ms3*1e3:7.1f} us {fl/ms3/1e9:6.1f} TF-equiv | max diff {(out-ref).abs().max().item():.2e}")
