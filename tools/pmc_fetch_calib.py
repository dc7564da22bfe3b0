#!/usr/bin/env python3
"""Calibration launches for rocprofv3's FETCH_SIZE on THIS library's access patterns (tools/pmc_fetch_calib.sh): the guide's
"x 2 on gfx950" holds for 128-B requests; a kernel whose DMA pieces fetch 64-B row segments may be counted exactly. Each launch
below has a read volume known by construction (one column block: no operand can be re-read; a plain copy), so raw FETCH_SIZE /
known bytes is the factor for that access pattern. Prints the known byte counts; the shell script joins them with the counters."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from funasr_amd import ops

dev = torch.device("cuda:0")
M = 32768
g = torch.Generator().manual_seed(0)
known = {}


def planes(rows, cols, e):
    return ops.split2((torch.randn(rows, cols, generator=g) * cols ** -0.5).to(dev), e)


a512, a2048 = ops.split2(torch.randn(M, 512, generator=g).to(dev), 8), ops.split2(torch.randn(M, 2048, generator=g).to(dev), 8)
# 1. a plain 16-B-per-lane streaming copy (torch): the guide's calibration case
x = torch.empty(M * 2048, device=dev, dtype=torch.float32).normal_()
for _ in range(3):
    y = x.clone()
known["copy"] = {"read_bytes": x.numel() * 4, "note": "torch clone of 268 MB"}
# 2. tile kernel, ONE column block of 256 (N = 256): every A panel is fetched by exactly one workgroup
w = planes(256, 512, 12)
b = torch.zeros(256, device=dev)
for _ in range(3):
    ops.gemm_f16x2(a512, w, b, scale_exp=20, tile=2)
known["tile_N256_K512"] = {"read_bytes": M * 512 * 4 + 256 * 512 * 4, "note": "A planes once + W"}
# 3. the w_1 shape (N = 2048, plane output) and the QKV-sized fp32 shape (N = 1536)
w = planes(2048, 512, 12)
b = torch.zeros(2048, device=dev)
for _ in range(3):
    ops.gemm_f16x2(a512, w, b, scale_exp=20, tile=2, relu=True, out_planes=True, out_scale_exp=9)
known["tile_N2048_K512_w1"] = {"read_bytes": M * 512 * 4 + 2048 * 512 * 4, "note": "algorithmic: A planes once + W once"}
w = planes(1536, 512, 12)
b = torch.zeros(1536, device=dev)
for _ in range(3):
    ops.gemm_f16x2(a512, w, b, scale_exp=20, tile=2)
known["tile_N1536_K512"] = {"read_bytes": M * 512 * 4 + 1536 * 512 * 4, "note": "algorithmic"}
# 4. the full-row kernel (w_2 shape): a workgroup owns whole rows, A is read once by construction
w = planes(512, 2048, 12)
b = torch.zeros(512, device=dev)
r = torch.randn(M, 512, generator=g).to(dev)
gamma, beta = torch.ones(512, device=dev), torch.zeros(512, device=dev)
for _ in range(3):
    ops.gemm_f16x2_row(a2048, w, b, add2=r, scale_exp=20, ln=(gamma, beta, 1e-12), out_scale_exp=7)
known["row_N512_K2048_w2"] = {"read_bytes": M * 2048 * 4 + M * 512 * 4, "note": "hidden planes + residual rows once (+ W per workgroup from L2)"}
torch.cuda.synchronize()
print("KNOWN " + json.dumps(known))
