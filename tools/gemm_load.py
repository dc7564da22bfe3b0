#!/usr/bin/env python3
"""Background load for tools/repro_frontend.py (its own process): one kind of this library's kernels in a loop for SECONDS.
usage: gemm_load.py SECONDS KIND   KIND: big256 (f16x2 GEMM, 256 x 256 blocks, ~130 KB of LDS per workgroup: nothing else fits on its
CU), tile128 (f16x2 GEMM, 128 x 128 blocks, LDS-DMA staged, two workgroups per CU), fp32tile (fp32 MFMA GEMM, LDS-DMA staged), skinny
(small-M GEMM: plain loads, LDS only for the final reduction), ln (LayerNorm: no LDS), encoder (the whole encoder on a short batch)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from funasr_amd import ops, _lib

dev = torch.device("cuda:0")
secs = float(sys.argv[1]) if len(sys.argv) > 1 else 30.0
kind = sys.argv[2] if len(sys.argv) > 2 else "big256"
g = torch.Generator().manual_seed(0)
if kind in ("big256", "tile128"):
    M = 32768 if kind == "big256" else 1024
    a = ops.split2(torch.randn(M, 512, generator=g).to(dev), 8)
    w = ops.split2((torch.randn(2048, 512, generator=g) * 512 ** -0.5).to(dev), 12)
    b = torch.zeros(2048, device=dev)
    run = (lambda: ops.gemm_f16x2(a, w, b, scale_exp=20, tile=2, relu=True, out_planes=True, out_scale_exp=9)) if kind == "big256" else \
          (lambda: ops.gemm_f16x2(a, w, b, scale_exp=20, tile=3))
elif kind in ("fp32tile", "skinny"):
    M = 1024 if kind == "fp32tile" else 15
    a = torch.randn(M, 512, generator=g).to(dev)
    w = (torch.randn(2048, 512, generator=g) * 512 ** -0.5).to(dev)
    b = torch.zeros(2048, device=dev)
    _lib.load().pf_set_skinny_max_m(0 if kind == "fp32tile" else 1 << 30)
    run = lambda: ops.gemm(a, w, b)
elif kind == "ln":
    x = torch.randn(4096, 512, generator=g).to(dev)
    gam, bet = torch.ones(512, device=dev), torch.zeros(512, device=dev)
    run = lambda: ops.layernorm(x, gam, bet, 1e-5)
elif kind == "encoder":
    from funasr_amd import synth
    from funasr_amd.paraformer import Paraformer
    cfg = synth.PARAFORMER_LARGE
    model = Paraformer.from_config(cfg)
    model.load_state_dict(synth.paraformer_state_dict(cfg, seed=0), strict=False)
    model = model.to(dev)
    feats = (torch.randn(8, 100, 560, generator=g) * 0.8).to(dev)
    lens = torch.full((8,), 100, dtype=torch.int32)
    run = lambda: model.encoder(feats, lens)
else:
    raise SystemExit(f"unknown kind {kind}")
torch.cuda.synchronize()
t0 = time.time()
n = 0
while time.time() - t0 < secs:
    for _ in range(20):
        run()
    torch.cuda.synchronize()
    n += 20
print(f"gemm_load[{kind}]: {n} launches in {time.time() - t0:.1f} s")
