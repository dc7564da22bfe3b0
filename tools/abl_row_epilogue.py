#!/usr/bin/env python
"""Per-phase dissection of `linear_out`'s full-row kernel (gemm_f16x2_row_kernel<6>: K = 512 projection + FSMN memory block + residual +
LayerNorm + planes; round 5's review item 3) at the headline shape, stand-alone, random planes. MEASUREMENT LIBRARY ONLY
(PF_LIB_PATH=funasr_amd/libparaformer_hip_measure.so): the switches arrive through PF_ROW_PREFETCH (read once per process), so this
script runs itself once per setting.  1 / 2 / 3: epilogue operands prefetched during the K loop (both / residual / v rows);
16 no v-row loads, 32 no residual loads, 64 no global stores, 128 no K loop, 256 no epilogue (sums of these are combinations)."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
SETTINGS = [(0, "the kernel"), (16, "no v-row loads"), (32, "no residual loads"), (48, "no v-row and no residual loads"), (64, "no stores"),
            (112, "no epilogue loads, no stores (loop + statistics + arithmetic)"), (128, "no K loop (epilogue alone)"),
            (128 + 64, "no K loop, no stores (epilogue's reads alone)"), (128 + 48, "no K loop, no epilogue loads (epilogue's stores alone)"),
            (128 + 112, "no K loop, no epilogue loads, no stores (the epilogue's on-chip work alone: slab, taps, statistics)"),
            (256, "K loop alone (no epilogue)"),
            (512, "the kernel, next stage's DMA pieces all at the top of the stage"), (512 + 256, "K loop alone, pieces at the top of the stage"),
            (1, "prefetch residual + v rows"), (2, "prefetch residual rows"), (3, "prefetch v rows")]


def one():
    import torch
    from funasr_amd import ops
    dev = torch.device("cuda:0")
    M, K, T = 32768, 512, 512
    g = torch.Generator(device=dev).manual_seed(11)
    a2 = ops.split2(torch.randn(M, K, device=dev, generator=g), 8)
    w2 = ops.split2(torch.randn(512, K, device=dev, generator=g) * K ** -0.5, 12)
    bias = torch.randn(512, device=dev, generator=g)
    v = torch.randn(M, 512, device=dev, generator=g)
    x = torch.randn(M, 512, device=dev, generator=g)
    taps = torch.randn(512, 11, device=dev, generator=g) * 0.3
    lo = (torch.arange(M // 16, device=dev, dtype=torch.int32) * 16 // T) * T
    hi = lo + 500
    ln = (torch.ones(512, device=dev), torch.zeros(512, device=dev), 1e-12)
    best = min(ops.gemm_f16x2_row_fsmn(a2, w2, bias, v, taps, lo, hi, add2=x, scale_exp=20, ln=ln, out_scale_exp=8, a_nt=1,
                                       time_iters=30, block_rows=128)[2] for _ in range(4))
    print(json.dumps({"setting": int(os.environ.get("PF_ROW_PREFETCH", "0")), "us_per_launch": round(best * 1e3, 1)}))


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "one":
        one()
    else:
        for val, what in SETTINGS:
            env = dict(os.environ, PF_ROW_PREFETCH=str(val), PF_LIB_PATH=os.path.join(ROOT, "funasr_amd", "libparaformer_hip_measure.so"))
            out = subprocess.run([sys.executable, os.path.abspath(__file__), "one"], env=env, capture_output=True, text=True, timeout=300)
            line = out.stdout.strip().splitlines()[-1] if out.stdout.strip() else json.dumps({"setting": val, "error": out.stderr[-300:]})
            print(json.dumps(dict(json.loads(line), what=what)), flush=True)
