#!/usr/bin/env python3
"""attention_f16x2.hip standalone: accuracy against float64 attention and time per launch at the encoder's shape
(B = 64, T = 512 padded from 500, 4 heads of 128), default vs pipelined schedule vs the four-wave two-workgroups-per-CU variant; plus the cross-attention shape."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from funasr_amd import ops
dev = torch.device("cuda:0")
out = {}
for name, B, Tq, Tp, klen in (("self", 64, 512, 512, 500), ("cross", 64, 230, 512, 500)):
    g = torch.Generator().manual_seed(0)
    q = torch.randn(B, Tq, 512, generator=g).to(dev); k = torch.randn(B, Tp, 512, generator=g).to(dev); v = torch.randn(B, Tp, 512, generator=g).to(dev)
    klens = torch.full((B,), klen, dtype=torch.int32, device=dev)
    klens[1] = 37; klens[2] = 1
    scale = 128 ** -0.5
    row = {}
    qq, kk, vv = (t[:4].double().view(4, -1, 4, 128).transpose(1, 2) for t in (q, k, v))
    sc = (qq * scale) @ kk.transpose(-1, -2)
    mask = torch.arange(Tp, device=dev)[None, None, None, :] >= klens[:4, None, None, None]
    ref = (torch.softmax(sc.masked_fill(mask, -float("inf")), -1).masked_fill(mask, 0) @ vv).transpose(1, 2).reshape(4, Tq, 512)
    for variant in (0, 1, 3, 4):
        o = ops.attention_f16x2(q, k, v, klens, 4, scale, variant=variant)
        row[f"err_v{variant}"] = float((o[:4].double() - ref).abs().max())
        row[f"us_v{variant}"] = min(ops.attention_f16x2(q, k, v, klens, 4, scale, variant=variant, time_iters=20)[1] for _ in range(3)) * 1e3
    row["variants_bitwise_equal"] = bool(torch.equal(ops.attention_f16x2(q, k, v, klens, 4, scale, variant=0), ops.attention_f16x2(q, k, v, klens, 4, scale, variant=1)))
    row["v2_bitwise_equal_v0"] = bool(torch.equal(ops.attention_f16x2(q, k, v, klens, 4, scale, variant=0), ops.attention_f16x2(q, k, v, klens, 4, scale, variant=2)))
    row["v4_bitwise_equal_v3"] = bool(torch.equal(ops.attention_f16x2(q, k, v, klens, 4, scale, variant=3), ops.attention_f16x2(q, k, v, klens, 4, scale, variant=4)))
    fl = 4.0 * sum(int(x) for x in klens.tolist()) * Tq * 512
    row["tflops_eq_v0"] = fl / row["us_v0"] / 1e6; row["tflops_eq_v1"] = fl / row["us_v1"] / 1e6; row["tflops_eq_v3"] = fl / row["us_v3"] / 1e6; row["tflops_eq_v4"] = fl / row["us_v4"] / 1e6
    out[name] = row
    print(name, json.dumps(row), flush=True)
os.makedirs("gpurun_out", exist_ok=True)
json.dump(out, open("gpurun_out/bench_attn2.json", "w"), indent=1)
