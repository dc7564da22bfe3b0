import json, os, sys
sys.path.insert(0, os.getcwd())
import torch
from funasr_amd import ops
dev = torch.device("cuda:0")
for M in (32768, 16384):
    for name, N, K in (("w1", 2048, 512), ("qkv", 1536, 512), ("w2", 512, 2048)):
        g = torch.Generator(device=dev).manual_seed(7)
        a = torch.randn(M, K, device=dev, generator=g); w = torch.randn(N, K, device=dev, generator=g) * K ** -0.5
        b = torch.randn(N, device=dev, generator=g)
        r2 = torch.randn(M, N, device=dev, generator=g)
        a2, w2 = ops.split2(a, 8), ops.split2(w, 12)
        row = {"M": M, "shape": name}
        for label, kw in (("fp32", {}), ("fp32+res", dict(add2=r2)), ("planes", dict(relu=True, out_planes=True, out_scale_exp=9))):
            ref = ops.gemm_f16x2(a2, w2, b, scale_exp=20, tile=2, **kw)
            out = ops.gemm_f16x2(a2, w2, b, scale_exp=20, tile=10, **kw)
            row["bits_" + label] = bool(torch.equal(out, ref))
            row["us2_" + label] = round(min(ops.gemm_f16x2(a2, w2, b, scale_exp=20, tile=2, time_iters=30, **kw)[1] for _ in range(3)) * 1e3, 1)
            row["us10_" + label] = round(min(ops.gemm_f16x2(a2, w2, b, scale_exp=20, tile=10, time_iters=30, **kw)[1] for _ in range(3)) * 1e3, 1)
        if name == "qkv":
            ref = ops.gemm_f16x2_qkv(a2, w2, b, 512, 20, 8.0, 16.0, 32.0, tile=2)
            out = ops.gemm_f16x2_qkv(a2, w2, b, 512, 20, 8.0, 16.0, 32.0, tile=10)
            row["bits_qkvform"] = all(bool(torch.equal(out[k], ref[k])) for k in ("q2", "k2", "v", "vt"))
            row["us2_qkvform"] = round(min(ops.gemm_f16x2_qkv(a2, w2, b, 512, 20, 8.0, 16.0, 32.0, tile=2, time_iters=30)["ms"] for _ in range(3)) * 1e3, 1)
            row["us10_qkvform"] = round(min(ops.gemm_f16x2_qkv(a2, w2, b, 512, 20, 8.0, 16.0, 32.0, tile=10, time_iters=30)["ms"] for _ in range(3)) * 1e3, 1)
        print(json.dumps(row), flush=True)
