#!/usr/bin/env python3
"""Two-plane fp16 split GEMM (gemm_f16x2.hip): subnormal-operand probe, accuracy against float64 next to the bf16x3 kernel,
and TFLOP/s of fp32-equivalent work on the encoder shapes (M = 64 x 500 rows) for both block shapes."""
import json, math, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from funasr_amd import ops

dev = torch.device("cuda:0")
out = {}

# ---- does v_mfma_f32_32x32x16_f16 keep fp16 subnormal inputs? (the single-accumulator split relies on lo planes
#      that may be subnormal). a = 2^-20 (subnormal in fp16, exactly representable), w = 2^10: a.w summed over K = 32
a = torch.full((256, 32), 2.0 ** -20, device=dev); w = torch.full((128, 32), 2.0 ** 10, device=dev)
r = ops.gemm_f16x2(ops.split2(a), ops.split2(w))
out["mfma_f16_subnormal_probe"] = {"expected": 32 * 2.0 ** -10, "got": float(r[0, 0]), "kept": bool(r[0, 0] == 32 * 2.0 ** -10)}
print(out["mfma_f16_subnormal_probe"], flush=True)

def exp_for(x, top=14):
    return top - int(math.floor(math.log2(float(x.abs().max()))))

M = 32000
tot = {0: 0.0, 1: 0.0, 2: 0.0}
rows = []
for name, N, K, planes in (("qkv", 1536, 512, False), ("out", 512, 512, False), ("ffn1", 2048, 512, True), ("ffn2", 512, 2048, False)):
    torch.manual_seed(0)
    a = torch.randn(M, K, device=dev) * torch.exp(torch.randn(M, K, device=dev))    # heavy-tailed activations
    if name == "ffn2": a = a.clamp_min(0)
    w = torch.randn(N, K, device=dev) * K ** -0.5; b = torch.randn(N, device=dev)
    ea, ew = exp_for(a), exp_for(w)
    a2, w2 = ops.split2(a, ea), ops.split2(w, ew)
    a3, w3 = ops.split3(a), ops.split3(w)
    ref = (a[:4096].double() @ w.double().T + b.double())
    sc = ref.abs().max()
    e2 = {t: ((ops.gemm_f16x2(a2[:, :4096].contiguous(), w2, b, scale_exp=ea + ew, tile=t).double() - ref).abs().max() / sc).item() for t in (1, 2)}
    e3 = ((ops.gemm_split3(a3[:, :4096].contiguous(), w3, b).double() - ref).abs().max() / sc).item()
    e32 = (((a[:4096] @ w.T + b).double() - ref).abs().max() / sc).item()
    same = bool(torch.equal(ops.gemm_f16x2(a2[:, :4096].contiguous(), w2, b, scale_exp=ea + ew, tile=1),
                            ops.gemm_f16x2(a2[:, :4096].contiguous(), w2, b, scale_exp=ea + ew, tile=2)))
    fl = 2.0 * M * N * K
    row = {"name": name, "N": N, "K": K, "err_f16x2": e2, "err_bf16x3": e3, "err_torch_f32": e32, "tiles_bitwise_equal": same}
    for t in (1, 2):
        ms = min(ops.gemm_f16x2(a2, w2, b, relu=planes, out_planes=planes, scale_exp=ea + ew, out_scale_exp=4, tile=t, time_iters=20)[1] for _ in range(3))
        row[f"us_tile{t}"] = ms * 1e3; row[f"tf_tile{t}"] = fl / ms / 1e9
        tot[t] += ms
    ms = min(ops.gemm_split3(a3, w3, b, relu=planes, out_planes=planes, time_iters=20)[1] for _ in range(3))
    row["us_bf16x3"] = ms * 1e3; row["tf_bf16x3"] = fl / ms / 1e9; tot[0] += ms
    rows.append(row)
    print(json.dumps(row), flush=True)
out["shapes"] = rows
out["layer_us"] = {"bf16x3": tot[0] * 1e3, "f16x2_256x128": tot[1] * 1e3, "f16x2_256x256": tot[2] * 1e3}
print(json.dumps(out["layer_us"]))
os.makedirs("gpurun_out", exist_ok=True)
json.dump(out, open("gpurun_out/bench_gemm2.json", "w"), indent=1)
