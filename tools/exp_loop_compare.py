import sys, json
sys.path.insert(0, "/root/repo")
import torch
from funasr_amd import ops
dev = torch.device("cuda:0")
g = torch.Generator(device=dev).manual_seed(7)
out = {}
for name, (M, N, K) in {"out_512": (32768, 512, 512), "w1_2048": (32768, 2048, 512), "out_M8192": (8192, 2048, 512)}.items():
    a2 = ops.split2(torch.randn(M, K, device=dev, generator=g), 8)
    w2 = ops.split2(torch.randn(N, K, device=dev, generator=g) * K ** -0.5, 12)
    b = torch.randn(N, device=dev, generator=g)
    row = {}
    for label, tile in (("t2_full", 2), ("t2_loop_only", 2 + 32), ("t2_no_dma", 2 + 48), ("t7_full", 7), ("t7_loop_only", 7 + 32)):
        try:
            row[label] = round(min(ops.gemm_f16x2(a2, w2, b, scale_exp=20, tile=tile, time_iters=30)[1] for _ in range(3)) * 1e3, 1)
        except Exception as e:
            row[label] = repr(e)[:80]
    out[name] = row
print(json.dumps(out))
