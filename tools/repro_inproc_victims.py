#!/usr/bin/env python3
"""Which OTHER kernels of this library change their output bits when the 128 x 128 f16x2 GEMM runs on a second HIP stream beside them
(the configuration that disturbed fbank_kernel's packed-fp32 arithmetic)? Each victim is run alone for a reference, then in
a loop next to the aggressor; every output is compared bitwise."""
import json, math, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from funasr_amd import ops, _lib

dev = torch.device("cuda:0")
secs = float(sys.argv[1]) if len(sys.argv) > 1 else 4.0
# aggressor: "gemm" = the 128 x 128 f16x2 GEMM of rounds 3 / 4; "regs" (round 5) = f16 MFMAs on register operands with a barrier every 16
# steps (tools/micro/pk_aggr.hip KIND 9), which disturbs a packed-fp32 fbank_kernel in ~90 % of its calls
AGGR = sys.argv[2] if len(sys.argv) > 2 else "gemm"
g = torch.Generator().manual_seed(0)
side = torch.cuda.Stream(device=dev)
with torch.cuda.stream(side):
    aa = ops.split2(torch.randn(1024, 512, generator=g).to(dev), 8)
    aw = ops.split2((torch.randn(2048, 512, generator=g) * 512 ** -0.5).to(dev), 12)
    ab = torch.zeros(2048, device=dev)
M = 2048
x = torch.randn(M, 512, generator=g).to(dev)
h2 = ops.split2(torch.randn(M, 2048, generator=g).to(dev), 8)
w2 = ops.split2((torch.randn(512, 2048, generator=g) * 2048 ** -0.5).to(dev), 12)
x2 = ops.split2(x, 8)
wq = ops.split2((torch.randn(1536, 512, generator=g) * 512 ** -0.5).to(dev), 12)
b512, b1536 = torch.randn(512, generator=g).to(dev), torch.randn(1536, generator=g).to(dev)
gam, bet = torch.rand(512, generator=g).to(dev) + 0.5, torch.randn(512, generator=g).to(dev)
xs = torch.randn(15, 512, generator=g).to(dev)
ws = (torch.randn(2048, 512, generator=g) / math.sqrt(512)).to(dev)
bs = torch.randn(2048, generator=g).to(dev)
h1 = ops.split2(torch.randn(M, 512, generator=g).to(dev), 8)
w1 = ops.split2((torch.randn(2048, 512, generator=g) * 512 ** -0.5).to(dev), 12)
b2048 = torch.randn(2048, generator=g).to(dev)
victims = {
    # round 5: the matrix kernels whose epilogues keep packed-fp32 VALU instructions, each with >= 1e6 checked outputs per run
    "gemm_f16x2 256 x 256 plane output (w_1 form: scale, bias, ReLU, hi / lo split)": lambda: ops.gemm_f16x2(h1, w1, b2048, scale_exp=20, relu=True, out_planes=True, out_scale_exp=9, tile=2),
    "gemm_f16x2 QKV form (Q / K planes, V^T planes, fp32 V)": lambda: [v for v in ops.gemm_f16x2_qkv(x2, wq, b1536, 512, 20, 8.0, 16.0, 32.0, tile=2).values() if isinstance(v, torch.Tensor)],
    "gemm_f16x2_w4 four-wave tile, fp32 + residual (w_2 form)": lambda: ops.gemm_f16x2(h2, w2, b512, add2=x, scale_exp=20, tile=7),
    "gemm_f16x2_row FSMN form (linear_out)": lambda: ops.gemm_f16x2_row_fsmn(x2, ops.split2((torch.randn(512, 512, generator=torch.Generator().manual_seed(1)) * 512 ** -0.5).to(dev), 12), b512, x,
                                                                              (torch.randn(512, 11, generator=torch.Generator().manual_seed(2)) * 0.3).to(dev),
                                                                              (torch.arange(M // 16, dtype=torch.int32) // 32 * 512).to(dev), (torch.arange(M // 16, dtype=torch.int32) // 32 * 512 + 500).to(dev),
                                                                              add2=x, scale_exp=20, ln=(gam, bet, 1e-12), out_scale_exp=7),
    "gemm_f16x2_row (w_2 shape: residual + LayerNorm epilogue)": lambda: ops.gemm_f16x2_row(h2, w2, b512, add2=x, scale_exp=20, ln=(gam, bet, 1e-12), out_scale_exp=7),
    "gemm_f16x2 128 x 128, fp32 out (QKV-sized)": lambda: ops.gemm_f16x2(x2, wq, b1536, scale_exp=20, tile=3),
    "layernorm": lambda: ops.layernorm(x, gam, bet, 1e-12),
    "small-M GEMM (15 rows)": lambda: ops.gemm_small_m_ln(xs, ws, bs, relu=True, want_stats=True),
}


def flat(o):
    return [t for t in (o if isinstance(o, (tuple, list)) else [o]) if isinstance(t, torch.Tensor)]


if AGGR == "regs":
    import ctypes as C
    pa = C.CDLL(os.path.join(os.path.dirname(os.path.abspath(__file__)), "micro", "pk_aggr.so"))
    pa.pk_aggr_launch.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_void_p]
    srcbuf = (torch.rand(65536 // 2, device=dev) * 1.875 + 0.125).to(torch.float16)
    sink = torch.zeros(16, device=dev)
    def aggress():
        for _ in range(40):
            pa.pk_aggr_launch(C.c_void_p(side.cuda_stream), 9, 16, srcbuf.data_ptr(), 512, 400, sink.data_ptr())
else:
    def aggress():
        for _ in range(40):
            ops.gemm_f16x2(aa, aw, ab, scale_exp=20, tile=3)
out = {"aggressor": AGGR}
for name, fn in victims.items():
    ref = [t.clone() for t in flat(fn())]
    torch.cuda.synchronize()
    t0, n, bad = time.time(), 0, 0
    while time.time() - t0 < secs:
        with torch.cuda.stream(side):
            aggress()
        for _ in range(8):
            got = flat(fn())
            n += 1
            bad += 0 if all(torch.equal(a, b) for a, b in zip(got, ref)) else 1
    torch.cuda.synchronize()
    out[name] = {"calls": n, "calls_with_a_different_output_bit": bad, "outputs_checked": n * sum(t.numel() for t in ref)}
print(json.dumps(out))
