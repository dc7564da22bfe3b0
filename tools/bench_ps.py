#!/usr/bin/env python3
"""Round 6 measurement of the persistent, wave-specialised f16x2 GEMM (csrc/gemm_f16x2_ps.hip, Gemm2Args.tile 10) against the
shapes the engine runs today (tile 2 = eight-wave 256 x 256, tile 7 = four-wave 256 x 256). Measurement infrastructure.

    python tools/bench_ps.py parity    bitwise tile 10 == tile 2 (fp32 / +residual / planes; ragged M; 200 repeats of one launch)
    python tools/bench_ps.py time      us per launch at M = 32768, random and zero planes: w_1 (planes), QKV-sized planes, w_2 (fp32 + residual)
Every section prints JSON lines (copied to profiles/r06*_*.jsonl by the caller)."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from funasr_amd import ops  # noqa: E402

dev = torch.device("cuda:0")
M = 32768
SHAPES = {"qkv": (1536, 512), "w1": (2048, 512), "w2": (512, 2048), "out": (512, 512)}
PS = 10


def operands(N, K, m=M, data="random"):
    g = torch.Generator(device=dev).manual_seed(7)
    if data == "zeros":
        a = torch.zeros(m, K, device=dev)
        w = torch.zeros(N, K, device=dev)
    else:
        a = torch.randn(m, K, device=dev, generator=g)
        w = torch.randn(N, K, device=dev, generator=g) * K ** -0.5
    b = torch.randn(N, device=dev, generator=g)
    return ops.split2(a, 8), ops.split2(w, 12), b


def parity():
    ok_all = True
    for m in (32768, 32768 - 96, 22528, 304, 16, 4096 + 16):
        for name, (N, K) in SHAPES.items():
            a2, w2, b = operands(N, K, m)
            r2 = torch.randn(m, N, device=dev)
            row = {"M": m, "shape": name}
            for label, kw in (("fp32", {}), ("fp32+relu", dict(relu=True)), ("fp32+res1", dict(add1=r2)), ("fp32+res2", dict(add2=r2)),
                              ("planes", dict(relu=True, out_planes=True, out_scale_exp=9)), ("planes_norelu", dict(out_planes=True, out_scale_exp=5))):
                ref = ops.gemm_f16x2(a2, w2, b, scale_exp=20, tile=2, **kw)
                out = ops.gemm_f16x2(a2, w2, b, scale_exp=20, tile=PS, **kw)
                row[label] = bool(torch.equal(out, ref))
                if not row[label]:
                    d = (out.float() - ref.float()).abs()
                    row[label + "_maxdiff"] = float(d.max())
                    row[label + "_nbad"] = int((d > 0).sum())
                ok_all &= row[label]
            # in place: C == R2 (the engine's w_2: x = x + ...)
            x = r2.clone()
            ref = ops.gemm_f16x2(a2, w2, b, scale_exp=20, tile=2, add2=r2)
            ops.gemm_f16x2(a2, w2, b, scale_exp=20, tile=PS, add2=x, out=x)
            row["fp32+res2_inplace"] = bool(torch.equal(x, ref))
            ok_all &= row["fp32+res2_inplace"]
            if name == "qkv" and m % 16 == 0:
                ref = ops.gemm_f16x2_qkv(a2, w2, b, 512, 20, 2.0 ** 3, 2.0 ** 4, 2.0 ** 5, tile=2)
                out = ops.gemm_f16x2_qkv(a2, w2, b, 512, 20, 2.0 ** 3, 2.0 ** 4, 2.0 ** 5, tile=PS)
                for k in ("q2", "k2", "v", "vt"):
                    row["qkv_" + k] = bool(torch.equal(out[k], ref[k]))
                    ok_all &= row["qkv_" + k]
            print(json.dumps(row), flush=True)
    a2, w2, b = operands(2048, 512)
    ref = ops.gemm_f16x2(a2, w2, b, scale_exp=20, tile=2, relu=True, out_planes=True, out_scale_exp=9)
    bad = sum(0 if torch.equal(ops.gemm_f16x2(a2, w2, b, scale_exp=20, tile=PS, relu=True, out_planes=True, out_scale_exp=9), ref) else 1
              for _ in range(200))
    print(json.dumps({"repeat_w1_planes_200": bad == 0, "mismatching_runs": bad}), flush=True)
    ok_all &= bad == 0
    print(json.dumps({"parity_all": ok_all}), flush=True)
    return ok_all


def best(f, n=3):
    return min(f() for _ in range(n)) * 1e3


def timing():
    for data in ("random", "zeros"):
        for name, (N, K) in SHAPES.items():
            a2, w2, b = operands(N, K, data=data)
            r2 = torch.randn(M, N, device=dev)
            row = {"shape": name, "data": data, "M": M}
            for label, tile in (("ps_nodma", PS + 16), ("ps_noepi", PS + 32), ("ps_loop", PS + 48), ("ps_nostore", PS + 64), ("ps1", 11), ("ps1_nostore", 11 + 64)):
                row[label + "_planes"] = round(best(lambda: ops.gemm_f16x2(a2, w2, b, scale_exp=20, tile=tile, relu=True, out_planes=True,
                                                                         out_scale_exp=9, time_iters=20)[1]), 1)
            for label, tile in (("t2", 2), ("t7", 7), ("ps", PS)):
                row[label + "_fp32"] = round(best(lambda: ops.gemm_f16x2(a2, w2, b, scale_exp=20, tile=tile, time_iters=20)[1]), 1)
                row[label + "_fp32_res"] = round(best(lambda: ops.gemm_f16x2(a2, w2, b, scale_exp=20, tile=tile, add2=r2, out=r2, time_iters=20)[1]), 1)
                row[label + "_planes"] = round(best(lambda: ops.gemm_f16x2(a2, w2, b, scale_exp=20, tile=tile, relu=True, out_planes=True,
                                                                         out_scale_exp=9, time_iters=20)[1]), 1)
            if name == "qkv":
                for label, tile in (("t2", 2), ("t7", 7), ("ps", PS), ("ps1", 11), ("ps_nostore", PS + 64), ("ps_noepi", PS + 32)):
                    row[label + "_qkvform"] = round(best(lambda: ops.gemm_f16x2_qkv(a2, w2, b, 512, 20, 8.0, 16.0, 32.0, tile=tile, time_iters=20)["ms"]), 1)
            fl = 2.0 * M * N * K * 3
            row["exec_TFLOPs_ps_planes"] = round(fl / row["ps_planes"] / 1e6)
            row["exec_TFLOPs_t2_planes"] = round(fl / row["t2_planes"] / 1e6)
            print(json.dumps(row), flush=True)
    # SenseVoice's row count (88 row blocks of 256: 11/16 of the CUs with 256 x 256 blocks)
    for name in ("w1", "qkv", "w2"):
        N, K = SHAPES[name]
        a2, w2, b = operands(N, K, m=22528)
        row = {"shape": name, "data": "random", "M": 22528}
        for label, tile in (("t0", 0), ("t2", 2), ("ps", PS)):
            row[label + "_planes"] = round(best(lambda: ops.gemm_f16x2(a2, w2, b, scale_exp=20, tile=tile, relu=True, out_planes=True,
                                                                     out_scale_exp=9, time_iters=20)[1]), 1)
            row[label + "_fp32"] = round(best(lambda: ops.gemm_f16x2(a2, w2, b, scale_exp=20, tile=tile, time_iters=20)[1]), 1)
        print(json.dumps(row), flush=True)


if __name__ == "__main__":
    what = sys.argv[1] if len(sys.argv) > 1 else "parity"
    if what == "parity":
        sys.exit(0 if parity() else 1)
    elif what == "time":
        timing()
    else:
        raise SystemExit(__doc__)
