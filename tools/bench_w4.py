#!/usr/bin/env python3
"""Round 5 measurement of the four-wave 256 x 256 f16x2 GEMM shape (csrc/gemm_f16x2_w4.hip, Gemm2Args.tile 7) and of the
matrix pipe's energy roofline for its instruction mix (tools/micro/mfma_peak.hip). Measurement infrastructure, not product code.

    python tools/bench_w4.py parity         bitwise tile 7 == tile 2 on the encoder's shapes (fp32 / planes / QKV forms, ragged M)
    python tools/bench_w4.py time           us per launch: tile 2 vs tile 7, with the ablation builds of both
    python tools/bench_w4.py peak           register-resident MFMA chains: TFLOP/s, MHz, W on random / zero planes
    python tools/bench_w4.py sustained      w_1 on tile 2 / 7 for a few seconds each with MHz / W sampled
Every section prints JSON lines (copied to profiles/r05*_*.jsonl by the caller)."""
import ctypes as C
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
import torch  # noqa: E402
from funasr_amd import ops  # noqa: E402
from gpu_telemetry import Sampler  # noqa: E402

dev = torch.device("cuda:0")
M = 32768
SHAPES = {"qkv": (1536, 512), "w1": (2048, 512), "w2": (512, 2048), "out": (512, 512)}


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
    for m in (32768, 32768 - 100, 300, 16, 4096 + 16):
        for name, (N, K) in SHAPES.items():
            a2, w2, b = operands(N, K, m)
            r1 = torch.randn(m, N, device=dev)
            r2 = torch.randn(m, N, device=dev)
            row = {"M": m, "shape": name}
            for label, kw in (("fp32", {}), ("fp32+relu", dict(relu=True)), ("fp32+res", dict(add1=r1, add2=r2)),
                              ("planes", dict(relu=True, out_planes=True, out_scale_exp=9))):
                ref = ops.gemm_f16x2(a2, w2, b, scale_exp=20, tile=2, **kw)
                out = ops.gemm_f16x2(a2, w2, b, scale_exp=20, tile=7, **kw)
                row[label] = bool(torch.equal(out, ref))
                ok_all &= row[label]
            if name == "qkv" and m % 16 == 0:
                ref = ops.gemm_f16x2_qkv(a2, w2, b, 512, 20, 2.0 ** 3, 2.0 ** 4, 2.0 ** 5, tile=2)
                out = ops.gemm_f16x2_qkv(a2, w2, b, 512, 20, 2.0 ** 3, 2.0 ** 4, 2.0 ** 5, tile=7)
                row["qkv_form"] = all(bool(torch.equal(out[k], ref[k])) for k in ("q2", "k2", "v", "vt"))
                ok_all &= row["qkv_form"]
            print(json.dumps(row), flush=True)
    # the same launch many times: a stale ring buffer shows up as run-to-run differences
    a2, w2, b = operands(2048, 512)
    ref = ops.gemm_f16x2(a2, w2, b, scale_exp=20, tile=2, relu=True, out_planes=True, out_scale_exp=9)
    bad = sum(0 if torch.equal(ops.gemm_f16x2(a2, w2, b, scale_exp=20, tile=7, relu=True, out_planes=True, out_scale_exp=9), ref) else 1
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
            row = {"shape": name, "data": data, "M": M}
            for label, tile in (("t2_full", 2), ("t2_nostore", 2 + 16), ("t2_noepi", 2 + 32), ("t2_loop", 2 + 48),
                                ("t7_full", 7), ("t7_nostore", 0x17), ("t7_noepi", 0x27), ("t7_loop", 0x37), ("t7_loop2", 0x37)):
                row[label] = round(best(lambda: ops.gemm_f16x2(a2, w2, b, scale_exp=20, tile=tile, time_iters=20)[1]), 1)
            if name == "w1":
                for label, tile in (("t2_planes", 2), ("t7_planes", 7)):
                    row[label] = round(best(lambda: ops.gemm_f16x2(a2, w2, b, scale_exp=20, tile=tile, relu=True, out_planes=True,
                                                                   out_scale_exp=9, time_iters=20)[1]), 1)
            if name == "qkv":
                for label, tile in (("t2_qkvform", 2), ("t7_qkvform", 7)):
                    row[label] = round(best(lambda: ops.gemm_f16x2_qkv(a2, w2, b, 512, 20, 8.0, 16.0, 32.0, tile=tile, time_iters=20)["ms"]), 1)
            fl = 2.0 * M * N * K * 3
            row["exec_TFLOPs_t2_full"] = round(fl / row["t2_full"] / 1e6)
            row["exec_TFLOPs_t7_full"] = round(fl / row["t7_full"] / 1e6)
            row["exec_TFLOPs_t7_loop"] = round(fl / row["t7_loop"] / 1e6)
            print(json.dumps(row), flush=True)


def rows():
    """the full-row forms (N = 512): block_rows 128 (eight waves 2 x 4), 129 (eight waves 1 x 8) and 130 (four waves 1 x 4): bitwise
    equality and us per launch at M = 32768 for w_2 (K = 2048, residual + LayerNorm planes) and linear_out (K = 512, FSMN form)"""
    g = torch.Generator(device=dev).manual_seed(11)
    gam, bet = torch.rand(512, device=dev, generator=g) + 0.5, torch.randn(512, device=dev, generator=g)
    for data in ("random", "zeros"):
        for name, K in (("w2_row", 2048), ("out_row", 512)):
            a2, w2, b = operands(512, K, data=data)
            res = torch.randn(M, 512, device=dev, generator=g)
            row = {"shape": name, "data": data, "M": M}
            ref = ops.gemm_f16x2_row(a2, w2, b, add2=res, scale_exp=20, ln=(gam, bet, 1e-12), out_scale_exp=8, block_rows=128)
            for br in (128, 129, 130):
                c, y = ops.gemm_f16x2_row(a2, w2, b, add2=res, scale_exp=20, ln=(gam, bet, 1e-12), out_scale_exp=8, block_rows=br)
                row[f"bits_{br}"] = bool(torch.equal(c, ref[0]) and torch.equal(y, ref[1]))
                row[f"us_{br}"] = round(best(lambda: ops.gemm_f16x2_row(a2, w2, b, add2=res, scale_exp=20, ln=(gam, bet, 1e-12), out_scale_exp=8,
                                                                        block_rows=br, a_nt=(K == 512), time_iters=20)[2]), 1)
            if K == 512:
                v = torch.randn(M, 512, device=dev, generator=g)
                taps = torch.randn(512, 11, device=dev, generator=g) * 0.3
                lo = (torch.arange(M // 16, device=dev, dtype=torch.int32) // 32) * 512
                hi = lo + 500
                reff = ops.gemm_f16x2_row_fsmn(a2, w2, b, v, taps, lo, hi, add2=res, scale_exp=20, ln=(gam, bet, 1e-12), out_scale_exp=8, block_rows=128)
                for br in (128, 130):
                    c, y = ops.gemm_f16x2_row_fsmn(a2, w2, b, v, taps, lo, hi, add2=res, scale_exp=20, ln=(gam, bet, 1e-12), out_scale_exp=8, block_rows=br)
                    row[f"fsmn_bits_{br}"] = bool(torch.equal(c, reff[0]) and torch.equal(y, reff[1]))
                    row[f"fsmn_us_{br}"] = round(best(lambda: ops.gemm_f16x2_row_fsmn(a2, w2, b, v, taps, lo, hi, add2=res, scale_exp=20, ln=(gam, bet, 1e-12),
                                                                                    out_scale_exp=8, block_rows=br, a_nt=True, time_iters=20)[2]), 1)
            print(json.dumps(row), flush=True)


def peak():
    so = os.path.join(ROOT, "tools", "micro", "mfma_peak.so")
    lib = C.CDLL(so)
    lib.mfma_peak_run.restype = C.c_float
    lib.mfma_peak_run.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p]
    scratch = torch.zeros(16, device=dev)
    g = torch.Generator(device=dev).manual_seed(3)
    for label, bf16 in (("f16", 0), ("bf16", 1)):
        for data in ("random", "zeros"):
            if data == "random":
                # the planes a GEMM sees: hi = f16(x 2^8), lo = f16(x 2^8 - hi) of N(0, 1) values (bf16: the same bits are fine as a toggle pattern)
                x = torch.randn(16 * 64 * 8, device=dev, generator=g) * 256.0
                hi = x.to(torch.float16)
                lo = (x - hi.float()).to(torch.float16)
                pl = torch.stack([hi[:4 * 64 * 8], lo[:4 * 64 * 8], hi[4 * 64 * 8:8 * 64 * 8], lo[4 * 64 * 8:8 * 64 * 8]]).contiguous()
            else:
                pl = torch.zeros(4, 4 * 64 * 8, device=dev, dtype=torch.float16)
            iters, blocks = 2000, 256
            flops = blocks * 4 * iters * 48 * 32768.0
            lib.mfma_peak_run(pl.data_ptr(), bf16, blocks, iters, 3, scratch.data_ptr())     # warm
            smp = Sampler(period_s=0.05).start()
            t0, ms = time.time(), []
            while time.time() - t0 < 3.0:
                ms.append(lib.mfma_peak_run(pl.data_ptr(), bf16, blocks, iters, 20, scratch.data_ptr()))
            tel = smp.stop()
            tail = ms[len(ms) // 2:]
            m = sum(tail) / len(tail)
            print(json.dumps({"kernel": "mfma_peak", "dtype": label, "data": data, "waves_per_simd": 1, "ms_per_launch": round(m, 4),
                              "TFLOPs": round(flops / m / 1e9, 1), "frac_of_2500": round(flops / m / 1e9 / 2500.0, 3), "telemetry": tel}), flush=True)


def sustained():
    for name in ("w1", "qkv"):
        N, K = SHAPES[name]
        for data in ("random", "zeros"):
            a2, w2, b = operands(N, K, data=data)
            for tile in (2, 7, 0x37):
                kw = dict(relu=True, out_planes=True, out_scale_exp=9) if (name == "w1" and tile < 16) else {}
                ops.gemm_f16x2(a2, w2, b, scale_exp=20, tile=tile, time_iters=20, **kw)
                smp = Sampler(period_s=0.05).start()
                t0, us = time.time(), []
                while time.time() - t0 < 3.0:
                    us.append(ops.gemm_f16x2(a2, w2, b, scale_exp=20, tile=tile, time_iters=200, **kw)[1] * 1e3)
                tel = smp.stop()
                tail = us[len(us) // 2:]
                print(json.dumps({"kernel": "gemm_f16x2", "shape": name, "data": data, "tile": tile, "us": round(sum(tail) / len(tail), 1),
                                  "exec_TFLOPs": round(2.0 * M * N * K * 3 / (sum(tail) / len(tail)) / 1e6), "telemetry": tel}), flush=True)


if __name__ == "__main__":
    what = sys.argv[1:] or ["parity", "time", "peak", "sustained"]
    rc = 0
    for w in what:
        print(json.dumps({"section": w}), flush=True)
        r = {"parity": parity, "time": timing, "peak": peak, "sustained": sustained, "rows": rows}[w]()
        if w == "parity" and not r:
            rc = 1
    sys.exit(rc)
