#!/usr/bin/env python3
"""Round-3 micro-benchmarks (one gpurun call, within-call A/B): attention_f16x2 schedules x workgroup order, gemm_f16x2 plain
vs de-phased rounds, the full-row form (fused LayerNorm) vs GEMM + layernorm_kernel. Prints one JSON object per line."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from funasr_amd import ops

dev = torch.device("cuda:0")
which = set(sys.argv[1:]) or {"attn", "gemm", "row"}
M = 32768


def best(fn, n=3):
    return min(fn() for _ in range(n))


if "attn" in which:
    B, T, H = 64, 512, 4
    g = torch.Generator().manual_seed(0)
    q = torch.randn(B, T, 512, generator=g).to(dev); k = torch.randn(B, T, 512, generator=g).to(dev); v = torch.randn(B, T, 512, generator=g).to(dev)
    klens = torch.full((B,), 500, dtype=torch.int32, device=dev)
    row = {}
    for variant in (0, 1, 3):
        for plain in (0, 16):
            ms = best(lambda: ops.attention_f16x2(q, k, v, klens, H, 128 ** -0.5, variant=variant + plain, time_iters=20)[1])
            row[f"v{variant}{'_plain' if plain else '_xcd'}"] = round(ms * 1e3, 1)
    print(json.dumps({"attention_f16x2_us_B64_T512": row}), flush=True)
    # cross-attention shape of the decoder: 172 tokens x 512 keys
    qx = torch.randn(B, 176, 512, generator=g).to(dev)
    row = {f"v{variant}": round(best(lambda: ops.attention_f16x2(qx, k, v, klens, H, 128 ** -0.5, variant=variant, time_iters=20)[1]) * 1e3, 1) for variant in (0, 1, 3)}
    print(json.dumps({"cross_attention_us_B64_Tq176": row}), flush=True)

if "gemm" in which:
    for name, N, K, kw in (("qkv_fp32out", 1536, 512, {}), ("w1_planes", 2048, 512, dict(relu=True, out_planes=True, out_scale_exp=9)),
                           ("w2_resid", 512, 2048, dict(resid=True)), ("out_resid", 512, 512, dict(resid=True)), ("out_fp32", 512, 512, {})):
        a = torch.randn(M, K, device=dev); w = torch.randn(N, K, device=dev) * K ** -0.5; b = torch.randn(N, device=dev)
        a2, w2 = ops.split2(a, 8), ops.split2(w, 12)
        kw = dict(kw)
        if kw.pop("resid", False):
            kw["add2"] = torch.randn(M, N, device=dev)
        row = {}
        ref = ops.gemm_f16x2(a2, w2, b, scale_exp=20, tile=2, **kw)
        a2k, w2k = ops.kblocked(a2), ops.kblocked(w2)
        variants = (("wide", 2, False), ("ring", 6, False), ("wide_again", 2, False), ("ring_again", 6, False)) if "ring" in which else \
                   (("wide", 2, False), ("wide_kblock", 2, True), ("wide_wblock", 2, "w"), ("wide_ablock", 2, "a"),
                    ("wide_again", 2, False), ("wide_kblock_again", 2, True), ("wide_wblock_again", 2, "w"))
        for label, tile, kb in variants:
            aa = a2k if kb in (True, "a") else a2
            ww = w2k if kb in (True, "w") else w2
            out = ops.gemm_f16x2(aa, ww, b, scale_exp=20, tile=tile, kblock=kb, **kw)
            ms = best(lambda: ops.gemm_f16x2(aa, ww, b, scale_exp=20, tile=tile, time_iters=20, kblock=kb, **kw)[1])
            row[label] = (round(ms * 1e3, 1), bool(torch.equal(out, ref)))
        print(json.dumps({name: row}), flush=True)
    # the QKV form (the engine's projection: Q / K planes, fp32 V, V^T planes)
    a = torch.randn(M, 512, device=dev); w = torch.randn(1536, 512, device=dev) * 512 ** -0.5; b = torch.randn(1536, device=dev)
    a2, w2 = ops.split2(a, 8), ops.split2(w, 12)
    row = {label: round(best(lambda: ops.gemm_f16x2_qkv(a2, w2, b, 512, 20, 2.0 ** 4, 2.0 ** 6, 2.0 ** 6, tile=tile, time_iters=20)["ms"]) * 1e3, 1)
           for label, tile in (("wide", 0), ("ring", 6), ("wide_again", 0), ("ring_again", 6))}
    print(json.dumps({"qkv_form_us": row}), flush=True)

if "row" in which:
    gamma = torch.rand(512, device=dev) + 0.5; beta = torch.randn(512, device=dev)
    for name, K, r1 in (("out_proj+ln2", 512, True), ("w2+ln1", 2048, False)):
        a = torch.randn(M, K, device=dev); w = torch.randn(512, K, device=dev) * K ** -0.5; b = torch.randn(512, device=dev)
        a2, w2 = ops.split2(a, 8), ops.split2(w, 12)
        add1 = torch.randn(M, 512, device=dev) if r1 else None
        add2 = torch.randn(M, 512, device=dev)
        row = {}
        for tile, label in ((2, "wide"), (1, "narrow")):
            row[f"gemm_{label}"] = round(best(lambda: ops.gemm_f16x2(a2, w2, b, add1=add1, add2=add2, scale_exp=20, tile=tile, time_iters=20)[1]) * 1e3, 1)
        c = ops.gemm_f16x2(a2, w2, b, add1=add1, add2=add2, scale_exp=20, tile=2)
        row["layernorm_planes"] = round(best(lambda: ops.layernorm_planes(c, gamma, beta, 1e-12, scale_exp=7, time_iters=20)[1]) * 1e3, 1)
        for nt in (False, True):
            row[f"row_fused{'_nt' if nt else ''}"] = round(best(lambda: ops.gemm_f16x2_row(a2, w2, b, add1=add1, add2=add2, scale_exp=20, ln=(gamma, beta, 1e-12),
                                                                                          out_scale_exp=7, a_nt=nt, time_iters=20)[2]) * 1e3, 1)
        row["row_no_ln"] = round(best(lambda: ops.gemm_f16x2_row(a2, w2, b, add1=add1, add2=add2, scale_exp=20, time_iters=20)[2]) * 1e3, 1)
        print(json.dumps({name: row}), flush=True)

if "dec" in which:
    # decoder token-side shapes (11 008 token rows = 64 clips x 172 tokens): which block shape is fastest at this M?
    Md = 11008
    for name, N, K, kw in (("dec_w1", 2048, 512, dict(relu=True)), ("dec_w2", 512, 2048, {}), ("dec_q_planes", 512, 512, dict(out_planes=True, out_scale_exp=6)),
                           ("dec_out_resid", 512, 512, dict(resid=True))):
        a = torch.randn(Md, K, device=dev); w = torch.randn(N, K, device=dev) * K ** -0.5; b = torch.randn(N, device=dev)
        a2, w2 = ops.split2(a, 8), ops.split2(w, 12)
        kw = dict(kw)
        if kw.pop("resid", False):
            kw["add2"] = torch.randn(Md, N, device=dev)
        row = {}
        tiles = (("auto", 0), ("wide", 2), ("narrow", 1), ("pair", 5)) + ((("small128", 3),) if not kw.get("out_planes") else ())
        for label, tile in tiles:
            row[label] = round(best(lambda: ops.gemm_f16x2(a2, w2, b, scale_exp=20, tile=tile, time_iters=20, **kw)[1]) * 1e3, 1)
        print(json.dumps({name: row}), flush=True)

if "fsmnrow" in which:
    # linear_out at the bench shape: fsmn_kernel + row kernel (memory as first addend) vs the FSMN form (memory in the epilogue)
    B, T = 64, 512
    g = torch.Generator().manual_seed(3)
    a = torch.randn(M, 512, device=dev); w = torch.randn(512, 512, device=dev) * 512 ** -0.5; b = torch.randn(512, device=dev)
    a2, w2 = ops.split2(a, 8), ops.split2(w, 12)
    v = torch.randn(M, 512, device=dev); taps = torch.randn(512, 11, device=dev) * 0.3; x = torch.randn(M, 512, device=dev)
    gamma = torch.rand(512, device=dev) + 0.5; beta = torch.randn(512, device=dev)
    lens = torch.full((B,), 500, dtype=torch.int32, device=dev)
    lo = (torch.arange(M // 16, dtype=torch.int32) // (T // 16) * T).to(dev); hi = lo + 500
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)

    def fsmn_ms():
        ops.fsmn(v.view(B, T, 512), taps, lens, 5)
        ev0.record()
        for _ in range(20):
            ops.fsmn(v.view(B, T, 512), taps, lens, 5)
        ev1.record(); torch.cuda.synchronize()
        return ev0.elapsed_time(ev1) / 20
    mem = ops.fsmn(v.view(B, T, 512), taps, lens, 5).view(M, 512)
    ln = (gamma, beta, 1e-12)
    c0, y0 = ops.gemm_f16x2_row(a2, w2, b, add1=mem, add2=x, scale_exp=20, ln=ln, out_scale_exp=7, a_nt=True)
    c1, y1 = ops.gemm_f16x2_row_fsmn(a2, w2, b, v, taps, lo, hi, add2=x, scale_exp=20, ln=ln, out_scale_exp=7, a_nt=True)
    row = {"bitwise": bool(torch.equal(c0, c1) and torch.equal(y0, y1)), "fsmn_kernel": round(best(fsmn_ms) * 1e3, 1)}
    for rep in ("", "_again"):
        row["row_r1" + rep] = round(best(lambda: ops.gemm_f16x2_row(a2, w2, b, add1=mem, add2=x, scale_exp=20, ln=ln, out_scale_exp=7, a_nt=True, time_iters=20)[2]) * 1e3, 1)
        row["row_fsmn" + rep] = round(best(lambda: ops.gemm_f16x2_row_fsmn(a2, w2, b, v, taps, lo, hi, add2=x, scale_exp=20, ln=ln, out_scale_exp=7, a_nt=True, time_iters=20)[2]) * 1e3, 1)
    print(json.dumps({"linear_out_us_M32768": row}), flush=True)

if "row8" in which:
    # block heights of the full-row kernel at the SenseVoice row count (22 528 = 0.69 of a round of 128-row blocks) and at
    # the headline's (32 768 = one round): 128 = 2 x 4 waves, 129 = 1 x 8 waves x 128 rows, 96 = 1 x 8 waves x 96 rows
    for Mr in (22528, 32768, 37120):
        for K in (512, 2048):
            a = torch.randn(Mr, K, device=dev); w = torch.randn(512, K, device=dev) * K ** -0.5; b = torch.randn(512, device=dev)
            a2, w2 = ops.split2(a, 8), ops.split2(w, 12)
            x = torch.randn(Mr, 512, device=dev); gamma = torch.rand(512, device=dev) + 0.5; beta = torch.randn(512, device=dev)
            ln = (gamma, beta, 1e-12)
            ref = ops.gemm_f16x2_row(a2, w2, b, add2=x, scale_exp=20, ln=ln, out_scale_exp=7, block_rows=128)
            row = {}
            for rep in ("", "_again"):
                for br in (128, 129, 96, 0):
                    out = ops.gemm_f16x2_row(a2, w2, b, add2=x, scale_exp=20, ln=ln, out_scale_exp=7, block_rows=br, a_nt=(K == 512))
                    ms = best(lambda: ops.gemm_f16x2_row(a2, w2, b, add2=x, scale_exp=20, ln=ln, out_scale_exp=7, block_rows=br,
                                                         a_nt=(K == 512), time_iters=20)[2])
                    row[f"bm{br}{rep}"] = (round(ms * 1e3, 1), bool(torch.equal(out[0], ref[0]) and torch.equal(out[1], ref[1])))
            print(json.dumps({f"row_M{Mr}_K{K}": row}), flush=True)

if "qkvsplit" in which:
    # QKV form at row counts that leave a nearly empty last round of 256 x 256 blocks: tile 0 = head / tail split, 2 = no split
    for Mr in (22528, 32768, 33536):
        a = torch.randn(Mr, 512, device=dev); w = torch.randn(1536, 512, device=dev) * 512 ** -0.5; b = torch.randn(1536, device=dev)
        a2, w2 = ops.split2(a, 8), ops.split2(w, 12)
        row = {}
        for rep in ("", "_again"):
            for label, tile in (("wide_only", 2), ("auto", 0), ("small_only", 3)):
                row[label + rep] = round(best(lambda: ops.gemm_f16x2_qkv(a2, w2, b, 512, 20, 2.0 ** 4, 2.0 ** 6, 2.0 ** 6, tile=tile, time_iters=20)["ms"]) * 1e3, 1)
        print(json.dumps({f"qkv_form_M{Mr}": row}), flush=True)
