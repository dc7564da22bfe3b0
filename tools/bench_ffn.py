#!/usr/bin/env python3
"""Micro-benchmark: the encoder block's feed-forward as the two-kernel pair (gemm_f16x2 w_1 with plane output ->
gemm_f16x2_row w_2 + residual + LayerNorm) against the one-launch form (gemm_f16x2_ffn.hip), same operands, random data,
HIP-event timing inside the library (pf_k_* time_iters). One JSON line per row count."""
import argparse
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rows", default="32768,22528,37120,11000")
    ap.add_argument("--iters", type=int, default=30)
    ap.add_argument("--ffn", type=int, default=2048)
    ap.add_argument("--ablate", action="store_true", help="also time the measurement-only variants of the fused kernel (wrong results)")
    args = ap.parse_args()
    from funasr_amd import ops
    dev = torch.device("cuda:0")
    F = args.ffn
    g = torch.Generator().manual_seed(1)
    e_x, e_w1, e_h, e_w2 = 8, 12, 6, 12
    w1 = ops.split2((torch.randn(F, 512, generator=g) * 512 ** -0.5).to(dev), e_w1)
    w2 = ops.split2((torch.randn(512, F, generator=g) * F ** -0.5).to(dev), e_w2)
    b1, b2 = (torch.randn(F, generator=g) * 0.3).to(dev), torch.randn(512, generator=g).to(dev)
    gamma, beta = (torch.rand(512, generator=g) + 0.5).to(dev), torch.randn(512, generator=g).to(dev)
    for M in [int(x) for x in args.rows.split(",")]:
        x2 = ops.split2(torch.randn(M, 512, generator=g).to(dev), e_x)
        resid = (torch.randn(M, 512, generator=g) * 3).to(dev)
        ln = (gamma, beta, 1e-12)
        for _ in range(2):                                    # warm both paths
            h2 = ops.gemm_f16x2(x2, w1, b1, relu=True, scale_exp=e_x + e_w1, out_planes=True, out_scale_exp=e_h)
            ops.gemm_f16x2_row(h2, w2, b2, add2=resid, scale_exp=e_h + e_w2, ln=ln, out_scale_exp=7)
            ops.ffn_f16x2(x2, w1, w2, b1, b2, resid, e_x, e_w1, e_h, e_w2, ln=ln, out_scale_exp=7)
        best = {}
        for rep in range(3):
            h2, ms1 = ops.gemm_f16x2(x2, w1, b1, relu=True, scale_exp=e_x + e_w1, out_planes=True, out_scale_exp=e_h, time_iters=args.iters)
            _, _, ms2 = ops.gemm_f16x2_row(h2, w2, b2, add2=resid, scale_exp=e_h + e_w2, ln=ln, out_scale_exp=7, time_iters=args.iters)
            _, _, msf = ops.ffn_f16x2(x2, w1, w2, b1, b2, resid, e_x, e_w1, e_h, e_w2, ln=ln, out_scale_exp=7, time_iters=args.iters)
            for k, v in (("w_1_us", ms1), ("w_2_row_us", ms2), ("fused_us", msf)):
                best[k] = min(best.get(k, 1e9), v * 1e3)
        w1k, w2k = ops.kblocked(w1), ops.kblocked(w2)
        ck, yk = ops.ffn_f16x2(x2, w1k, w2k, b1, b2, resid, e_x, e_w1, e_h, e_w2, ln=ln, out_scale_exp=7, w_kblocked=True)
        cr, yr = ops.ffn_f16x2(x2, w1, w2, b1, b2, resid, e_x, e_w1, e_h, e_w2, ln=ln, out_scale_exp=7)
        best["fused_kblocked_w_us"] = min(ops.ffn_f16x2(x2, w1k, w2k, b1, b2, resid, e_x, e_w1, e_h, e_w2, ln=ln, out_scale_exp=7,
                                                        w_kblocked=True, time_iters=args.iters)[2] for _ in range(3)) * 1e3
        best["kblocked_bitwise_equal"] = bool(torch.equal(ck, cr) and torch.equal(yk, yr))
        flops = 2 * 2.0 * M * 512 * F
        row = {"M": M, "F": F, **{k: (round(v, 1) if not isinstance(v, bool) else v) for k, v in best.items()}, "pair_us": round(best["w_1_us"] + best["w_2_row_us"], 1),
               "fused_over_pair": round(best["fused_us"] / (best["w_1_us"] + best["w_2_row_us"]), 3),
               "fused_tflops_fp32_equiv": round(flops / (best["fused_us"] * 1e-6) / 1e12, 1),
               "pair_tflops_fp32_equiv": round(flops / ((best["w_1_us"] + best["w_2_row_us"]) * 1e-6) / 1e12, 1)}
        if args.ablate:
            for abl, name in ((1, "no_dma"), (2, "no_first_product"), (3, "no_second_product"), (4, "no_fragment_reads"), (5, "no_xn_dma"), (6, "no_weight_dma"),
                              (7, "no_w1_dma"), (8, "no_w2_dma"), (9, "same_xn_tile_everywhere"),
                              (12, "counted_waits_WRONG_RESULTS")):
                _, _, msa = ops.ffn_f16x2(x2, w1, w2, b1, b2, resid, e_x, e_w1, e_h, e_w2, ln=ln, out_scale_exp=7, time_iters=args.iters, abl=abl)
                row["fused_" + name + "_us"] = round(msa * 1e3, 1)
        print(json.dumps(row), flush=True)


if __name__ == "__main__":
    main()
