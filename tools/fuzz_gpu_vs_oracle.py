#!/usr/bin/env python3
"""Shape fuzz on the GPU: random batch sizes / frame counts / ragged lengths through the HIP pipeline in all three
fp32-accurate modes against the CPU oracle (token ids, CIF fire positions, encoder error), plus the row-packed production call
in the f16x2 mode against the all-rows call. Not part of the test run."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from funasr_amd import synth                      # noqa: E402
from funasr_amd.paraformer import Paraformer      # noqa: E402
from oracle import paraformer_oracle as O         # noqa: E402

dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(int(sys.argv[1]) if len(sys.argv) > 1 else 0)
n_cases = int(sys.argv[2]) if len(sys.argv) > 2 else 16
worst, bad, n_packed, near_ties = {"fp32": 0.0, "bf16x3": 0.0, "f16x2": 0.0}, 0, 0, []
fire_shift_cases = []
for ci in range(n_cases):
    cfg = synth.tiny(synth.PARAFORMER_LARGE, enc_blocks=int(torch.randint(1, 4, (1,), generator=g)),
                     dec_blocks=int(torch.randint(1, 3, (1,), generator=g)), vocab=int(torch.randint(30, 300, (1,), generator=g)))
    cfg["decoder"]["num_blocks"] += int(torch.randint(0, 3, (1,), generator=g))        # 0..2 decoders2 blocks (no cross-attention)
    sd = synth.paraformer_state_dict(cfg, seed=500 + ci, cif_bias=float(torch.rand(1, generator=g)) * 0.8 - 0.2)
    model = Paraformer.from_config(cfg)
    model.load_state_dict(sd, strict=False)
    model = model.to(dev)
    # FUZZ_MAX_B / FUZZ_MAX_T: larger batches / longer clips than the default draw (e.g. 48 / 2500: three-minute clips, many key tiles)
    B = int(torch.randint(1, int(os.environ.get("FUZZ_MAX_B", "8")) + 1, (1,), generator=g))
    T = int(torch.randint(3, int(os.environ.get("FUZZ_MAX_T", "299")) + 1, (1,), generator=g))
    lens = torch.randint(1, T + 1, (B,), generator=g, dtype=torch.int32)
    lens[0] = T
    x = torch.randn(B, T, 560, generator=g) * 0.7
    for b in range(B):
        x[b, lens[b]:] = 0
    try:
        ref = O.paraformer_greedy(x, lens, sd, cfg)
    except IndexError:
        continue
    for mode in ("fp32", "bf16x3", "f16x2"):
        model.set_precision(mode)
        res = model.recognize_features(x.to(dev), lens, return_intermediate=True)
        err = (res["enc"].cpu() - ref["enc"]).abs().max().item()
        worst[mode] = max(worst[mode], err)
        same = res["token_num"] == ref["token_num"].tolist() and \
            torch.equal(torch.floor(res["peaks"].cpu()) >= 1, torch.floor(ref["peaks"]) >= 1)
        if same and res["raw_ids"] != ref["raw_ids"]:
            # random-init output layers: an id may differ only where the ORACLE's own top-2 logits are a near-tie (< 1e-4)
            top2 = torch.topk(ref["logits"], 2, dim=-1).values
            for b, (u, v) in enumerate(zip(res["raw_ids"], ref["raw_ids"])):
                for pos, (a_, b_) in enumerate(zip(u, v)):
                    if a_ != b_:
                        gap = float(top2[b, pos, 0] - top2[b, pos, 1])
                        near_ties.append(dict(case=ci, mode=mode, clip=b, pos=pos, cpu_top2_logit_gap=gap))
                        # (default draw: 1e-4, the bar of tests/test_parity_gpu.py; FUZZ_MAX_T > 1000 -- clips of minutes, token positions in
                        #  the hundreds, where the logits of the two float32 paths differ by up to ~3e-4 in EVERY mode incl. exact fp32 --
                        #  the activation bar 1e-3; every gap is recorded)
                        same = same and gap < (1e-3 if int(os.environ.get("FUZZ_MAX_T", "299")) > 1000 else 1e-4)
        if mode == "f16x2":
            # the production call: only len + 1 encoder rows per clip are computed (row packing) -- same integers
            packed = model.recognize_features(x.to(dev), lens)
            same = same and packed["raw_ids"] == res["raw_ids"] and packed["token_num"] == res["token_num"]
            n_packed += 1
        if not same or err > 1e-3:
            # classify: the documented statistical effect (a CIF prefix sum within float32 round-off of an integer fires one
            # frame earlier / later; ~2 % of 30-s clips, more for longer ones) against anything else
            gf, rf = torch.floor(res["peaks"].cpu()) >= 1, torch.floor(ref["peaks"]) >= 1
            shifted, other = 0, 0
            for b in range(B):
                if torch.equal(gf[b], rf[b]):
                    continue
                a_, b_ = set(torch.nonzero(gf[b]).flatten().tolist()), set(torch.nonzero(rf[b]).flatten().tolist())
                only_g, only_r = sorted(a_ - b_), sorted(b_ - a_)
                if len(only_g) == len(only_r) and all(abs(x - y) == 1 for x, y in zip(only_g, only_r)):
                    shifted += 1
                else:
                    other += 1
            counts_equal = res["token_num"] == ref["token_num"].tolist()
            alpha_d = (res["alphas"].cpu() - ref["alphas"]).abs().max().item()
            benign = other == 0 and counts_equal and err <= 1e-3 and shifted > 0
            if benign:
                fire_shift_cases.append(dict(case=ci, mode=mode, B=B, T=T, clips_with_a_fire_one_frame_off=shifted, alpha_max_abs_diff=alpha_d))
            else:
                bad += 1
            print(f"case {ci} mode {mode} B={B} T={T} err={err:.2e} same={same}: clips with a fire one frame off {shifted}, other fire differences {other}, "
                  f"token counts equal {counts_equal}, alpha |d| {alpha_d:.2e} -> {'fire-index statistic' if benign else 'FAILURE'}")
print(f"{n_cases} cases: max encoder |d| {worst}, failures {bad}")
import json
os.makedirs("gpurun_out", exist_ok=True)
json.dump({"cases": n_cases, "seed": int(sys.argv[1]) if len(sys.argv) > 1 else 0, "max_encoder_abs_diff": worst, "failures": bad,
           "row_packed_runs_compared": n_packed, "near_tie_token_flips": near_ties, "fire_one_frame_off_cases": fire_shift_cases},
          open("gpurun_out/fuzz_gpu_vs_oracle.json", "w"))
