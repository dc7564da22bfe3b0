#!/usr/bin/env python3
"""CPU baseline: is the oracle ("port") as fast as the reference's own nn.Modules on the same host?  Build container only
(needs /root/reference); writes profiles/r06_cpu_port_vs_reference.json.

bench.py's `cpu_baseline` runs on the GPU box, where /root/reference does not exist, so it times the oracle
(oracle/paraformer_oracle.py, `"kind": "port"`). This tool times, on ONE host and the SAME clips / thread counts,
  * the reference's own SANMEncoder + CifPredictorV2 + ParaformerSANMDecoder (imported from /root/reference, chained as
    Paraformer.inference does, funasr/models/paraformer/model.py:585-640), batch_size 1 like AutoModel on device="cpu"
    (funasr/auto/auto_model.py:551-566), and
  * the oracle on the same features,
and compares their outputs, so the port's number can stand in for the reference's with a measured ratio. The frontend
(fbank) is the oracle's in both legs: the reference's is torchaudio's, which this image does not have.
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--clips", type=int, default=4)
    ap.add_argument("--seconds", type=float, default=30.0)
    ap.add_argument("--threads", default="4,8")
    ap.add_argument("--repeats", type=int, default=2)
    ap.add_argument("--out", default=os.path.join(ROOT, "profiles", "r06_cpu_port_vs_reference.json"))
    args = ap.parse_args()
    from funasr_amd import synth
    from oracle import paraformer_oracle as O
    from oracle import make_golden_full as MG
    cfg = synth.PARAFORMER_LARGE
    sd = synth.paraformer_state_dict(cfg, seed=0, cif_bias=synth.BENCH_CIF_BIAS)
    shift, scale = synth.synthetic_cmvn(560)
    cmvn = torch.stack([shift, scale])
    clips = [synth.speech_like(int(args.seconds * 16000), seed=i) for i in range(args.clips)]
    enc, pred, dec = MG.build_reference_modules(cfg, sd)
    rows, equal = [], dict(encoder_bit_equal=True, token_ids_equal=True, token_counts_equal=True, alpha_max_abs_diff=0.0,
                           hidden_max_abs_diff=0.0)
    usable = len(os.sched_getaffinity(0))
    with torch.no_grad():
        for threads in [int(x) for x in args.threads.split(",")]:
            threads = min(threads, usable)
            torch.set_num_threads(threads)
            feats = [O.wav_frontend([c], cmvn) for c in clips]
            MG.run_reference(enc, pred, dec, *feats[0])            # warm both legs once
            O.paraformer_greedy(*feats[0], sd, cfg)
            t_ref = t_port = 0.0
            for _ in range(args.repeats):
                for f, fl in feats:
                    t0 = time.perf_counter()
                    r = MG.run_reference(enc, pred, dec, f, fl)
                    ids_r = torch.log_softmax(r["logits"][0, : int(r["token_num"][0])], -1).argmax(-1).tolist()
                    t1 = time.perf_counter()
                    p = O.paraformer_greedy(f, fl, sd, cfg)
                    t2 = time.perf_counter()
                    t_ref += t1 - t0
                    t_port += t2 - t1
                    equal["encoder_bit_equal"] &= bool(torch.equal(r["enc"], p["enc"]))
                    equal["token_ids_equal"] &= ids_r == p["raw_ids"][0]
                    equal["token_counts_equal"] &= int(r["token_num"][0]) == int(p["token_num"][0])
                    equal["alpha_max_abs_diff"] = max(equal["alpha_max_abs_diff"], float((r["alphas"] - p["alphas"]).abs().max()))
                    equal["hidden_max_abs_diff"] = max(equal["hidden_max_abs_diff"], float((r["hidden"] - p["hidden"]).abs().max()))
            audio = args.repeats * args.clips * args.seconds
            rows.append(dict(threads=threads, reference_modules_audio_s_per_s=round(audio / t_ref, 2),
                             port_audio_s_per_s=round(audio / t_port, 2), port_over_reference=round(t_ref / t_port, 4)))
            print(rows[-1], flush=True)
    out = dict(what="encoder + predictor + decoder + arg-max on LFR features, batch_size 1, fp32, same host, same clips; "
                    "reference = the nn.Modules imported from /root/reference, port = oracle/paraformer_oracle.py",
               host_cores_usable=usable, torch=torch.__version__, clips=args.clips, clip_seconds=args.seconds,
               repeats=args.repeats, settings=rows, outputs=equal)
    with open(args.out, "w") as f:
        json.dump(out, f, indent=1)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
