#!/usr/bin/env python3
"""BiCifParaformer (the timestamp model behind `paraformer-zh`) at the headline shape: 64 x 30 s clips, 50 + 16 blocks,
vocabulary 8404, random-init weights; wav resident in HBM -> fbank/LFR/CMVN -> encoder -> CifPredictorV3 -> decoder ->
arg-max -> upsampled timestamp head (GEMMs + BLSTM + scan) -> ids and weights on the host. Prints one JSON line: audio-s/s
with the timestamps, the cost of the timestamp branch alone (HIP events), and the same step without it."""
import argparse
import copy
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=64)
    ap.add_argument("--seconds", type=float, default=30.0)
    ap.add_argument("--steps", type=int, default=4)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--precision", default="bf16x3")
    args = ap.parse_args()
    from funasr_amd import synth
    from funasr_amd.bicif_paraformer import BiCifParaformer
    from funasr_amd.wav_frontend import WavFrontend

    dev = torch.device("cuda:0")
    cfg = copy.deepcopy(synth.PARAFORMER_LARGE)
    cfg["predictor"] = dict(cfg["predictor"], smooth_factor2=0.25, noise_threshold2=0.01, upsample_times=3, use_cif1_cnn=False,
                            upsample_type="cnn_blstm")
    cfg["predictor"].pop("tail_mask", None)
    sd = synth.paraformer_state_dict(cfg, seed=0, cif_bias=synth.BENCH_CIF_BIAS)
    g = torch.Generator().manual_seed(5)
    D = 512
    sd["predictor.upsample_cnn.weight"] = torch.randn(D, D, 3, generator=g) / D ** 0.5
    sd["predictor.upsample_cnn.bias"] = torch.zeros(D)
    for sfx in ("", "_reverse"):
        sd[f"predictor.blstm.weight_ih_l0{sfx}"] = torch.randn(4 * D, D, generator=g) / D ** 0.5
        sd[f"predictor.blstm.weight_hh_l0{sfx}"] = torch.randn(4 * D, D, generator=g) * 0.7 / D ** 0.5
        sd[f"predictor.blstm.bias_ih_l0{sfx}"] = torch.zeros(4 * D)
        sd[f"predictor.blstm.bias_hh_l0{sfx}"] = torch.zeros(4 * D)
    sd["predictor.cif_output2.weight"] = torch.randn(1, 2 * D, generator=g) * 2.0 / (2 * D) ** 0.5
    sd["predictor.cif_output2.bias"] = torch.full((1,), -0.5)
    model = BiCifParaformer.from_config(cfg)
    model.load_state_dict(sd, strict=False)
    model = model.to(dev)
    model.set_precision(args.precision)
    sh, sc = synth.synthetic_cmvn(560)
    fe = WavFrontend(cmvn=torch.stack([sh, sc]), lfr_m=7, lfr_n=6, dither=0.0, device=dev)
    n = int(args.seconds * 16000)
    base = [synth.speech_like(n, seed=500 + i) for i in range(8)]
    wav = torch.stack([base[i % 8].roll(97 * (i // 8)) for i in range(args.batch)]).to(dev)
    lens = [n] * args.batch

    def step():
        feats, flens = fe(wav, lens)
        return model.recognize_features(feats, flens)

    def timed(fn):
        for _ in range(args.warmup):
            r = fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            r = fn()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / args.steps, r

    dt, res = timed(step)
    # the timestamp branch alone on the same encoder output
    feats, flens = fe(wav, lens)
    enc, olens = model.encode(feats, flens)
    tok = res["token_num"]
    for _ in range(2):
        model.calc_predictor_timestamp(enc, olens, tok)
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(args.steps):
        model.calc_predictor_timestamp(enc, olens, tok)
    b.record()
    torch.cuda.synchronize()
    ts_ms = a.elapsed_time(b) / args.steps
    T = enc.shape[1]
    flops = 2.0 * args.batch * T * 512 * 1536 + 2.0 * 2 * 2048 * 512 * (3 * T * args.batch) + 2.0 * 2 * 2048 * 512 * (3 * T * args.batch)
    print(json.dumps({"metric": "audio-seconds/sec BiCifParaformer (text + token timestamps), 30 s clips @ bs64",
                      "value": round(args.batch * args.seconds / dt, 1), "unit": "audio-s/s", "ms_per_step": round(dt * 1e3, 2),
                      "precision": args.precision, "n_gpus": 1, "tokens_per_clip": round(sum(tok) / len(tok), 1),
                      "timestamp_branch_ms": round(ts_ms, 2),
                      "timestamp_branch": {"frames": 3 * T, "lstm_steps": 3 * T, "gflop": round(flops / 1e9, 1),
                                           "tflops": round(flops / ts_ms / 1e9, 1)},
                      "config": {"workload": f"BiCifParaformer (50+16 blocks, CifPredictorV3 cnn_blstm x3, random-init), "
                                             f"{args.batch} x {args.seconds:g} s"}}))


if __name__ == "__main__":
    main()
