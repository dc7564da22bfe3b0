#!/usr/bin/env python3
"""BASELINE configs[2]: SenseVoiceSmall encoder-only (50 + 20 SAN-M blocks, CTC head 25055), batch 128 x 10 s clips on one
MI355X: wav (resident in HBM) -> fbank/LFR/CMVN -> 4 query frames + encoder -> CTC GEMM with fused arg-max -> ids on host.
Random-init weights of the exact architecture. Prints one JSON line (audio-seconds/s, ms per step, CPU-oracle check)."""
import argparse
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=128)
    ap.add_argument("--seconds", type=float, default=10.0)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--cpu-clips", type=int, default=4)
    ap.add_argument("--modes", default="f16x2,fp32,bf16", help="arithmetic modes to time; the first is the reported one")
    ap.add_argument("--enc-option", action="append", default=[], metavar="KEY=VALUE", help="A/B runs: encoder schedule options "
                    "(pf_encoder_set_option), e.g. row_bm=128 gemm_tile=2 = the block shapes before the row-count-aware choice")
    args = ap.parse_args()
    from funasr_amd import synth
    from funasr_amd.sense_voice import SenseVoiceSmall
    from funasr_amd.wav_frontend import WavFrontend

    dev = torch.device("cuda:0")
    cfg = synth.SENSEVOICE_SMALL
    sd = synth.sensevoice_state_dict(cfg, seed=0)
    model = SenseVoiceSmall.from_config(cfg)
    model.load_state_dict(sd, strict=False)
    model = model.to(dev)
    for kv in args.enc_option:
        k, v = kv.split("=")
        model.encoder.set_option(k, int(v))
    sh, sc = synth.synthetic_cmvn(560)
    cmvn = torch.stack([sh, sc])
    fe = WavFrontend(cmvn=cmvn, lfr_m=7, lfr_n=6, dither=0.0, device=dev)
    n = int(args.seconds * 16000)
    base = [synth.speech_like(n, seed=500 + i) for i in range(8)]
    clips = [base[i % 8].roll(97 * (i // 8)) for i in range(args.batch)]
    wav = torch.stack(clips).to(dev)
    lens = [n] * args.batch

    def enqueue():
        feats, flens = fe(wav, lens)
        return model.enqueue_features(feats, flens, "auto", "woitn")

    def run_steps(k):
        """k batches, software-pipelined like a serving loop (and like bench.py's Paraformer loop): batch i+1 is enqueued before
        batch i's frame ids are collapsed on the host; every batch is fully processed and collected inside the call"""
        pending = enqueue()
        for _ in range(k - 1):
            nxt = enqueue()
            model.collect(pending)
            pending = nxt
        return model.collect(pending)

    out = {}
    modes = args.modes.split(",")
    for mode in modes:
        model.set_precision(mode)
        if args.warmup > 0:
            run_steps(args.warmup)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        res = run_steps(args.steps)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        out[mode] = dict(value=round(args.batch * args.seconds * args.steps / dt, 1), ms_per_step=round(dt / args.steps * 1e3, 2),
                         res=res)
    main_mode = modes[0]
    line = {"metric": "audio-seconds/sec SenseVoiceSmall encoder+CTC, 10 s clips @ bs128", "value": out[main_mode]["value"],
            "unit": "audio-s/s", "ms_per_step": out[main_mode]["ms_per_step"], "mode": main_mode, "n_gpus": 1,
            "config": {"workload": f"SenseVoiceSmall (70 SAN-M blocks, CTC 25055, random-init), {args.batch} x {args.seconds:g} s"},
            "other_modes": {m: {"value": out[m]["value"], "ms_per_step": out[m]["ms_per_step"]} for m in modes[1:]}}
    if args.cpu_clips > 0:
        from oracle import paraformer_oracle as O
        ok = True
        t0 = time.perf_counter()
        with torch.no_grad():
            for i in range(args.cpu_clips):
                f, fl = O.wav_frontend([clips[i]], cmvn)
                ref = O.sensevoice_greedy(f, fl, sd, cfg)
                ok = ok and ref["ids"][0] == out[main_mode]["res"]["ids"][i]
        line.update(ids_equal_cpu_oracle=bool(ok), cpu_oracle_audio_s_per_s=round(args.cpu_clips * args.seconds / (time.perf_counter() - t0), 1))
    print(json.dumps(line), flush=True)


if __name__ == "__main__":
    main()
