#!/usr/bin/env python3
"""Streaming step benchmark (BASELINE configs[4]: Paraformer-large-streaming, chunk 600 ms, hipGraph-captured step).

Times `pf_stream_step` for S lock-step streams on synthetic online features (random-init Paraformer-large online
architecture: 50 encoder blocks, 16 decoder blocks with sanm_shfit 5, vocab 8404), eager vs hipGraph replay.
Prints one JSON line per configuration: chunks/s, audio-seconds/s (= S * 0.6 s per step), step latency."""
import argparse
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--streams", type=int, nargs="+", default=[1, 16, 64])
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--precision", nargs="+", default=["fp32"], choices=["fp32", "f16x2"], help="GEMM arithmetic of the step "
                    "(StreamBatch precision): fp32 kernels, or two-plane fp16 operands on the fp16 matrix cores (fp32-class results)")
    ap.add_argument("--graph", nargs="+", type=int, default=[0, 1], help="0 eager, 1 hipGraph replay")
    ap.add_argument("--ln-carry", nargs="+", type=int, default=[1], help="fp32 step: 1 = LayerNorms carried by the small-M GEMMs "
                    "(default), 0 = stand-alone LayerNorm launches")
    ap.add_argument("--replicas", type=int, nargs="+", default=[1], help="R independent handles (own model objects, own HIP streams), "
                    "each with --streams streams, a step of every replica in flight at a time: R x S streams per wall step")
    ap.add_argument("--set", nargs="*", default=[], metavar="KEY=VALUE", help="pf_stream_set_option pairs applied to every "
                    "configuration (fsmn_rides, kv_batched: A/B of the step's launch fusions)")
    args = ap.parse_args()
    from funasr_amd import synth
    from funasr_amd.paraformer_streaming import ParaformerStreaming, StreamBatch
    import copy

    dev = torch.device("cuda:0")
    cfg = copy.deepcopy(synth.PARAFORMER_LARGE)
    cfg["decoder"]["sanm_shfit"] = 5
    sd = synth.paraformer_state_dict(cfg, seed=0, cif_bias=synth.BENCH_CIF_BIAS)
    models = []
    for _ in range(max(args.replicas)):
        m = ParaformerStreaming.from_config(cfg)
        m.load_state_dict(sd, strict=False)
        models.append(m.to(dev))
    model = models[0]
    first_ids = {}
    import itertools
    for S, prec, graph, carry, R in itertools.product(args.streams, args.precision, [bool(v) for v in args.graph], args.ln_carry, args.replicas):
        if prec != "fp32" and carry != args.ln_carry[0]:
            continue
        sbs = [StreamBatch(models[r], S, [0, 10, 5], 4, 1, use_graph=graph, pe_rows=16384, precision=prec) for r in range(R)]
        for b in sbs:
            b.set_option("ln_carry", carry)
            for kv in args.set:
                b.set_option(kv.split("=")[0], int(kv.split("=")[1]))

        class _All:                                   # every replica's step in flight, then every replica collected
            def step(self, f):
                for b in sbs:
                    b.step_begin(f)
                outs = [b.step_end() for b in sbs]
                return outs[0]

            def close(self):
                for b in sbs:
                    b.close()
        sb = _All()
        g = torch.Generator().manual_seed(S)
        feats = (torch.randn(S, 10, 560, generator=g) * 0.8).to(dev)
        ntok = 0
        trace = []
        for _ in range(args.warmup):
            trace.append(sb.step(feats))
        # the same session in every (precision, graph) setting: the warm-up steps' ids must agree (fp32-class arithmetic)
        same = first_ids.setdefault(S, trace) == trace
        torch.cuda.synchronize()
        tel = None
        try:                                      # shader clock / socket power of the timed region (tools/gpu_telemetry.py)
            sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
            from gpu_telemetry import Sampler
            tel = Sampler(0, period_s=0.005).start()
        except Exception:
            tel = None
        lat = []
        t0 = time.perf_counter()
        for _ in range(args.steps):
            t1 = time.perf_counter()
            ids = sb.step(feats)
            lat.append(time.perf_counter() - t1)
            ntok += sum(len(r) for r in ids)
        dt = time.perf_counter() - t0
        tstat = tel.stop() if tel is not None else {}
        lat.sort()
        print(json.dumps({"metric": "streaming chunks/s (600 ms chunk, Paraformer-large-online)", "streams": S,
                          "replicas": R, "hipgraph": graph, "steps": args.steps, "chunks_per_s": round(R * S * args.steps / dt, 1),
                          "audio_s_per_s": round(R * S * args.steps * 0.6 / dt, 1),
                          "step_ms_p50": round(lat[len(lat) // 2] * 1e3, 3), "step_ms_p99": round(lat[int(len(lat) * 0.99)] * 1e3, 3),
                          "tokens_per_chunk": round(ntok / (S * args.steps), 2),
                          "dtype": "f32" if prec == "fp32" else "f32 (GEMM operands as 2 fp16 planes, 3 fp16 MFMA products)",
                          "sclk_mhz_mean": tstat.get("sclk_mhz_mean"), "power_w_mean": tstat.get("power_w_mean"), "precision": prec, "ln_carry": bool(carry) if prec == "fp32" else None, "options": args.set, "warmup_ids_equal_first_setting": same}), flush=True)
        sb.close()


if __name__ == "__main__":
    main()
