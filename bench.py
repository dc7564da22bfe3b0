#!/usr/bin/env python3
"""Headline benchmark: audio-seconds per wall-second (RTF^-1) of the full Paraformer-large hot path on MI355X.

One "step" = one pass of the hot path over one batch of synthetic input that is already resident in HBM:
  [B, 480000] float32 PCM -> fbank/LFR/CMVN -> 50-block SAN-M encoder -> CIF predictor -> 16-block SAN-M decoder
  -> fused vocabulary arg-max -> token ids on the host (one D2H copy), i.e. BASELINE.json configs[1]
  ("Paraformer-large, batch=64 synthetic 30 s 16 kHz clips, 1 x MI355X"), random-initialised weights of the exact
  architecture (funasr_amd/synth.py). Arithmetic is fp32: the dense GEMMs run on the bf16 matrix cores with every operand
  split into three bf16 planes (x = hi + mid + lo exactly, six products, fp32 accumulate: gemm_split3.hip, mode
  "bf16x3"), everything else on the fp32 kernels; this mode meets every fp32 parity bar of tests/test_parity_gpu.py.
  The all-fp32-MFMA mode ("fp32") and the bf16-operand throughput mode ("bf16") are timed beside it.

Multi-GPU (utterance-level data parallelism, weak scaling): one process per GPU, rank 0 builds the weights and
broadcasts ONE packed arena over RCCL, every rank decodes its own 64 clips, hypotheses are gathered on rank 0
inside the timed region. No collective touches the data path.

Contract: python bench.py --gpus N --steps K --warmup W   (N>1 is launched through torch.distributed.run)
prints ONE JSON line on rank 0 (see README / DESIGN.md for the fields, incl. `roofline` and `cpu_baseline`).
"""
from __future__ import annotations

import argparse
import ctypes as C
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_F32_MFMA_TFLOPS = 157.3      # /opt/skills/guides/MI355X_MICROARCH.md, "Peak FP32 (matrix)"
PEAK_BF16_MFMA_TFLOPS = 2500.0    # same guide, dense bf16 (never the 2:1-sparsity figure)
SPLIT3_PRODUCTS = 6               # bf16 MFMA products per fp32-equivalent product in gemm_split3.hip
PEAK_HBM_GBS = 8000.0


_T0 = time.perf_counter()


def trace(msg):
    """stage timestamps on stderr (stdout carries only the JSON line)"""
    print(f"[bench +{time.perf_counter() - _T0:7.2f}s] {msg}", file=sys.stderr, flush=True)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--batch", type=int, default=64, help="clips per GPU per step")
    ap.add_argument("--seconds", type=float, default=30.0, help="clip length")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-bf16", action="store_true", help="skip the secondary measurements (fp32-MFMA mode, bf16-operand mode)")
    ap.add_argument("--precision", default="bf16x3", choices=["fp32", "bf16", "bf16x3", "f16x2"], help="mode of the MAIN timed region: "
                    "bf16x3 (default) = fp32 results, GEMM operands as three bf16 planes on the bf16 MFMA; fp32 = every GEMM on "
                    "the fp32 MFMA; bf16 = bf16 operands (bf16-class error; for profiling the throughput mode)")
    ap.add_argument("--dist-backend", default="nccl", help="nccl (= RCCL over xGMI) | gloo (dry run of the N>1 code "
                    "path with every rank on cuda:0 of a single-GPU box)")
    ap.add_argument("--cpu-clips", type=int, default=64, help="max clips of the same workload timed on the host "
                    "oracle (stops after ~20 s of CPU work)")
    ap.add_argument("--cpu-threads", type=int, default=0, help="cap on host threads for the CPU baseline (0 = all usable)")
    return ap.parse_args()


def build_model(cfg, rank, world, device):
    """rank 0 draws the synthetic checkpoint; with world > 1 it is shipped as one packed arena (RCCL broadcast)."""
    from funasr_amd import synth
    from funasr_amd.paraformer import Paraformer
    import torch.distributed as dist

    model = Paraformer.from_config(cfg)
    names = [n for n, _ in model.named_parameters()]
    if rank == 0:
        sd = synth.paraformer_state_dict(cfg, seed=0, cif_bias=synth.BENCH_CIF_BIAS)
        model.load_state_dict(sd, strict=False)
    model = model.to(device)
    if world > 1:
        from funasr_amd import dp
        nbytes = dp.broadcast_model(model, src=0)          # ONE packed fp32 arena over RCCL (~880 MB)
        trace(f"rank {rank}: weight arena broadcast, {nbytes / 1e6:.0f} MB")
    return model, names


def main():
    args = parse()
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an AMD GPU: the HIP path is the only implementation")
    if args.dist_backend == "gloo":
        local_rank = 0                                       # dry run: all ranks share the one visible GPU
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    import torch.distributed as dist
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.dist_backend == "gloo":
            dist.init_process_group(backend="gloo")
        else:
            dist.init_process_group(backend="nccl", device_id=device)    # backend "nccl" is RCCL on ROCm

    from funasr_amd import _lib, dp, synth
    from funasr_amd.wav_frontend import WavFrontend

    lib = _lib.load()
    cfg = synth.PARAFORMER_LARGE
    trace("library loaded, building model")
    model, _ = build_model(cfg, rank, world, device)
    trace("model on device")
    shift, scale = synth.synthetic_cmvn(560)
    frontend = WavFrontend(cmvn=torch.stack([shift, scale]), lfr_m=7, lfr_n=6, dither=0.0, device=device)

    # ---- synthetic workload, resident in HBM before the clock starts
    B = args.batch
    n_samples = int(args.seconds * 16000)
    base = [synth.speech_like(n_samples, seed=1000 * rank + i) for i in range(min(B, 8))]
    clips = [base[i % len(base)].roll(137 * (i // len(base))) for i in range(B)]   # distinct but cheap to generate
    wav = torch.stack(clips).to(device)
    lens = [n_samples] * B
    N_PAD = 512

    def enqueue():
        feats, flens = frontend(wav, lens)
        return model.enqueue_features(feats, flens)

    def collect(pending):
        res = model.collect(pending)
        if world > 1:      # gather hypotheses on rank 0 (fixed-stride int32 ids + counts), the path's only exchange
            dp.gather_hypotheses(res["raw_ids"], N_PAD, dst=0, device=device)
        return res

    def step():
        return collect(enqueue())

    def run_steps(k):
        """k batches, software-pipelined like a serving loop: batch i+1 is enqueued (frontend .. fused arg-max) before
        batch i's ids are brought to the host, so host-side post-processing never leaves the GPU idle. Every batch is
        fully processed and collected inside the call."""
        res = None
        pending = enqueue()
        for _ in range(k - 1):
            nxt = enqueue()
            res = collect(pending)
            pending = nxt
        return collect(pending)

    def sync():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize()

    def max_over_ranks(x: float) -> float:
        tt = torch.tensor([x], dtype=torch.float64, device="cpu" if args.dist_backend == "gloo" else device)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        return float(tt.item())

    model.set_precision(args.precision)
    trace("workload resident in HBM")
    for i in range(args.warmup):
        res = step()
        torch.cuda.synchronize()
        trace(f"warmup step {i} done")
    sync()
    lib.pf_prof_reset()
    lib.pf_prof_enable(1)
    t0 = time.perf_counter()
    res = run_steps(args.steps)
    sync()
    dt = time.perf_counter() - t0
    lib.pf_prof_enable(0)
    trace(f"timed region done: {dt:.3f} s for {args.steps} steps")
    if world > 1:
        dt = max_over_ranks(dt)

    if rank != 0:
        if world > 1:
            dist.destroy_process_group()
        return

    audio_s = world * B * args.seconds * args.steps
    value = audio_s / dt
    # ---- roofline of the dominant kernel (f32 MFMA GEMM), from hipEvents recorded around every launch in the
    #      timed region on the launch stream
    kinds = {0: "gemm_f32_mfma", 1: "attention_f32", 2: "fsmn", 3: "layernorm", 4: "fbank", 5: "gemm_bf16x3"}
    prof = {}
    for k, name in kinds.items():
        ms, work, n = C.c_double(0), C.c_double(0), C.c_int64(0)
        lib.pf_prof_read(k, C.byref(ms), C.byref(work), C.byref(n))
        prof[name] = dict(ms_per_step=ms.value / args.steps, work_per_step=work.value / args.steps,
                          launches_per_step=n.value / args.steps)
    if args.precision == "f16x2":
        # dominant kernel: gemm_f16x2_kernel: 3 fp16 MFMA flops per algorithmic flop, ceiling = dense fp16 peak / 3
        gemm, kname, peak = prof["gemm_bf16x3"], "gemm_f16x2_kernel", PEAK_BF16_MFMA_TFLOPS / 3
    elif args.precision == "bf16x3":
        # dominant kernel: gemm_split3_kernel. `achieved` = algorithmic (fp32-equivalent) 2MNK flops per second; the
        # kernel executes 6 bf16 MFMA flops per algorithmic flop, so its ceiling is the dense bf16 peak / 6
        gemm, kname, peak = prof["gemm_bf16x3"], "gemm_split3_kernel", PEAK_BF16_MFMA_TFLOPS / SPLIT3_PRODUCTS
    else:
        gemm, kname, peak = prof["gemm_f32_mfma"], "gemm_f32_mfma_kernel", (PEAK_BF16_MFMA_TFLOPS if args.precision == "bf16" else PEAK_F32_MFMA_TFLOPS)
    ach = gemm["work_per_step"] / (gemm["ms_per_step"] * 1e-3) / 1e12 if gemm["ms_per_step"] > 0 else 0.0
    pmc = pmc_traffic(kname)
    roofline = dict(bound="mfma", kernel=kname, achieved=round(ach, 2), peak=round(peak, 1),
                    unit="TFLOP/s", frac=round(ach / peak, 4), traffic=pmc[0] if pmc else None,
                    traffic_unit="HBM bytes per launch (PMC)", traffic_source=pmc[1] if pmc else None,
                    flops_per_launch=gemm["work_per_step"] / max(gemm["launches_per_step"], 1),
                    avg_launch_ms=gemm["ms_per_step"] / max(gemm["launches_per_step"], 1),
                    launches_per_step=gemm["launches_per_step"])
    if args.precision == "f16x2":
        roofline.update(peak_note="dense fp16 MFMA peak 2500 TFLOP/s / 3 products per fp32-equivalent product",
                        executed_f16_tflops=round(ach * 3, 1), fp32_mfma_peak_for_comparison=PEAK_F32_MFMA_TFLOPS)
    if args.precision == "bf16x3":
        roofline.update(peak_note="dense bf16 MFMA peak 2500 TFLOP/s / 6 products per fp32-equivalent product",
                        executed_bf16_tflops=round(ach * SPLIT3_PRODUCTS, 1),
                        fp32_mfma_peak_for_comparison=PEAK_F32_MFMA_TFLOPS)
    kernels = {k: dict(ms_per_step=round(v["ms_per_step"], 3), launches=v["launches_per_step"]) for k, v in prof.items()}
    for nm in ("gemm_f32_mfma", "gemm_bf16x3"):
        if prof[nm]["ms_per_step"] > 0:
            kernels[nm]["tflops"] = round(prof[nm]["work_per_step"] / (prof[nm]["ms_per_step"] * 1e-3) / 1e12, 2)
    attn = prof["attention_f32"]
    if attn["ms_per_step"] > 0:
        kernels["attention_f32"]["tflops"] = round(attn["work_per_step"] / (attn["ms_per_step"] * 1e-3) / 1e12, 2)
    for nm in ("fsmn", "layernorm", "fbank"):
        if prof[nm]["ms_per_step"] > 0:
            kernels[nm]["GBps"] = round(prof[nm]["work_per_step"] / (prof[nm]["ms_per_step"] * 1e-3) / 1e9, 1)

    # ---- secondary measurements (N = 1 only) on the same batch: the all-fp32-MFMA mode and the bf16-operand throughput
    #      mode, each with its token agreement against the main result measured, not assumed
    def time_mode(mode):
        model.set_precision(mode)
        for _ in range(max(1, args.warmup)):
            r = step()
        torch.cuda.synchronize()
        lib.pf_prof_reset()
        lib.pf_prof_enable(1)
        t1 = time.perf_counter()
        r = run_steps(args.steps)
        torch.cuda.synchronize()
        dtm = time.perf_counter() - t1
        lib.pf_prof_enable(0)
        ms, work, n = C.c_double(0), C.c_double(0), C.c_int64(0)
        lib.pf_prof_read(0, C.byref(ms), C.byref(work), C.byref(n))
        from funasr_amd.metrics import micro_error_rate
        ter, _, _ = micro_error_rate(res["raw_ids"], r["raw_ids"])
        same = sum(1 for a, b in zip(res["raw_ids"], r["raw_ids"]) if a == b)
        same_n = sum(1 for a, b in zip(res["token_num"], r["token_num"]) if a == b)
        tok = sum(len(a) for a in res["raw_ids"])
        diff = sum(sum(1 for x, y in zip(a, b) if x != y) + abs(len(a) - len(b)) for a, b in zip(res["raw_ids"], r["raw_ids"]))
        return {"value": round(B * args.seconds * args.steps / dtm, 1), "unit": "audio-s/s",
                "ms_per_step": round(dtm / args.steps * 1e3, 2),
                "gemm_f32_mfma_kernel_tflops": round(work.value / (ms.value * 1e-3) / 1e12, 1) if ms.value > 0 else None,
                "clips_with_identical_token_ids_vs_main": f"{same}/{B}",
                "clips_with_identical_token_count_vs_main": f"{same_n}/{B}",
                "token_positions_differing": f"{diff}/{tok}",
                "token_error_rate_vs_main": round(ter, 4)}

    bf16_mode = fp32_mfma_mode = None
    if world == 1 and not args.no_bf16 and args.precision in ("bf16x3", "f16x2"):
        try:
            fp32_mfma_mode = time_mode("fp32")
            fp32_mfma_mode["dtype"] = "f32, every GEMM on v_mfma_f32_32x32x2_f32 (157.3 TFLOP/s peak)"
            trace(f"fp32-MFMA mode: {fp32_mfma_mode['value']} audio-s/s")
            bf16_mode = time_mode("bf16")
            bf16_mode["dtype"] = ("bf16 operands (encoder + decoder GEMMs and attention), fp32 accumulate/residual/LN/softmax/FSMN; "
                                  "CIF predictor fp32; bf16-class error")
            trace(f"bf16-operand mode: {bf16_mode['value']} audio-s/s")
        finally:
            model.set_precision(args.precision)

    cpu_baseline = None
    if world == 1 and not args.no_cpu_baseline:
        trace("cpu baseline (oracle on host cores) ...")
        cpu_baseline = run_cpu_baseline(cfg, clips, shift, scale, res, args)
        trace("cpu baseline done")

    line = {
        "metric": "audio-seconds/sec (RTF^-1) Paraformer-large 30s@bs64", "value": round(value, 1),
        "unit": "audio-s/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(dt / args.steps * 1e3, 2), "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None,
        "dtype": {"fp32": "f32", "bf16": "bf16 operands / f32 accumulate",
                  "f16x2": "f32 (GEMM / attention operands split into 2 fp16 planes, 3 fp16 MFMA products, f32 accumulate; all else f32)",
                  "bf16x3": "f32 (GEMM operands split into 3 bf16 planes, 6 bf16 MFMA products, f32 accumulate; all else f32)"}[args.precision],
        "data": "synthetic",
        "config": {"workload": f"Paraformer-large (50 enc + 16 dec blocks, vocab 8404, random-init), "
                               f"{B} x {args.seconds:g} s 16 kHz clips per GPU, wav in HBM -> token ids on host",
                   "clips_per_gpu": B, "clip_seconds": args.seconds, "parallelism": f"utterance-dp{world}",
                   "tokens_per_clip": round(sum(res["token_num"]) / len(res["token_num"]), 1)},
        "roofline": roofline, "cpu_baseline": cpu_baseline, "kernels": kernels, "fp32_mfma_mode": fp32_mfma_mode,
        "bf16_mode": bf16_mode,
    }
    print(json.dumps(line), flush=True)
    if world > 1:
        dist.destroy_process_group()


def pmc_traffic(kernel: str):
    """HBM bytes per launch of `kernel` from the newest committed PMC summary (profiles/*_traffic.json, written by
    tools/summarize_prof.py from separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this same command, with
    the guide's gfx950 corrections). null when no summary is present: PMC passes are not run inside the timed bench."""
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "*_traffic.json")))
    if not files:
        return None
    try:
        with open(files[-1]) as f:
            t = json.load(f)
        rows = [v for k, v in t["kernels"].items() if k.startswith(kernel)]
        n = sum(r["launches"] for r in rows)
        if n == 0:
            return None
        b = sum((r["read_bytes_per_launch"] + r["write_bytes_per_launch"]) * r["launches"] for r in rows) / n
        return round(b), os.path.basename(files[-1])
    except (OSError, KeyError, ValueError):
        return None


def host_cores() -> int:
    """CPU threads this process may really use: min(affinity mask, cgroup quota). os.cpu_count() reports the
    machine's cores even inside a CPU-limited container, and an oversubscribed OpenMP pool is ~100x slower."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        with open("/sys/fs/cgroup/cpu.max") as f:
            quota, period = f.read().split()[:2]
        if quota != "max":
            n = min(n, max(1, int(float(quota) / float(period))))
    except (OSError, ValueError):
        pass
    return max(1, n)


def run_cpu_baseline(cfg, clips, shift, scale, gpu_res, args):
    """The oracle (CPU port of the reference path: the same ATen CPU kernels the reference's nn.Modules call) timed on
    the host cores on a BOUNDED sample of the same workload, batch_size 1 like AutoModel on device="cpu"
    (funasr/auto/auto_model.py:551-561). A 3 s probe clip calibrates the host; the sample is then the first
    `--cpu-clips` clips, shortened if the host is too slow for ~25 s of work. Full-length clips double as an
    end-of-run token-id parity check against the GPU result."""
    from funasr_amd import synth
    from oracle import paraformer_oracle as O

    cores = min(host_cores(), args.cpu_threads) if args.cpu_threads > 0 else host_cores()
    torch.set_num_threads(cores)
    sd = synth.paraformer_state_dict(cfg, seed=0, cif_bias=synth.BENCH_CIF_BIAS)
    cmvn = torch.stack([shift, scale])

    def run(w):
        feats, flens = O.wav_frontend([w], cmvn)
        return O.paraformer_greedy(feats, flens, sd, cfg)

    with torch.no_grad():
        t0 = time.perf_counter()
        run(clips[0][: 3 * 16000])
        probe = time.perf_counter() - t0
        rate = 3.0 / probe                                   # audio-s per wall-s on a short clip (optimistic for 30 s)
        trace(f"cpu probe: 3 s clip in {probe:.2f} s on {cores} threads")
        budget_s = 20.0
        full = clips[0].numel() / 16000.0
        n_full = int(min(len(clips), args.cpu_clips)) if (budget_s * rate) // full >= 1 else 0   # loop stops at budget_s
        match = None
        if n_full >= 1:
            sample = clips[:n_full]
        else:                                                # host too slow for one whole clip inside the budget
            sample = [clips[0][: max(16000, int(budget_s * rate * 0.7) * 16000)]]
        t0 = time.perf_counter()
        done = 0
        cpu_ids = []
        for i, w in enumerate(sample):
            r = run(w)
            done += 1
            if n_full >= 1:
                cpu_ids.append(r["raw_ids"][0])
                ok = r["raw_ids"][0] == gpu_res["raw_ids"][i]
                match = ok if match is None else (match and ok)
            if time.perf_counter() - t0 > budget_s:
                break
        sample = sample[:done]
        dt = time.perf_counter() - t0
    secs = sum(w.numel() for w in sample) / 16000.0
    ter_cpu = None
    if cpu_ids:                      # the metric's "CER vs CPU ref" on what can be compared here: token ids, micro-averaged
        from funasr_amd.metrics import micro_error_rate
        ter_cpu = round(micro_error_rate(cpu_ids, gpu_res["raw_ids"][: len(cpu_ids)])[0], 6)
    return {"value": round(secs / dt, 2), "unit": "audio-s/s", "cores": cores, "kind": "port",
            "sample": f"{len(sample)} x {secs / len(sample):g} s clip(s) of the same batch, batch_size 1, fp32, "
                      f"torch {torch.__version__} CPU ATen kernels, {cores} threads, {dt:.1f} s of CPU work",
            "token_ids_match_gpu": match, "token_error_rate_gpu_vs_cpu_ref": ter_cpu}


if __name__ == "__main__":
    main()
