#!/usr/bin/env python3
"""Corpus sweep (BASELINE configs[3], scaled): decode N clips with AISHELL-1-like durations (mean 5.0 s, 1.9-14.7 s;
the 7176 test durations sum to 36 108.9 s, runtime/triton_gpu/client/aishell_test.txt) through the full path, batched by a
seconds budget after a length sort (the reference's `batch_size_s` policy, funasr/auto/auto_model.py:893-955) and, under
torchrun, sharded over the ranks with funasr_amd.dp (weights broadcast as one arena, hypotheses gathered on rank 0).

  python tools/sweep.py --clips 2000                       # 1 GPU
  python -m torch.distributed.run --nproc-per-node 8 tools/sweep.py --clips 10000
Prints one JSON line on rank 0: audio-seconds/s over the whole sweep (wall clock incl. H2D of the waveforms), padding
efficiency, number of batches. `--model sensevoice` runs SenseVoiceSmall (encoder + CTC greedy, configs[2] shapes)."""
import argparse
import json
import math
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def durations(n, seed=0):
    g = torch.Generator().manual_seed(seed)
    mu, sigma = math.log(4.75), 0.38                     # lognormal: mean ~5.1 s before clipping
    d = torch.exp(mu + sigma * torch.randn(n, generator=g)).clamp(1.86, 14.70)
    return d.tolist()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--clips", type=int, default=2000)
    ap.add_argument("--batch-seconds", type=float, default=1920.0, help="padded audio seconds per batch (64 x 30 s)")
    ap.add_argument("--model", default="paraformer", choices=["paraformer", "sensevoice"])
    ap.add_argument("--precision", default="fp32", choices=["fp32", "bf16"])
    ap.add_argument("--dist-backend", default="nccl")
    ap.add_argument("--verbose", action="store_true")
    args = ap.parse_args()
    rank, world = int(os.environ.get("RANK", 0)), int(os.environ.get("WORLD_SIZE", 1))
    local = 0 if args.dist_backend == "gloo" else int(os.environ.get("LOCAL_RANK", 0))
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    import torch.distributed as dist
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group(backend=args.dist_backend, **({} if args.dist_backend == "gloo" else {"device_id": dev}))
    from funasr_amd import dp, synth
    from funasr_amd.wav_frontend import WavFrontend

    if args.model == "paraformer":
        from funasr_amd.paraformer import Paraformer
        cfg = synth.PARAFORMER_LARGE
        model = Paraformer.from_config(cfg)
        if rank == 0:
            model.load_state_dict(synth.paraformer_state_dict(cfg, seed=0, cif_bias=synth.BENCH_CIF_BIAS), strict=False)
    else:
        from funasr_amd.sense_voice import SenseVoiceSmall
        cfg = synth.SENSEVOICE_SMALL
        model = SenseVoiceSmall.from_config(cfg)
        if rank == 0:
            model.load_state_dict(synth.sensevoice_state_dict(cfg, seed=0), strict=False)
    model = model.to(dev)
    if world > 1:
        dp.broadcast_model(model)
    model.encoder.set_precision(args.precision)
    sh, sc = synth.synthetic_cmvn(560)
    fe = WavFrontend(cmvn=torch.stack([sh, sc]), lfr_m=7, lfr_n=6, dither=0.0, device=dev)

    durs = durations(args.clips)
    lens = [int(d * 16000) for d in durs]
    mine = dp.shard_indices(lens, world, rank)            # length-sorted round-robin deal, already descending
    # a small pool of base clips, cut / rolled per utterance (cheap to generate, distinct content)
    pool = [synth.speech_like(int(14.7 * 16000) + 1, seed=1000 + i) for i in range(16)]
    clips = {i: pool[i % 16].roll(31 * i)[: lens[i]] for i in mine}
    # batches by padded-seconds budget
    batches, cur = [], []
    for i in mine:
        longest = lens[cur[0]] if cur else lens[i]
        if cur and (len(cur) + 1) * longest / 16000.0 > args.batch_seconds:
            batches.append(cur)
            cur = []
        cur.append(i)
    if cur:
        batches.append(cur)

    pinned = torch.empty(int(args.batch_seconds * 16000) + 16 * 240000).pin_memory()   # one staging buffer, reused

    def decode(batch):
        """host side: zero-padded [B, n_max] float32 (pad_sequence, load_utils.py:413) staged in pinned memory"""
        L = [lens[i] for i in batch]
        wav = pinned[: len(batch) * max(L)].view(len(batch), max(L))
        wav.zero_()
        for k, i in enumerate(batch):
            wav[k, : L[k]] = clips[i]
        feats, flens = fe(wav.to(dev, non_blocking=True), L)
        res = model.recognize_features(feats, flens)
        return res["ids"]

    decode(batches[0])                                     # warm-up (allocations, weight push)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    t0 = time.perf_counter()
    hyps = {}
    for b in batches:
        tb = time.perf_counter()
        for i, ids in zip(b, decode(b)):
            hyps[i] = ids
        if args.verbose and rank == 0:
            torch.cuda.synchronize()
            print(f"batch of {len(b)} clips, longest {max(lens[i] for i in b) / 16000:.1f} s: {(time.perf_counter() - tb) * 1e3:.1f} ms",
                  file=sys.stderr, flush=True)
    torch.cuda.synchronize()
    if world > 1:
        order = list(mine)
        width = max(len(dp.shard_indices(lens, world, r)) for r in range(world))
        mylist = [hyps[i] for i in order] + [[]] * (width - len(order))
        gathered = dp.gather_hypotheses(mylist, 256, dst=0, device=dev)
        dist.barrier()
    dt = time.perf_counter() - t0
    if world > 1:
        tt = torch.tensor([dt], dtype=torch.float64, device="cpu" if args.dist_backend == "gloo" else dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())
    if rank == 0:
        total_s = sum(durs)
        padded = sum(len(b) * max(lens[i] for i in b) for b in batches) / 16000.0
        mine_s = sum(lens[i] for i in mine) / 16000.0
        print(json.dumps({"metric": f"corpus sweep audio-seconds/s ({args.model}, {args.precision})", "value": round(total_s / dt, 1),
                          "n_gpus": world, "clips": args.clips, "audio_hours": round(total_s / 3600, 2), "wall_s": round(dt, 3),
                          "batches_rank0": len(batches), "padding_efficiency_rank0": round(mine_s / padded, 3),
                          "tokens_rank0": sum(len(v) for v in hyps.values())}), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
