#!/usr/bin/env python3
"""Corpus sweep (BASELINE configs[3], scaled): decode N clips with AISHELL-1-like durations (mean 5.0 s, 1.9-14.7 s;
the 7176 test durations sum to 36 108.9 s, runtime/triton_gpu/client/aishell_test.txt) through the full path, batched after a
length sort by a budget of padded encoder ROWS (default; sized to whole rounds of GEMM blocks on the 256 CUs) or of padded
seconds (`--batch-seconds`, the reference's `batch_size_s` policy, funasr/auto/auto_model.py:893-955) and, under
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
    ap.add_argument("--gpus", type=int, default=0, help="ranks (one per GPU); > 1 without a launcher re-executes the script under "
                    "torch.distributed.run (funasr_amd.dp.ensure_ranks); 0 = whatever the launcher started")
    ap.add_argument("--clips", type=int, default=2000)
    ap.add_argument("--batch-seconds", type=float, default=0.0, help="padded audio seconds per batch (the reference's batch_size_s "
                    "policy); 0 = use --batch-rows")
    ap.add_argument("--batch-rows", type=int, default=32768, help="encoder rows per batch (the rows the encoder really computes: "
                    "16-row slots of frames + 1 in the f16x2 mode, B * ceil16(longest) otherwise). 32768 rows = 128 row tiles "
                    "of 256 = exactly one round of 256 x 256 blocks for the N = 512 GEMMs on 256 CUs (64 x 30 s clips is the "
                    "same amount of work)")
    ap.add_argument("--model", default="paraformer", choices=["paraformer", "sensevoice"])
    ap.add_argument("--precision", default="f16x2", choices=["fp32", "bf16", "bf16x3", "f16x2"])
    ap.add_argument("--dump", default=None, help="rank 0 writes the hypotheses in CORPUS order to this JSON file")
    ap.add_argument("--no-pack", action="store_true", help="A/B: plan batches for the padded layout (no encoder row packing)")
    ap.add_argument("--no-overlap", action="store_true", help="assemble, decode and collect one batch at a time")
    ap.add_argument("--no-decoder-stream", action="store_true", help="Paraformer: whole batches enqueued one after the other on one stream (round 5's loop) "
                    "instead of the two-phase loop with batch i's decoder on a second stream beside batch i+1's encoder")
    ap.add_argument("--dist-backend", default="nccl")
    ap.add_argument("--verbose", action="store_true")
    ap.add_argument("--serialize-gpu", action="store_true", help="single-GPU dry run of the N > 1 path (every rank on cuda:0): the "
                    "ranks take turns on the GPU instead of time-sharing it. Two processes interleaving kernels on ONE GPU is not a "
                    "deployment configuration (one process per GPU is), and the frontend was seen to return sporadically different "
                    "features there (DESIGN 6, known issue); the dry run is about rendezvous, broadcast, sharding, gather and order")
    ap.add_argument("--trace-hash", default=None, help="debug: write per-clip bit hashes of the features, encoder output and CIF "
                    "weights to this JSON file (rank r appends .r)")
    ap.add_argument("--enc-option", action="append", default=[], metavar="KEY=VALUE", help="encoder schedule options (pf_encoder_set_option)")
    args = ap.parse_args()
    rank, world = int(os.environ.get("RANK", 0)), int(os.environ.get("WORLD_SIZE", 1))
    if args.gpus:
        from funasr_amd.dp import ensure_ranks
        world = ensure_ranks(args.gpus)
    local = 0 if args.dist_backend == "gloo" else int(os.environ.get("LOCAL_RANK", 0))
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    import torch.distributed as dist
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group(backend=args.dist_backend, **({} if args.dist_backend == "gloo" else {"device_id": dev}))
    from funasr_amd import dp, synth
    from funasr_amd.wav_frontend import WavFrontend
    dp.guard_shared_gpu(world, all_on_one=args.dist_backend == "gloo")

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
    model.set_precision(args.precision) if hasattr(model, "set_precision") else model.encoder.set_precision(args.precision)
    for kv in args.enc_option:
        key, _, val = kv.partition("=")
        model.encoder.set_option(key, int(val))
    sh, sc = synth.synthetic_cmvn(560)
    fe = WavFrontend(cmvn=torch.stack([sh, sc]), lfr_m=7, lfr_n=6, dither=0.0, device=dev)

    durs = durations(args.clips)
    lens = [int(d * 16000) for d in durs]
    mine = dp.shard_indices(lens, world, rank)            # length-sorted round-robin deal, already descending
    # a small pool of base clips, cut / rolled per utterance (cheap to generate, distinct content)
    pool = [synth.speech_like(int(14.7 * 16000) + 1, seed=1000 + i) for i in range(16)]
    clips = {i: pool[i % 16].roll(31 * i)[: lens[i]] for i in mine}
    # batches by encoder-rows (or padded-seconds) budget. Rows the encoder computes for one clip: its frames (+ SenseVoice's 4
    # query frames) plus the one padding row the CIF predictor reads, in a 16-row slot (pf_encoder_set_row_packing)
    q = 4 if args.model == "sensevoice" else 0
    extra = 0 if args.model == "sensevoice" else 1
    packed = args.precision == "f16x2" and not args.no_pack
    if args.batch_seconds > 0:
        batches, cur = [], []
        for i in mine:
            longest = lens[cur[0]] if cur else lens[i]
            if cur and (len(cur) + 1) * longest / 16000.0 > args.batch_seconds:
                batches.append(cur)
                cur = []
            cur.append(i)
        if cur:
            batches.append(cur)
    else:
        plan = dp.plan_batches_by_rows([fe.num_frames(lens[i]) + q for i in mine], args.batch_rows, extra_rows=extra, packed=packed)
        batches = [list(mine[b:e]) for b, e in plan]

    # The loaded corpus lives in one pinned host arena (what a data-loader worker hands over); a batch is built ON THE
    # DEVICE: a zeroed [B, n_max] tensor (pad_sequence semantics, load_utils.py:413) receives one asynchronous H2D copy
    # per clip, so the host never touches the samples again. Batches are software-pipelined with enqueue_features /
    # collect: batch i+1 is staged and enqueued before batch i's ids are brought to the host.
    arena = torch.empty(sum(lens[i] for i in mine)).pin_memory()
    off = 0
    for i in mine:
        arena[off: off + lens[i]] = clips[i]
        clips[i] = arena[off: off + lens[i]]
        off += lens[i]
    timing = {"stage": 0.0, "enqueue": 0.0, "collect": 0.0}
    trace_rows = []
    gpu_events = []
    paraformer = hasattr(model, "enqueue_features")
    # the bench's loop: begin(i+1) -> finish(i) on a second stream -> collect(i-1); ids are those of enqueue_features
    two_phase = (paraformer and hasattr(model, "begin_features") and model._one_call_ok() and not args.no_overlap and not args.no_decoder_stream
                 and args.trace_hash is None)
    dec_stream = torch.cuda.Stream(device=dev) if two_phase else None
    if two_phase:
        from funasr_amd import _lib
        _lib.load().pf_set_concurrency_guard(1)

    def launch(batch):
        t = time.perf_counter()
        L = [lens[i] for i in batch]
        wav = torch.zeros(len(batch), max(L), device=dev)
        for j, i in enumerate(batch):
            wav[j, : L[j]].copy_(clips[i], non_blocking=True)
        timing["stage"] += time.perf_counter() - t
        t = time.perf_counter()
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        ev0.record()
        feats, flens = fe(wav, L)
        if args.trace_hash is not None and args.model == "paraformer":
            hb = lambda t: int(t.contiguous().view(torch.int32).to(torch.int64).sum().item())
            r = model.recognize_features(feats, flens, return_intermediate=True)
            trace_rows.append({"clips": list(batch), "wav": hb(wav), "feats": hb(feats), "enc": hb(r["enc"]), "alphas": hb(r["alphas"]),
                               "embeds": hb(r["embeds"]), "tok": r["token_num"], "ids": r["raw_ids"]})
        if two_phase:
            pending = model.begin_features(feats, flens)
        elif paraformer:
            pending = model.enqueue_features(feats, flens)
        elif args.model == "sensevoice":
            pending = model.recognize_features(feats, flens, "auto", "woitn")
        else:
            pending = model.recognize_features(feats, flens)
        ev1.record()
        gpu_events.append((ev0, ev1))
        timing["enqueue"] += time.perf_counter() - t
        return pending

    def finish(pending):
        t = time.perf_counter()
        if two_phase and "ticket" in pending:
            pending = model.finish_features(pending, stream=dec_stream)
        ids = model.collect(pending)["ids"] if paraformer else pending["ids"]
        timing["collect"] += time.perf_counter() - t
        return ids

    # warm-up: the longest-clip batch and the one with the most clips, so that every grow-only buffer (device workspaces,
    # pinned id buffers) has its final size before the clock starts
    finish(launch(batches[0]))
    finish(launch(max(batches, key=len)))
    torch.cuda.synchronize()
    for k in timing:
        timing[k] = 0.0
    gpu_events.clear()
    if world > 1:
        dist.barrier()
    t0 = time.perf_counter()
    hyps = {}

    def decode_all_two_phase():
        ticket = pend = None                      # (batch, ticket of begin) / (batch, pending of finish)
        for b in batches:
            nxt = launch(b)
            if ticket is not None:
                t = time.perf_counter()
                fin = model.finish_features(ticket[1], stream=dec_stream)
                timing["enqueue"] += time.perf_counter() - t
                if pend is not None:
                    for i, ids in zip(pend[0], finish(pend[1])):
                        hyps[i] = ids
                pend = (ticket[0], fin)
            ticket = (b, nxt)
        if ticket is not None:
            fin = model.finish_features(ticket[1], stream=dec_stream)
            if pend is not None:
                for i, ids in zip(pend[0], finish(pend[1])):
                    hyps[i] = ids
            for i, ids in zip(ticket[0], finish(fin)):
                hyps[i] = ids
        torch.cuda.synchronize()

    def decode_all():
        if two_phase:
            return decode_all_two_phase()
        inflight = None
        for b in batches:
            pending = launch(b)
            if args.no_overlap:
                for i, ids in zip(b, finish(pending)):
                    hyps[i] = ids
                continue
            if inflight is not None:
                for i, ids in zip(inflight[0], finish(inflight[1])):
                    hyps[i] = ids
            inflight = (b, pending)
        if inflight is not None:
            for i, ids in zip(inflight[0], finish(inflight[1])):
                hyps[i] = ids
        torch.cuda.synchronize()

    if args.serialize_gpu and world > 1:
        for turn in range(world):                 # the ranks take turns on the one GPU of the dry run
            if turn == rank:
                decode_all()
            dist.barrier()
    else:
        decode_all()
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
    if rank == 0 and args.dump:
        if world > 1:                                          # rank r decoded the clips dp.shard_indices(lens, world, r), in that order
            corpus = [None] * args.clips
            for r in range(world):
                for k, i in enumerate(dp.shard_indices(lens, world, r)):
                    corpus[i] = gathered[r][k]
        else:
            corpus = [hyps[i] for i in range(args.clips)]
        with open(args.dump, "w") as f:
            json.dump(corpus, f)
    if rank == 0:
        total_s = sum(durs)
        padded = sum(len(b) * max(lens[i] for i in b) for b in batches) / 16000.0
        mine_s = sum(lens[i] for i in mine) / 16000.0
        print(json.dumps({"metric": f"corpus sweep audio-seconds/s ({args.model}, {args.precision})", "value": round(total_s / dt, 1),
                          "loop": ("two-phase, decoder on a second stream" if two_phase else "one batch at a time" if args.no_overlap else "whole batches, one stream"),
                          "n_gpus": world, "clips": args.clips, "audio_hours": round(total_s / 3600, 2), "wall_s": round(dt, 3),
                          "batches_rank0": len(batches), "batch_budget": (f"{args.batch_seconds} s" if args.batch_seconds > 0 else f"{args.batch_rows} rows"),
                          "padding_efficiency_rank0": round(mine_s / padded, 3),
                          "tokens_rank0": sum(len(v) for v in hyps.values()),
                          "host_seconds_rank0": {k: round(v, 3) for k, v in timing.items()},
                          # GPU time between the first and the last kernel of every batch (events on the launch stream): when it
                          # adds up to the wall time the sweep is GPU-bound and `enqueue` is back-pressure, not host work
                          "gpu_seconds_rank0": round(sum(a.elapsed_time(b) for a, b in gpu_events) * 1e-3, 3)}), flush=True)
    print(f"[sweep] rank {rank}: frontend cross-check {'on' if fe.verify else 'off'}, disagreements seen: {fe.faults()} log(cycles a, cycles b, frame, retry)={fe.fault_log()}", file=sys.stderr, flush=True)
    if args.trace_hash is not None:
        with open(f"{args.trace_hash}.{rank}", "w") as f:
            json.dump(trace_rows, f)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
