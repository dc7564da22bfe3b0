#!/usr/bin/env python3
"""Streaming fuzz on the GPU: random chunk geometries ([left, cur, right]), encoder / decoder look-backs, session lengths, final-chunk
sizes and both step arithmetics through StreamBatch (the hipGraph-captured step) against the reference-pinned streaming oracle
(oracle/streaming_oracle.py) on random online features: token ids and counts per chunk, encoder window within 1e-3.
Not part of the test run. usage: fuzz_gpu_streaming_vs_oracle.py [seed] [sessions]"""
import json
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from funasr_amd import synth                                             # noqa: E402
from funasr_amd.paraformer_streaming import ParaformerStreaming, StreamBatch   # noqa: E402
from oracle import streaming_oracle as S                                 # noqa: E402

GOLD = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")
dev = torch.device("cuda:0")
seed = int(sys.argv[1]) if len(sys.argv) > 1 else 0
n_sessions = int(sys.argv[2]) if len(sys.argv) > 2 else 20
gen = torch.Generator().manual_seed(seed)
g = np.load(os.path.join(GOLD, "streaming.npz"), allow_pickle=False)
cfg = json.loads(bytes(g["config"]).decode())


def ri(lo, hi):
    return int(torch.randint(lo, hi + 1, (1,), generator=gen))


bad, worst, steps, skipped = 0, 0.0, 0, 0
for si in range(n_sessions):
    sd = synth.paraformer_state_dict(cfg, seed=300 + si, cif_bias=float(g["cif_bias"]) + (float(torch.rand(1, generator=gen)) - 0.5) * 0.6)
    model = ParaformerStreaming.from_config(cfg)
    model.load_state_dict(sd, strict=False)
    model = model.to(dev)
    cur = ri(4, 20)
    chunk = [ri(0, 1) * ri(1, 6), cur, ri(1, max(1, cur // 2))]
    enc_lb, dec_lb = ri(0, 4), ri(0, 2)
    precision = ("fp32", "f16x2")[ri(0, 1)]
    n_str = 1 if si % 2 == 0 else ri(2, 5)           # lock-step streams: own features and oracle state each, same chunk sizes
    try:
        sb = StreamBatch(model, n_str, chunk, enc_lb, dec_lb, precision=precision)
    except (ValueError, RuntimeError) as e:          # geometries the step refuses loudly (token rows per step capped, ...)
        skipped += 1
        print(f"session {si}: chunk={chunk} lb=({enc_lb},{dec_lb}) refused: {str(e)[:100]}")
        continue
    sts = [S.model_init(cfg, tuple(chunk), enc_lb, dec_lb) for _ in range(n_str)]
    quiet = [n_str > 1 and ri(0, 2) == 0 for _ in range(n_str)]      # features near zero: such a stream fires (almost) nothing in a step
    n_chunks = ri(2, 9)
    ragged = si % 3 == 2          # API-level sessions: non-final chunks of any size (the reference's own loop always brings chunk_cur frames)
    ok = True
    for i in range(n_chunks):
        fin = i == n_chunks - 1
        n = (ri(1, cur) if ragged else cur) if not fin else ri(1, cur + 2)
        feats = torch.randn(n_str, n, 560, generator=gen) * 0.7
        for k in range(n_str):
            if quiet[k]:
                feats[k] *= 0.02
        ids, enc = sb.step(feats.to(dev), is_final=fin, return_enc=True)
        steps += 1
        for k in range(n_str):
            trace = []
            with torch.no_grad():
                oids = S.generate_chunk(feats[k:k + 1].clone(), sts[k], sd, cfg, fin, trace)
            d = (enc[k:k + 1].cpu() - trace[0]["enc"]).abs().max().item()
            worst = max(worst, d)
            same = [t for t in ids[k] if t not in (0, 1, 2)] == oids and len(ids[k]) == trace[0]["n"] and d < 1e-3
            if not same:
                ok = False
                print(f"   enc window |d| {d:.3e}, counts {len(ids[k])} / {trace[0]['n']}")
                print(f"session {si} chunk {i} stream {k}/{n_str}: geometry={chunk} lb=({enc_lb},{dec_lb}) {precision} fin={fin} n={n}: ids {ids[k]} oracle {oids}")
    bad += 0 if ok else 1
    print(f"session {si}: chunk={chunk} lb=({enc_lb},{dec_lb}) {precision} streams={n_str} chunks={n_chunks}{' ragged' if ragged else ''} -> {'ok' if ok else 'MISMATCH'}")
    sb.close()
print(json.dumps(dict(tool="fuzz_gpu_streaming_vs_oracle", seed=seed, sessions=n_sessions, refused=skipped, steps=steps, bad_sessions=bad,
                      worst_encoder_window_abs_diff=worst)))
sys.exit(1 if bad else 0)
