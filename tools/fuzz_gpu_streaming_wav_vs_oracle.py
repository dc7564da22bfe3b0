#!/usr/bin/env python3
"""Wav-level streaming fuzz on the GPU: utterances of random length (0.05 s .. 12 s, incl. lengths that leave a last piece shorter than
960 samples -- the "tail chunk" -- and utterances shorter than one chunk) fed to ParaformerStreaming.inference in calls of arbitrary
sizes with a session cache (online frontend with its LFR splice cache and final flush, chunk loop, left-over samples, hipGraph step),
random chunk geometry / look-backs, against oracle/streaming_oracle.py streaming_inference on the same calls: token ids per call.
Not part of the test run. usage: fuzz_gpu_streaming_wav_vs_oracle.py [seed] [sessions]"""
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from funasr_amd import synth                                                          # noqa: E402
from funasr_amd.paraformer_streaming import ParaformerStreaming, WavFrontendOnline    # noqa: E402
from oracle import paraformer_oracle as O                                             # noqa: E402
from oracle import streaming_oracle as S                                              # noqa: E402

GOLD = os.path.join(ROOT, "tests", "golden")
dev = torch.device("cuda:0")
seed = int(sys.argv[1]) if len(sys.argv) > 1 else 0
n_sessions = int(sys.argv[2]) if len(sys.argv) > 2 else 20
gen = torch.Generator().manual_seed(seed)
g = np.load(os.path.join(GOLD, "streaming.npz"), allow_pickle=False)
cfg = json.loads(bytes(g["config"]).decode())
cmvn = O.load_cmvn(os.path.join(GOLD, "am.mvn"))


def ri(lo, hi):
    return int(torch.randint(lo, hi + 1, (1,), generator=gen))


bad, calls, tokens = 0, 0, 0
for si in range(n_sessions):
    sd = synth.paraformer_state_dict(cfg, seed=400 + si, cif_bias=float(g["cif_bias"]) + (float(torch.rand(1, generator=gen)) - 0.5) * 0.6)
    model = ParaformerStreaming.from_config(cfg)
    model.load_state_dict(sd, strict=False)
    model = model.to(dev)
    fe = WavFrontendOnline(cmvn_file=os.path.join(GOLD, "am.mvn"), lfr_m=7, lfr_n=6, dither=0.0, device=dev)
    if si % 2 == 0:
        chunk, enc_lb, dec_lb = [0, 10, 5], 4, 1                   # the published geometry
    else:
        cur = ri(4, 14)
        chunk, enc_lb, dec_lb = [ri(0, 1) * ri(1, 5), cur, ri(1, max(1, cur // 2))], ri(0, 4), ri(0, 2)
    stride = chunk[1] * 960
    kind = si % 5
    if kind == 0:
        n = ri(800, stride - 1)                                    # shorter than one chunk
    elif kind == 1:
        n = ri(1, 6) * stride + ri(1, 959)                         # the last piece is a tail chunk (< 960 samples)
    else:
        n = ri(stride, 12 * 16000)
    wav = synth.speech_like(n, seed=5000 + 100 * seed + si)
    cuts = sorted({ri(1, n - 1) for _ in range(ri(0, 4))}) if n > 2 else []
    pieces = [wav[a:b] for a, b in zip([0] + cuts, cuts + [n])]
    kw = dict(chunk_size=chunk, encoder_chunk_look_back=enc_lb, decoder_chunk_look_back=dec_lb)
    st = S.model_init(cfg, tuple(chunk), enc_lb, dec_lb)
    cache, ok = {}, True
    for pi, piece in enumerate(pieces):
        fin = pi == len(pieces) - 1
        with torch.no_grad():
            want = S.streaming_inference(piece, st, sd, cfg, cmvn, fin)
        try:
            res, _ = model.inference([piece], key=["utt"], tokenizer=None, frontend=fe, cache=cache, is_final=fin, **kw)
            got = res[0]["token_int"] if res else []
        except Exception as e:                                     # noqa: BLE001
            got = f"{type(e).__name__}: {str(e)[:120]}"
        calls += 1
        tokens += len(want)
        if got != want:
            ok = False
            print(f"session {si} call {pi}/{len(pieces)} ({len(piece)} samples, final={fin}) geometry={chunk} lb=({enc_lb},{dec_lb}): got {got} want {want}")
    bad += 0 if ok else 1
    print(f"session {si}: {n} samples in {len(pieces)} call(s), geometry={chunk} lb=({enc_lb},{dec_lb}) -> {'ok' if ok else 'MISMATCH'}")
print(json.dumps(dict(tool="fuzz_gpu_streaming_wav_vs_oracle", seed=seed, sessions=n_sessions, calls=calls, tokens=tokens, bad_sessions=bad)))
sys.exit(1 if bad else 0)
