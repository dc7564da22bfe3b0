#!/usr/bin/env python3
"""Frontend fuzz on the GPU: random ragged batches (1 .. 50 000 samples per clip, down to fewer samples than one window with
snip_edges=False), every window type, snip_edges either way, LFR on and off, clips shorter than one 25-ms window (per-clip window, wav_frontend.py:176), against the CPU oracle's kaldi_fbank / apply_lfr / apply_cmvn
(itself pinned to the reference-vendored kaldi-native-fbank in tests/test_oracle.py). Not part of the test run.
usage: fuzz_gpu_frontend_options.py [seed] [cases]"""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from funasr_amd import synth                           # noqa: E402
from funasr_amd.wav_frontend import WINDOW_TYPES, WavFrontend   # noqa: E402
from oracle import paraformer_oracle as O              # noqa: E402

dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(int(sys.argv[1]) if len(sys.argv) > 1 else 0)
n_cases = int(sys.argv[2]) if len(sys.argv) > 2 else 40
sh, sc = synth.synthetic_cmvn()
worst, bad = 0.0, 0
for ci in range(n_cases):
    window = WINDOW_TYPES[int(torch.randint(0, len(WINDOW_TYPES), (1,), generator=g))]
    snip = bool(torch.randint(0, 2, (1,), generator=g))
    lfr = bool(torch.randint(0, 2, (1,), generator=g))
    B = int(torch.randint(1, 7, (1,), generator=g))
    lo = 2 if snip else 81                              # snip_edges=False: at least one frame needs (n + 80) // 160 >= 1
    lens = [int(torch.randint(lo, 50001 if ci % 3 else 1200, (1,), generator=g)) for _ in range(B)]
    waves = [synth.speech_like(n, seed=1000 * ci + i) for i, n in enumerate(lens)]
    batch = torch.zeros(B, max(lens))
    for i, w in enumerate(waves):
        batch[i, : lens[i]] = w
    fe = WavFrontend(cmvn=torch.stack([sh, sc]) if lfr else None, lfr_m=7 if lfr else 1, lfr_n=6 if lfr else 1, dither=0.0,
                     window=window, snip_edges=snip)
    feats, flens, fb = fe(batch.to(dev), lens, return_fbank=True)
    feats, fb = feats.cpu(), fb.cpu()
    ok = True
    for i, w in enumerate(waves):
        # clips shorter than 25 ms: one window of the clip's own length (wav_frontend.py:176), the FFT size following
        ofb = O.kaldi_fbank(w * 32768.0, 80, O.short_clip_frame_length_ms(lens[i]), window_type=window, snip_edges=snip)
        ok = ok and ofb.shape[0] == fe.num_fbank_frames(lens[i])
        # bar in the log domain: 4e-3 (float32 FFT against float32 FFT in another order) -- except for mel bins whose energy lies more
        # than four decades under the frame's strongest bin: there the FFT's absolute round-off (~1e-7 of the frame's energy) is a
        # visible FRACTION of the bin (windows with deep side lobes -- blackman, and rectangular's leakage -- produce such bins)
        e = torch.exp(ofb)
        weak = e < 1e-4 * e.max(dim=1, keepdim=True).values if ofb.shape[0] else None
        dd = (fb[i, : ofb.shape[0]] - ofb).abs()
        d = dd[~weak].max().item() if ofb.shape[0] and bool((~weak).any()) else 0.0
        if ofb.shape[0] and bool(weak.any()) and dd[weak].max().item() > 5e-2:
            ok = False
        of = O.apply_cmvn(O.apply_lfr(ofb, 7, 6), torch.stack([sh, sc])) if lfr else ofb
        ok = ok and int(flens[i]) == of.shape[0]
        d2 = (feats[i, : of.shape[0]] - of).abs().max().item()
        if ofb.shape[0] and bool(weak.any()):
            d2 = 0.0 if d2 <= 5e-2 else d2                          # (the stacked / normalised features hold the weak bins too)
        ok = ok and bool((feats[i, of.shape[0]:] == 0).all())
        worst = max(worst, d)
        # near-cancelled mel energies of the rectangular window: the looser bar of tests/test_oracle.py
        if d > (2e-2 if window == "rectangular" else 4e-3) or d2 > (2e-2 if window == "rectangular" else 4e-3):
            ok = False
    if not ok:
        bad += 1
        print(f"case {ci} window={window} snip={snip} lfr={lfr} lens={lens}: MISMATCH (worst so far {worst:.2e})")
print(json.dumps(dict(tool="fuzz_gpu_frontend_options", cases=n_cases, bad=bad, worst_logmel_abs_diff=worst)))
sys.exit(1 if bad else 0)
