#!/usr/bin/env python3
"""Stress: the frontend on the same clip over and over while another process uses the GPU; reports where results differ."""
import os, sys, json, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from funasr_amd import synth
from funasr_amd.wav_frontend import WavFrontend
dev = torch.device("cuda:0")
tag = sys.argv[1] if len(sys.argv) > 1 else "p"
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 400
sh, sc = synth.synthetic_cmvn(560)
fe = WavFrontend(cmvn=torch.stack([sh, sc]), lfr_m=7, lfr_n=6, dither=0.0, device=dev)
clips = [synth.speech_like(n, seed=7 + i).to(dev)[None] for i, n in enumerate((98859, 70000, 235000, 34020))]
lens = [c.shape[1] for c in clips]
ref = []
for c, n in zip(clips, lens):
    f, fl, fb = fe(c, [n], return_fbank=True)
    ref.append((f.clone(), fb.clone()))
torch.cuda.synchronize()
bad = []
filler = torch.randn(4096, 4096, device=dev)
for it in range(iters):
    k = it % len(clips)
    if it % 7 == 0:
        filler = filler @ filler * 1e-4                     # other work on the stream
    f, fl, fb = fe(clips[k], [lens[k]], return_fbank=True)
    f2, _ = fe(clips[k], [lens[k]])
    for name, got, want in (("feats_with_fbank_out", f, ref[k][0]), ("fbank", fb, ref[k][1]), ("feats", f2, ref[k][0])):
        if not torch.equal(got, want):
            d = (got - want).abs()
            idx = torch.nonzero(d > 0)
            bad.append({"iter": it, "clip": k, "what": name, "n_diff": int(idx.shape[0]), "max": float(d.max()),
                        "rows": sorted(set(idx[:, 1].tolist()))[:12], "cols": sorted(set(idx[:, 2].tolist()))[:12],
                        "shape": list(got.shape)})
print(json.dumps({"tag": tag, "iters": iters, "mismatches": len(bad), "first": bad[:8]}))
