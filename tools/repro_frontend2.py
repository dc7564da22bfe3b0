#!/usr/bin/env python3
"""Under external GPU load: which fbank frames come out wrong, and what do they look like?"""
import os, sys, json, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from funasr_amd import synth
from funasr_amd.wav_frontend import WavFrontend
dev = torch.device("cuda:0")
sh, sc = synth.synthetic_cmvn(560)
fe = WavFrontend(cmvn=torch.stack([sh, sc]), lfr_m=7, lfr_n=6, dither=0.0, device=dev)
n = 235000
wav = synth.speech_like(n, seed=9).to(dev)[None]
_, _, ref = fe(wav, [n], return_fbank=True)
ref = ref.clone()
torch.cuda.synchronize()
t0 = time.time(); it = 0; found = []
while time.time() - t0 < float(os.environ.get("REPRO_SECONDS", "20")) and len(found) < 12:
    it += 1
    _, _, fb = fe(wav, [n], return_fbank=True)
    if not torch.equal(fb, ref):
        d = (fb - ref).abs().amax(dim=2)[0]
        for f in torch.nonzero(d > 0)[:, 0].tolist()[:4]:
            row, want = fb[0, f], ref[0, f]
            dist = (ref[0] - row[None]).abs().amax(dim=1)
            j = int(dist.argmin())
            found.append({"iter": it, "frame": f, "n_frames": ref.shape[1], "maxdiff": float((row - want).abs().max()),
                          "n_bins_diff": int(((row - want).abs() > 0).sum()), "closest_ref_frame": j, "closest_dist": float(dist[j]),
                          "got_first6": [round(float(v), 4) for v in row[:6]], "want_first6": [round(float(v), 4) for v in want[:6]],
                          "finite": bool(torch.isfinite(row).all()), "const_offset": round(float((row - want).mean()), 4),
                          "offset_std": round(float((row - want).std()), 4)})
print(json.dumps({"iters": it, "found": found}, indent=0))
