#!/usr/bin/env python3
"""Stress: the sweep's staging pattern (pinned arena -> zeroed device tensor -> non_blocking copies -> frontend) while another
process uses the GPU; reports where the features differ from those of the resident waveform."""
import os, sys, json, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from funasr_amd import synth
from funasr_amd.wav_frontend import WavFrontend
dev = torch.device("cuda:0")
tag = sys.argv[1] if len(sys.argv) > 1 else "p"
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 400
sync_before = len(sys.argv) > 3 and sys.argv[3] == "sync"
start_at = float(sys.argv[4]) if len(sys.argv) > 4 else 0.0
sh, sc = synth.synthetic_cmvn(560)
fe = WavFrontend(cmvn=torch.stack([sh, sc]), lfr_m=7, lfr_n=6, dither=0.0, device=dev)
lens = [98859, 70000, 235000, 34020, 104924, 49546]
host = [synth.speech_like(n, seed=7 + i) for i, n in enumerate(lens)]
arena = torch.empty(sum(lens)).pin_memory()
off, pinned = 0, []
for c in host:
    arena[off: off + c.numel()] = c
    pinned.append(arena[off: off + c.numel()])
    off += c.numel()
ref = [fe(c.to(dev)[None], [c.numel()])[0].clone() for c in host]
torch.cuda.synchronize()
while time.time() < start_at:
    time.sleep(0.01)
bad = []
t0 = time.time()
budget = float(os.environ.get("REPRO_SECONDS", "0"))
it = -1
while True:
    it += 1
    if budget > 0:
        if time.time() - t0 > budget:
            break
    elif it >= iters:
        break
    k = it % len(lens)
    wav = torch.zeros(1, lens[k], device=dev)
    wav[0, : lens[k]].copy_(pinned[k], non_blocking=True)
    if sync_before:
        torch.cuda.current_stream().synchronize()
    f, fl = fe(wav, [lens[k]])
    if not torch.equal(f, ref[k]):
        d = (f - ref[k]).abs()
        idx = torch.nonzero(d > 0)
        wav_ok = bool(torch.equal(wav[0].cpu(), host[k]))
        bad.append({"iter": it, "clip": k, "n_diff": int(idx.shape[0]), "max": float(d.max()), "wav_final_ok": wav_ok,
                    "rows": sorted(set(idx[:, 1].tolist()))[:10], "n_rows": len(set(idx[:, 1].tolist())), "T": f.shape[1]})
print(json.dumps({"tag": tag, "verify": bool(fe.verify), "faults_seen_by_the_cross_check": fe.faults(), "sync_before": sync_before, "iters": it, "seconds": round(time.time() - t0, 2), "mismatches": len(bad), "first": bad[:6]}))
