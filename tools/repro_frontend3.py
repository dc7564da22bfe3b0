#!/usr/bin/env python3
"""Under external GPU load: at which stage of fbank_kernel does a wrong frame first differ?"""
import os, sys, json, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from funasr_amd import synth, _lib
from funasr_amd.wav_frontend import WavFrontend
dev = torch.device("cuda:0")
lib = _lib.load()
sh, sc = synth.synthetic_cmvn(560)
fe = WavFrontend(cmvn=torch.stack([sh, sc]), lfr_m=7, lfr_n=6, dither=0.0, device=dev)
n = 235000
wav = synth.speech_like(n, seed=9).to(dev)[None]
nf = 1 + (n - 400) // 160
dump = torch.zeros(nf, 2400, device=dev)
lib.pf_debug_set_fbank_dump(dump.data_ptr())
runs = []
for _ in range(5):
    fe(wav, [n]); torch.cuda.synchronize(); runs.append(dump.clone())
ref = torch.stack(runs).median(dim=0).values
t0 = time.time(); it = 0; found = []
while time.time() - t0 < float(os.environ.get("REPRO_SECONDS", "20")) and len(found) < 10:
    it += 1
    fe(wav, [n]); torch.cuda.synchronize()
    if not torch.equal(dump, ref):
        d = (dump != ref)
        d[0, 2399] = False
        for f in torch.nonzero(d.any(dim=1))[:, 0].tolist()[:3]:
            row = d[f]
            st = {"samples": row[:512], "power": row[512:769], "pieces": row[769:897], "regs_natural": row[900:1412],
                  "z_after_write": row[1412:1924], "z_after_power": row[1924:2436]}
            rec = {"iter": it, "frame": f}
            for k, m in st.items():
                idx = torch.nonzero(m)[:, 0].tolist()
                rec[k] = {"n": len(idx), "idx": idx[:40]}
            found.append(rec)
lib.pf_debug_set_fbank_dump(None)
print(json.dumps({"iters": it, "n_found": len(found), "rewrites_counter_bits": int(dump[0, 2399].view(torch.int32).item()), "found": found[:3]}))
