#!/usr/bin/env python3
"""One-off (GPU): find the predictor bias for which the synthetic checkpoint emits ~4 tokens per second on the
synthetic speech-like clips (N ~ 120 tokens per 30 s clip), so the decoder sees a realistic N."""
import math
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from funasr_amd import synth
from funasr_amd.paraformer import Paraformer
from funasr_amd.wav_frontend import WavFrontend

dev = torch.device("cuda:0")
cfg = synth.PARAFORMER_LARGE
model = Paraformer.from_config(cfg)
sd = synth.paraformer_state_dict(cfg, seed=0)
b0 = float(sd["predictor.cif_output.bias"][0])
model.load_state_dict(sd, strict=False)
model = model.to(dev)
shift, scale = synth.synthetic_cmvn(560)
fe = WavFrontend(cmvn=torch.stack([shift, scale]), lfr_m=7, lfr_n=6, dither=0.0, device=dev)
n = 480000
wav = torch.stack([synth.speech_like(n, seed=i) for i in range(8)]).to(dev)
feats, flens = fe(wav, [n] * 8)
res = model.recognize_features(feats, flens, return_intermediate=True)
a = res["alphas"][:, :500].double().cpu().clamp(1e-9, 1 - 1e-9)
z = torch.log(a / (1 - a)) - b0
print("bias0", b0, "tokens", res["token_num"], "z mean/std", z.mean().item(), z.std().item())
for b in [x * 0.25 for x in range(-12, 9)]:
    m = torch.sigmoid(z + b).mean().item()
    print(f"bias {b:+.2f}: mean alpha {m:.4f} -> tokens/30s {m * 500:.1f}")
