#!/usr/bin/env python3
"""FSMN-VAD throughput: whole recordings through FsmnVADStreaming.inference (frontend fbank/LFR(5,1)/CMVN + FSMN network
+ frame energies on the GPU, decision logic in native host code), audio-seconds per wall-second."""
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from funasr_amd import synth                                      # noqa: E402
from funasr_amd.fsmn_vad import FsmnVADStreaming                  # noqa: E402
from funasr_amd.wav_frontend import WavFrontend                   # noqa: E402
from oracle import vad_oracle                                     # noqa: E402  (seeded weights only)

dev = torch.device("cuda:0")
cfg = dict(input_dim=400, input_affine_dim=140, fsmn_layers=4, linear_dim=250, proj_dim=128, lorder=20, rorder=0, lstride=1,
           rstride=0, output_affine_dim=140, output_dim=248)
model = FsmnVADStreaming(encoder="FSMN", encoder_conf=cfg)
model.encoder.load_state_dict(vad_oracle.synthetic_state_dict(cfg, seed=1), strict=True)
model = model.to(dev)
cmvn = torch.zeros(2, 400); cmvn[0] = -8.0; cmvn[1] = 0.25
fe = WavFrontend(cmvn=cmvn, lfr_m=5, lfr_n=1, dither=0.0, device=dev)
out = {}
for seconds, reps in ((30, 40), (600, 10)):
    wav = synth.speech_like(seconds * 16000, seed=3) * (torch.arange(seconds * 16000) // 32000 % 2).float() + 1e-4 * torch.randn(seconds * 16000)
    wav = wav.to(dev)                                              # resident, like bench.py
    model.inference([wav], key=["w"], frontend=fe)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        res, _ = model.inference([wav], key=["w"], frontend=fe)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / reps
    out[f"{seconds}s"] = {"ms_per_recording": round(dt * 1e3, 2), "audio_s_per_s": round(seconds / dt, 1), "segments": len(res[0]["value"])}
print(json.dumps({"metric": "FSMN-VAD audio-seconds/sec, one recording per call (reference contract: batch 1)", **out}))
