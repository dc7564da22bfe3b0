#!/usr/bin/env python3
"""FSMN-VAD fuzz on the GPU: recordings of random length (0.3 s .. 150 s: zero, one or several 60-s decision blocks) with random bursts of
speech-like signal through FsmnVADStreaming.inference (HIP frontend, network, frame energies; host decision logic) against the CPU pipeline
of tests/test_vad_gpu.py::test_vad_inference_equals_cpu_pipeline: oracle frontend -> oracle FSMN -> numpy energies -> the decision logic that
tests/test_vad_decision.py and oracle/fuzz_vad_vs_reference.py pin to the reference. Segments must be equal frame for frame.
Not part of the test run. usage: fuzz_gpu_vad_vs_oracle.py [seed] [recordings]"""
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from funasr_amd import synth                                                        # noqa: E402
from funasr_amd.fsmn_vad import DEFAULT_SILENCE_SCHEDULE, FsmnVADStreaming          # noqa: E402
from funasr_amd.vad_decision import IN_SPEECH, VadDecision                          # noqa: E402
from funasr_amd.wav_frontend import WavFrontend                                     # noqa: E402
from oracle import paraformer_oracle as O                                           # noqa: E402
from oracle import vad_oracle                                                       # noqa: E402
from tests.test_vad_gpu import _enc_gold, _energy_tracking_weights                  # noqa: E402

dev = torch.device("cuda:0")
seed = int(sys.argv[1]) if len(sys.argv) > 1 else 0
n_rec = int(sys.argv[2]) if len(sys.argv) > 2 else 12
gen = torch.Generator().manual_seed(seed)
_, cfg = _enc_gold()
sd = _energy_tracking_weights(cfg)
model = FsmnVADStreaming(encoder="FSMN", encoder_conf=cfg)
model.encoder.load_state_dict(sd, strict=True)
model = model.to(dev)
cmvn = torch.zeros(2, 400); cmvn[0] = -8.0; cmvn[1] = 0.25
fe = WavFrontend(cmvn=cmvn, lfr_m=5, lfr_n=1, dither=0.0, device=dev)
fs = 16000


def rnd():
    return float(torch.rand(1, generator=gen))


bad, n_segments = 0, 0
for ri in range(n_rec):
    seconds = 0.3 + rnd() * (3.0 if ri % 4 == 0 else 150.0)
    total = int(seconds * fs)
    wav = 1e-4 * torch.randn(total, generator=gen)
    t = rnd() * 2.0
    k = 0
    while t < seconds - 0.2:
        dur = 0.05 + rnd() * (1.0 if rnd() < 0.4 else 25.0)
        dur = min(dur, seconds - t)
        seg = synth.speech_like(max(int(dur * fs), 1), seed=1000 * seed + 10 * ri + k)
        a = int(t * fs)
        wav[a: a + seg.numel()] += seg[: total - a]
        t += dur + 0.05 + rnd() * (0.3 if rnd() < 0.3 else 4.0)
        k += 1
    res, meta = model.inference([wav], key=["rec"], frontend=fe)
    got = res[0]["value"]
    feats, flens = O.wav_frontend([wav], cmvn, lfr_m=5, lfr_n=1)
    T = int(flens[0])
    p_sil = vad_oracle.fsmn_forward(feats[:, :T], sd, cfg)[0, :, 0].tolist()
    db = vad_oracle.frame_decibel(wav.numpy(), T).tolist()
    dec = VadDecision(model.vad_opts)
    want, done, acc, in_sp = [], 0, 0, False
    n_blocks = total // (60 * fs) + 1
    for b in range(n_blocks):
        last = b == n_blocks - 1
        if dec.state == IN_SPEECH or in_sp:
            acc, in_sp = acc + 60000, True
        for lim, sil in DEFAULT_SILENCE_SCHEDULE:
            if acc <= lim:
                dec.max_end_sil_ms, dec.speech_noise_thres = max(sil - 150, 0), 0.5
                break
        seen = min((b + 1) * 60 * fs, total)
        upto = T if last else (seen - 400) // 160 + 1 - 2
        segs = dec.push(p_sil[done:upto], db[done:upto], is_final=last)
        done = upto
        if segs:
            want += segs
            acc, in_sp = 0, False
    n_segments += len(want)
    if got != want:
        bad += 1
        first = next((i for i, (a, b) in enumerate(zip(got, want)) if a != b), min(len(got), len(want)))
        print(f"recording {ri}: {seconds:.2f} s, {k} bursts: {len(got)} segments against {len(want)}; first difference at {first}: "
              f"{got[first:first + 2]} / {want[first:first + 2]}")
print(json.dumps(dict(tool="fuzz_gpu_vad_vs_oracle", seed=seed, recordings=n_rec, segments=n_segments, bad=bad)))
sys.exit(1 if bad else 0)
