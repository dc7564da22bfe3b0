#!/usr/bin/env python3
"""Round 6 experiment (not product code): does MORE concurrency than "decoder beside the next encoder" pay? Same 64 x 30 s clips per step,
(a) the bench's loop: one replica, two phases, decoder on a second stream;
(b) two replicas, each a HALF batch (32 clips) per step on its own stream pair (encoder stream + decoder stream): four streams;
(c) two replicas, full batches dealt round-robin, each with its decoder stream.
Prints one JSON line per variant (audio-s/s over the same number of clips), same process, same box."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from funasr_amd import _lib, synth
from funasr_amd.paraformer import Paraformer
from funasr_amd.wav_frontend import WavFrontend

dev = torch.device("cuda:0")
cfg = synth.PARAFORMER_LARGE
sd = synth.paraformer_state_dict(cfg, seed=0, cif_bias=synth.BENCH_CIF_BIAS)
shift, scale = synth.synthetic_cmvn(560)
cmvn = torch.stack([shift, scale])
_lib.load().pf_set_concurrency_guard(1)


def replica():
    m = Paraformer.from_config(cfg)
    m.load_state_dict(sd, strict=False)
    return m.to(dev).set_precision("f16x2"), WavFrontend(cmvn=cmvn, lfr_m=7, lfr_n=6, dither=0.0, device=dev)


B, n = 64, 480000
wav = torch.stack([synth.speech_like(n, seed=i) for i in range(B)]).to(dev)
models = [replica(), replica()]
enc_s = [torch.cuda.Stream(device=dev) for _ in range(2)]
dec_s = [torch.cuda.Stream(device=dev) for _ in range(2)]


class Loop:
    """begin(i+1) -> finish(i) on the decoder stream -> collect(i-1), for one replica"""
    def __init__(self, r, rows):
        self.m, self.fe = models[r]
        self.es, self.ds = enc_s[r], dec_s[r]
        self.wav, self.lens = wav[rows], [n] * len(range(*rows.indices(B)))
        self.ticket = self.pending = None

    def step(self):
        with torch.cuda.stream(self.es):
            f, fl = self.fe(self.wav, self.lens)
            nxt = self.m.begin_features(f, fl)
            if self.ticket is not None:
                fin = self.m.finish_features(self.ticket, stream=self.ds)
                if self.pending is not None:
                    self.m.collect(self.pending)
                self.pending = fin
            self.ticket = nxt

    def drain(self):
        with torch.cuda.stream(self.es):
            fin = self.m.finish_features(self.ticket, stream=self.ds)
            if self.pending is not None:
                self.m.collect(self.pending)
            out = self.m.collect(fin)
        self.ticket = self.pending = None
        return out


def timed(name, loops, steps_each, clips_per_round):
    def run(k):
        for _ in range(k):
            for lp in loops:
                lp.step()
        return [lp.drain() for lp in loops]
    run(3)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    run(steps_each)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print(json.dumps({"variant": name, "audio_s_per_s": round(clips_per_round * steps_each * 30.0 / dt, 1), "ms_per_64_clips": round(dt / (clips_per_round * steps_each / 64) * 1e3, 2)}), flush=True)


for rep in range(2):
    timed("a: one replica, decoder stream", [Loop(0, slice(0, 64))], 20, 64)
    timed("b: two replicas x half batch, 4 streams", [Loop(0, slice(0, 32)), Loop(1, slice(32, 64))], 20, 64)
    timed("c: two replicas x full batch, 4 streams", [Loop(0, slice(0, 64)), Loop(1, slice(0, 64))], 10, 128)
