"""Fuzz the FSMN-VAD decision logic against the REFERENCE class with injected scores (build container only; TEST
INFRASTRUCTURE): random option sets, random speech / silence patterns, random block sizes, offline and streaming reporting.
Compares funasr_amd.vad_decision.VadDecision and NativeVadDecision with FsmnVADStreaming.forward block by block."""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
from oracle import ref_import  # noqa: E402
from oracle.make_golden_vad import FLEN, SHIFT, pattern  # noqa: E402


def main(n_cases=400):
    ref_import.install()
    import funasr.models.fsmn_vad_streaming.encoder  # noqa: F401
    from funasr.models.fsmn_vad_streaming.model import FsmnVADStreaming
    from funasr_amd.vad_decision import NativeVadDecision, VadDecision, VadOptions
    rng = np.random.default_rng(2024)
    enc_conf = dict(input_dim=400, input_affine_dim=140, fsmn_layers=4, linear_dim=250, proj_dim=128, lorder=20, rorder=0,
                    lstride=1, rstride=0, output_affine_dim=140, output_dim=248)
    bad = {"python": 0, "native": 0}
    events = 0
    for ci in range(n_cases):
        opts = dict(do_extend=int(rng.integers(0, 2)), detect_mode=int(rng.integers(0, 2)),
                    max_end_silence_time=int(rng.choice([200, 300, 800, 1500])),
                    max_start_silence_time=int(rng.choice([500, 3000])),
                    max_single_segment_time=int(rng.choice([1500, 4000, 60000])),
                    window_size_ms=int(rng.choice([100, 200, 300])), sil_to_speech_time_thres=int(rng.choice([50, 150])),
                    speech_to_sil_time_thres=int(rng.choice([50, 150])), lookback_time_start_point=int(rng.choice([0, 100, 200])),
                    lookahead_time_end_point=int(rng.choice([0, 100, 200])), decibel_thres=float(rng.choice([-100.0, -50.0, -40.0])),
                    snr_thres=float(rng.choice([-100.0, 3.0])), speech_noise_thres=float(rng.choice([0.3, 0.6, 0.8])),
                    speech_2_noise_ratio=float(rng.choice([1.0, 1.5])))
        if opts["speech_to_sil_time_thres"] > opts["window_size_ms"] or opts["sil_to_speech_time_thres"] > opts["window_size_ms"]:
            continue
        if opts["max_end_silence_time"] < opts["speech_to_sil_time_thres"]:
            continue
        kind = ("normal", "blips", "long")[ci % 3]
        n = int(rng.integers(30, 1500))
        p_sil, amp = pattern(rng, n, kind)
        n_samp = (n - 1) * SHIFT + FLEN
        env = np.concatenate([np.repeat(amp, SHIFT), np.full(FLEN - SHIFT, amp[-1], np.float32)])[:n_samp]
        wave = rng.standard_normal(n_samp).astype(np.float32) * env.astype(np.float32)
        events_mode = bool(rng.integers(0, 2))
        blocks, left = [], n
        while left > 0:
            b = min(left, int(rng.integers(1, 60 if events_mode else 500)))
            blocks.append(b)
            left -= b
        model = FsmnVADStreaming(encoder="FSMN", encoder_conf=enc_conf, **opts)
        feed = {}

        class Injected(torch.nn.Module):
            def forward(self, feats, cache=None):
                return feed["scores"]
        model.encoder = Injected()
        cache = {}
        model.init_cache(cache)
        mine = {"python": VadDecision(VadOptions(**opts)), "native": NativeVadDecision(VadOptions(**opts))}
        f0 = 0
        ok = {"python": True, "native": True}
        for bi, b in enumerate(blocks):
            final = bi == len(blocks) - 1
            span = wave[f0 * SHIFT: (f0 + b - 1) * SHIFT + FLEN]
            frames = span[np.arange(0, span.shape[0] - FLEN + 1, SHIFT)[:, None] + np.arange(FLEN)]
            db = (10 * np.log10(np.sum(np.square(frames), axis=1) + 0.000001))
            sc = torch.zeros(1, b, 2)
            sc[0, :, 0] = torch.from_numpy(p_sil[f0: f0 + b])
            feed["scores"] = sc
            try:
                seg = model.forward(feats=torch.zeros(1, b, 400), waveform=torch.from_numpy(span)[None], cache=cache,
                                    is_final=final, is_streaming_input=events_mode)
                want = [list(map(int, s)) for s in (seg[0] if len(seg) else [])]
            except Exception as e:                          # noqa: BLE001 - e.g. the reference's own assert on odd options
                want = ("EXC", type(e).__name__)
            for k, d in mine.items():
                if not ok[k]:
                    continue
                try:
                    got = d.push(p_sil[f0: f0 + b].tolist() if k == "python" else p_sil[f0: f0 + b],
                                 db.tolist() if k == "python" else db.astype(np.float32), final, events_mode)
                except Exception as e:                      # noqa: BLE001
                    got = ("EXC", type(e).__name__)
                if isinstance(want, tuple) or isinstance(got, tuple):
                    if isinstance(want, tuple) != isinstance(got, tuple):
                        ok[k] = False
                        print(k, "case", ci, "block", bi, opts, want, got)
                    ok[k] = False if isinstance(want, tuple) else ok[k]
                    continue
                if got != want:
                    ok[k] = False
                    print(k, "case", ci, "block", bi, opts, want, got)
                events += len(got)
            if isinstance(want, tuple):
                break
            f0 += b
        for k in ok:
            bad[k] += 0 if ok[k] else 1
    print(f"cases {n_cases}, events compared {events}, scenarios with a difference: {bad}")


if __name__ == "__main__":
    main(int(sys.argv[1]) if len(sys.argv) > 1 else 400)
