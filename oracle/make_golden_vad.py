#!/usr/bin/env python3
"""Golden vectors for the FSMN-VAD decision logic, produced by the REFERENCE's own class
(funasr/models/fsmn_vad_streaming/model.py `FsmnVADStreaming.forward`) with the network replaced by injected scores:
the reference computes the frame energies from the waveform, classifies frames, runs its window detector and start /
end point state machine, drops history and reports segments exactly as in production; only the FSMN's output is given.
TEST INFRASTRUCTURE: run in the build container (needs /root/reference); writes tests/golden/vad_decision.npz.

Scenarios: speech / silence patterns with blips, long pauses and over-long speech; offline (one final block; several
blocks with the last one final) and streaming-event reporting in small blocks; option variants (no look-ahead
extension, single-utterance mode with start-silence timeout, energy and SNR thresholds that actually bite, short
maximum segment length, different end-silence times).
"""
import json
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
from oracle import ref_import  # noqa: E402

SHIFT, FLEN = 160, 400


def pattern(rng, n_frames, kind):
    """-> (p_sil [n], amplitude per frame [n])"""
    p = np.empty(n_frames, np.float32)
    amp = np.empty(n_frames, np.float32)
    t, speech = 0, bool(rng.random() < 0.3)
    while t < n_frames:
        if speech:
            d = int(rng.integers(3, 40)) if kind == "blips" else int(rng.integers(30, 900 if kind == "long" else 300))
        else:
            d = int(rng.integers(2, 30)) if kind == "blips" else int(rng.integers(10, 250))
        e = min(n_frames, t + d)
        if speech:
            p[t:e] = np.clip(0.08 + 0.1 * rng.standard_normal(e - t), 1e-4, 0.6)
            amp[t:e] = 0.05 + 0.2 * rng.random(e - t)
        else:
            p[t:e] = np.clip(0.92 + 0.1 * rng.standard_normal(e - t), 0.3, 1 - 1e-4)
            amp[t:e] = 1e-4 * (1 + 20 * rng.random()) * (1 + rng.random(e - t))
        t, speech = e, not speech
    return p, amp


def main():
    ref_import.install()
    import funasr.models.fsmn_vad_streaming.encoder  # noqa: F401  (registers FSMN)
    from funasr.models.fsmn_vad_streaming.model import FsmnVADStreaming
    rng = np.random.default_rng(77)
    enc_conf = dict(input_dim=400, input_affine_dim=140, fsmn_layers=4, linear_dim=250, proj_dim=128, lorder=20, rorder=0,
                    lstride=1, rstride=0, output_affine_dim=140, output_dim=248)
    variants = [dict(), dict(do_extend=0), dict(detect_mode=0, max_start_silence_time=1200), dict(decibel_thres=-45.0),
                dict(snr_thres=5.0), dict(max_single_segment_time=3000), dict(max_end_silence_time=300),
                dict(max_end_silence_time=1500, lookahead_time_end_point=200, lookback_time_start_point=100),
                dict(speech_noise_thres=0.3, window_size_ms=100, sil_to_speech_time_thres=60, speech_to_sil_time_thres=60)]
    cases, all_p, all_db = [], [], []
    for ci in range(72):
        opts = variants[ci % len(variants)]
        kind = ("normal", "blips", "long")[ci % 3]
        n = int(rng.integers(40, 2500 if kind == "long" else 1200))
        p_sil, amp = pattern(rng, n, kind)
        n_samp = (n - 1) * SHIFT + FLEN
        env = np.concatenate([np.repeat(amp, SHIFT), np.full(FLEN - SHIFT, amp[-1], np.float32)])[:n_samp]
        wave = rng.standard_normal(n_samp).astype(np.float32) * env.astype(np.float32)
        mode = ("offline_one", "offline_blocks", "stream_events")[(ci // 3) % 3]
        if mode == "offline_one":
            blocks = [n]
        elif mode == "offline_blocks":
            blocks, left = [], n
            while left > 0:
                b = min(left, int(rng.integers(50, 700)))
                blocks.append(b)
                left -= b
        else:
            blocks, left = [], n
            while left > 0:
                b = min(left, int(rng.integers(1, 40)))
                blocks.append(b)
                left -= b
        model = FsmnVADStreaming(encoder="FSMN", encoder_conf=enc_conf, **opts)
        feed = {}

        class Injected(torch.nn.Module):                    # stands in for the FSMN: returns the scores of this block
            def forward(self, feats, cache=None):
                return feed["scores"]
        model.encoder = Injected()
        cache = {}
        model.init_cache(cache)
        outs, decibels, f0 = [], [], 0
        for bi, b in enumerate(blocks):
            final = bi == len(blocks) - 1
            span = wave[f0 * SHIFT: (f0 + b - 1) * SHIFT + FLEN]
            frames = span[np.arange(0, span.shape[0] - FLEN + 1, SHIFT)[:, None] + np.arange(FLEN)]
            decibels += (10 * np.log10(np.sum(np.square(frames), axis=1) + 0.000001)).tolist()      # as ComputeDecibel does
            sc = torch.zeros(1, b, 2)
            sc[0, :, 0] = torch.from_numpy(p_sil[f0: f0 + b])
            sc[0, :, 1] = 1 - sc[0, :, 0]
            feed["scores"] = sc
            seg = model.forward(feats=torch.zeros(1, b, 400), waveform=torch.from_numpy(span)[None], cache=cache,
                                is_final=final, is_streaming_input=(mode == "stream_events"))
            outs.append([list(map(int, s)) for s in (seg[0] if len(seg) else [])])
            f0 += b
        cases.append(dict(options=opts, kind=kind, mode=mode, blocks=blocks, segments_per_block=outs, n=n))
        all_p.append(p_sil.astype(np.float32))
        all_db.append(np.asarray(decibels, dtype=np.float64))
    out = os.path.join(os.path.dirname(HERE), "tests", "golden", "vad_decision.npz")
    meta = dict(generator="oracle/make_golden_vad.py",
                reference="funasr/models/fsmn_vad_streaming/model.py FsmnVADStreaming.forward with injected scores", cases=cases)
    np.savez_compressed(out, p_sil=np.concatenate(all_p), decibel=np.concatenate(all_db), meta=json.dumps(meta))
    # ---- the network: the reference's FSMN class on seeded weights, whole utterance and chunked with its cache dict
    from funasr.models.fsmn_vad_streaming.encoder import FSMN
    from oracle import vad_oracle
    enc_cfg = dict(enc_conf)
    sd = vad_oracle.synthetic_state_dict(enc_cfg, seed=5)
    ref = FSMN(**enc_cfg)
    ref.load_state_dict(sd, strict=True)
    ref.eval()
    tg = torch.Generator().manual_seed(6)
    feats = torch.randn(2, 137, 400, generator=tg)
    with torch.no_grad():
        whole = ref(feats, cache=None)                                  # zero left context
        rc, parts = {}, []
        for a, b in ((0, 1), (1, 9), (9, 60), (60, 137)):
            parts.append(ref(feats[:1, a:b], cache=rc))
        chunked = torch.cat(parts, dim=1)
    mine = vad_oracle.fsmn_forward(feats, sd, enc_cfg)
    oc, oparts = {}, []
    for a, b in ((0, 1), (1, 9), (9, 60), (60, 137)):
        oparts.append(vad_oracle.fsmn_forward(feats[:1, a:b], sd, enc_cfg, cache=oc))
    err = max((mine - whole).abs().max().item(), (torch.cat(oparts, 1) - chunked).abs().max().item())
    assert err < 1e-6, f"oracle restatement differs from the reference FSMN by {err}"
    enc_out = os.path.join(os.path.dirname(HERE), "tests", "golden", "vad_encoder.npz")
    np.savez_compressed(enc_out, cfg=json.dumps(enc_cfg), seed=5, feats=feats.numpy(), probs_whole=whole.numpy(),
                        probs_chunked=chunked.numpy(), chunks=np.array([[0, 1], [1, 9], [9, 60], [60, 137]]))
    print(f"wrote {enc_out}: oracle vs reference FSMN max |d| = {err:.2e}; chunked vs whole (reference) "
          f"{(chunked - whole[:1]).abs().max().item():.2e}")
    # ---- end to end: the reference's FsmnVADStreaming.inference with ITS WavFrontendOnline (only torchaudio's kaldi.fbank is
    #      the oracle's restatement, pinned to kaldi-native-fbank in tests/test_oracle.py), a hand-wired network whose
    #      silence posterior follows the frame level, on a seeded 70 s recording: whole recording (60 s blocks, dynamic
    #      end-silence schedule) and streaming input in 200 ms chunks fed over several calls
    import tempfile
    import torchaudio.compliance.kaldi as kaldi
    from oracle import paraformer_oracle as O
    from funasr_amd import synth
    sys.path.insert(0, os.path.join(os.path.dirname(HERE), "tests"))
    from _model_dir import write_mvn
    from test_vad_gpu import _energy_tracking_weights

    def fbank(waveform, num_mel_bins=80, frame_length=25, frame_shift=10, dither=0.0, energy_floor=0.0,
              window_type="hamming", sample_frequency=16000, **kw):
        assert dither == 0.0 and window_type == "hamming" and energy_floor == 0.0
        return O.kaldi_fbank(waveform[0], num_mel_bins, float(frame_length), float(frame_shift), float(sample_frequency))
    kaldi.fbank = fbank
    import funasr.frontends.wav_frontend as wf
    wf.kaldi.fbank = fbank
    tmp = tempfile.mkdtemp()
    write_mvn(os.path.join(tmp, "am.mvn"), torch.full((400,), -8.0), torch.full((400,), 0.25))
    fe = wf.WavFrontendOnline(cmvn_file=os.path.join(tmp, "am.mvn"), fs=16000, window="hamming", n_mels=80, frame_length=25,
                              frame_shift=10, dither=0.0, lfr_m=5, lfr_n=1)
    fs, total = 16000, 70 * 16000
    wav = 1e-4 * torch.randn(total, generator=torch.Generator().manual_seed(3))
    bursts = [(1.0, 4.2), (6.0, 6.3), (9.5, 21.0), (30.0, 58.0), (59.2, 61.0), (66.0, 69.8)]
    for i, (a, b) in enumerate(bursts):
        seg = synth.speech_like(int((b - a) * fs), seed=50 + i)
        wav[int(a * fs): int(a * fs) + seg.numel()] += seg
    vm = FsmnVADStreaming(encoder="FSMN", encoder_conf=enc_conf)
    vm.encoder.load_state_dict(_energy_tracking_weights(enc_conf), strict=True)
    vm.eval()
    with torch.no_grad():
        offline, _ = vm.inference([wav], key=["rec"], frontend=fe, device="cpu")
        cache, calls, events, pos = {}, [7000, 16000, 3300, 200000, 50000, 400000, 123456], [], 0
        ci = 0
        while pos < total:
            n = calls[ci % len(calls)]
            ci += 1
            chunk = wav[pos: pos + n]
            pos += n
            res, _ = vm.inference([chunk], key=["rec"], frontend=fe, device="cpu", cache=cache, chunk_size=200,
                                  is_final=pos >= total)
            events.append(dict(samples=int(chunk.numel()), final=bool(pos >= total), value=res[0]["value"]))
    e2e = dict(bursts=bursts, offline=offline[0]["value"], streaming_calls=events)
    with open(os.path.join(os.path.dirname(HERE), "tests", "golden", "vad_e2e.json"), "w") as f:
        json.dump(e2e, f)
    print(f"e2e: offline {offline[0]['value']}; streaming {sum(len(e['value']) for e in events)} events over {len(events)} calls")
    nseg = sum(len(s) for c in cases for s in c["segments_per_block"])
    print(f"wrote {out}: {len(cases)} cases, {nseg} reported segments/events, {os.path.getsize(out) / 1e6:.2f} MB")


if __name__ == "__main__":
    main()
