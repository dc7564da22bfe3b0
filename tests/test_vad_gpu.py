"""FSMN-VAD on the GPU: the network against golden posteriors from the reference's own FSMN class, the frame energies
against ComputeDecibel's numpy expression, and the whole `FsmnVADStreaming.inference` against the CPU oracle pipeline
(oracle frontend -> oracle FSMN -> numpy energies -> the decision logic that tests/test_vad_decision.py pins to the
reference)."""
import json
import os

import numpy as np
import pytest
import torch

from funasr_amd import synth

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _enc_gold():
    g = np.load(os.path.join(GOLD, "vad_encoder.npz"), allow_pickle=False)
    return g, json.loads(str(g["cfg"]))


def test_oracle_network_equals_reference_golden():
    from oracle import vad_oracle
    g, cfg = _enc_gold()
    sd = vad_oracle.synthetic_state_dict(cfg, seed=int(g["seed"]))
    feats = torch.from_numpy(g["feats"])
    assert (vad_oracle.fsmn_forward(feats, sd, cfg) - torch.from_numpy(g["probs_whole"])).abs().max().item() < 1e-6
    cache, parts = {}, []
    for a, b in g["chunks"].tolist():
        parts.append(vad_oracle.fsmn_forward(feats[:1, a:b], sd, cfg, cache=cache))
    assert (torch.cat(parts, 1) - torch.from_numpy(g["probs_chunked"])).abs().max().item() < 1e-6


@pytest.mark.gpu
def test_fsmn_network_vs_reference_golden(cuda):
    from funasr_amd.fsmn_vad import FSMN
    from oracle import vad_oracle
    g, cfg = _enc_gold()
    enc = FSMN(**cfg)
    enc.load_state_dict(vad_oracle.synthetic_state_dict(cfg, seed=int(g["seed"])), strict=True)
    enc = enc.to(cuda)
    feats = torch.from_numpy(g["feats"]).to(cuda)
    probs = enc(feats)                                            # batch of 2, zero left context
    assert (probs.cpu() - torch.from_numpy(g["probs_whole"])).abs().max().item() < 2e-6
    p_sil = enc.silence_posterior(feats, None, [0, 3, 7])
    assert torch.allclose(p_sil.cpu(), torch.from_numpy(g["probs_whole"])[..., [0, 3, 7]].sum(-1), atol=2e-6)
    cache, parts = {}, []                                         # stream 0 chunk by chunk, context carried in HBM
    for a, b in g["chunks"].tolist():
        parts.append(enc(feats[:1, a:b], cache=cache))
    chunked = torch.cat(parts, 1).cpu()
    assert (chunked - torch.from_numpy(g["probs_chunked"])).abs().max().item() < 2e-6
    assert cache["fsmn_ctx"].shape == (1, cfg["fsmn_layers"], (cfg["lorder"] - 1) * cfg["lstride"], cfg["proj_dim"])


@pytest.mark.gpu
def test_frame_decibel_vs_numpy_expression(cuda):
    from funasr_amd.fsmn_vad import frame_decibel
    from oracle import vad_oracle
    g = torch.Generator().manual_seed(1)
    wav = torch.randn(16000 * 3, generator=g) * torch.linspace(1e-5, 0.5, 16000 * 3)
    n = (wav.numel() - 400) // 160 + 1
    got = frame_decibel(wav.to(cuda), n).cpu().numpy()
    want = vad_oracle.frame_decibel(wav.numpy(), n)
    assert got.shape == want.shape and np.abs(got - want).max() < 1e-3
    with pytest.raises(RuntimeError, match="aligned"):
        frame_decibel(wav.to(cuda), n + 1)


def _energy_tracking_weights(cfg):
    """hand-wired FSMN whose silence posterior follows the mean log-mel of a frame (loud -> speech, quiet -> silence),
    with a memory tap so that the context cache matters"""
    sd = {}
    z = lambda *s: torch.zeros(*s)                                                  # noqa: E731
    w = z(cfg["input_affine_dim"], cfg["input_dim"]); w[0] = 1.0 / cfg["input_dim"]
    sd["in_linear1.linear.weight"], sd["in_linear1.linear.bias"] = w, z(cfg["input_affine_dim"])
    w = z(cfg["linear_dim"], cfg["input_affine_dim"]); w[0, 0], w[1, 0] = -1.0, 1.0
    b = z(cfg["linear_dim"]); b[0], b[1] = 0.3, 0.3
    sd["in_linear2.linear.weight"], sd["in_linear2.linear.bias"] = w, b
    for i in range(cfg["fsmn_layers"]):
        w = z(cfg["proj_dim"], cfg["linear_dim"]); w[0, 0] = w[1, 1] = 1.0
        sd[f"fsmn.{i}.linear.linear.weight"] = w
        c = z(cfg["proj_dim"], 1, cfg["lorder"], 1); c[:2, 0, -3:-1, 0] = 0.05        # a little smoothing over past frames
        sd[f"fsmn.{i}.fsmn_block.conv_left.weight"] = c
        w = z(cfg["linear_dim"], cfg["proj_dim"]); w[0, 0] = w[1, 1] = 1.0
        sd[f"fsmn.{i}.affine.linear.weight"], sd[f"fsmn.{i}.affine.linear.bias"] = w, z(cfg["linear_dim"])
    w = z(cfg["output_affine_dim"], cfg["linear_dim"]); w[0, 0] = w[1, 1] = 1.0
    sd["out_linear1.linear.weight"], sd["out_linear1.linear.bias"] = w, z(cfg["output_affine_dim"])
    w = z(cfg["output_dim"], cfg["output_affine_dim"]); w[0, 0], w[0, 1] = 15.0, -15.0
    sd["out_linear2.linear.weight"], sd["out_linear2.linear.bias"] = w, z(cfg["output_dim"])
    return sd


@pytest.mark.gpu
def test_vad_inference_equals_cpu_pipeline(cuda):
    """a 70 s recording (two 60 s blocks for the decision logic, dynamic end-silence schedule on) with bursts of
    speech-like signal: the segments of the HIP path equal those of the CPU pipeline, frame for frame"""
    from funasr_amd.fsmn_vad import DEFAULT_SILENCE_SCHEDULE, FsmnVADStreaming
    from funasr_amd.vad_decision import IN_SPEECH, VadDecision
    from funasr_amd.wav_frontend import WavFrontend
    from oracle import paraformer_oracle as O
    from oracle import vad_oracle
    _, cfg = _enc_gold()
    model = FsmnVADStreaming(encoder="FSMN", encoder_conf=cfg)
    sd = _energy_tracking_weights(cfg)
    model.encoder.load_state_dict(sd, strict=True)
    model = model.to(cuda)
    fs, total = 16000, 70 * 16000
    wav = 1e-4 * torch.randn(total, generator=torch.Generator().manual_seed(3))
    bursts = [(1.0, 4.2), (6.0, 6.3), (9.5, 21.0), (30.0, 58.0), (59.2, 61.0), (66.0, 69.8)]
    for i, (a, b) in enumerate(bursts):
        seg = synth.speech_like(int((b - a) * fs), seed=50 + i)
        wav[int(a * fs): int(a * fs) + seg.numel()] += seg
    cmvn = torch.zeros(2, 400); cmvn[0] = -8.0; cmvn[1] = 0.25                       # centre the log-mels around 0
    fe = WavFrontend(cmvn=cmvn, lfr_m=5, lfr_n=1, dither=0.0, device=cuda)
    res, meta = model.inference([wav], key=["rec"], frontend=fe)
    got = res[0]["value"]
    assert res[0]["key"] == "rec" and len(got) >= 4 and meta["batch_data_time"] > 69
    # ---- the same on the CPU
    feats, flens = O.wav_frontend([wav], cmvn, lfr_m=5, lfr_n=1)
    T = int(flens[0])
    p_sil = vad_oracle.fsmn_forward(feats[:, :T], sd, cfg)[0, :, 0].tolist()
    db = vad_oracle.frame_decibel(wav.numpy(), T).tolist()
    dec = VadDecision(model.vad_opts)
    want, done, acc, in_sp = [], 0, 0, False
    for b in range(total // (60 * fs) + 1):
        last = b == total // (60 * fs)
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
    assert got == want, (got, want)
    # ... and those of the REFERENCE's own FsmnVADStreaming.inference (its online frontend, its network class, its state
    # machine) on the same recording and weights (tests/golden/vad_e2e.json, oracle/make_golden_vad.py)
    with open(os.path.join(GOLD, "vad_e2e.json")) as f:
        e2e = json.load(f)
    assert got == e2e["offline"], (got, e2e["offline"])
    # sanity of the hand-wired network: every long burst is found, the gaps are silence
    for a, b in [(9.5, 21.0), (30.0, 58.0)]:
        assert any(s[0] <= a * 1000 + 400 and s[1] >= min(b, s[1] / 1000) * 1000 - 400 for s in got)
    assert model.encoder.silence_posterior(fe(wav[None].to(cuda), [total])[0][:, 300:310]).mean().item() < 0.5   # inside a burst


@pytest.mark.gpu
def test_vad_streaming_input_equals_reference_events(cuda):
    """the same recording fed in 13 calls of arbitrary length with chunk_size = 200 ms and streaming reporting: the
    events of every call ([beg, -1] / [-1, end]) equal the reference's (per-chunk dynamic end-silence schedule, online
    frontend with its LFR look-ahead and final flush, network left context carried in HBM)"""
    from funasr_amd.fsmn_vad import FsmnVADStreaming
    from funasr_amd.paraformer_streaming import WavFrontendOnline
    _, cfg = _enc_gold()
    with open(os.path.join(GOLD, "vad_e2e.json")) as f:
        e2e = json.load(f)
    model = FsmnVADStreaming(encoder="FSMN", encoder_conf=cfg)
    model.encoder.load_state_dict(_energy_tracking_weights(cfg), strict=True)
    model = model.to(cuda)
    fs, total = 16000, 70 * 16000
    wav = 1e-4 * torch.randn(total, generator=torch.Generator().manual_seed(3))
    for i, (a, b) in enumerate(e2e["bursts"]):
        seg = synth.speech_like(int((b - a) * fs), seed=50 + i)
        wav[int(a * fs): int(a * fs) + seg.numel()] += seg
    cmvn = torch.zeros(2, 400); cmvn[0] = -8.0; cmvn[1] = 0.25
    fe = WavFrontendOnline(cmvn=cmvn, lfr_m=5, lfr_n=1, dither=0.0, device=cuda)
    cache, pos = {}, 0
    for ci, call in enumerate(e2e["streaming_calls"]):
        chunk = wav[pos: pos + call["samples"]]
        pos += call["samples"]
        res, _ = model.inference([chunk], key=["rec"], frontend=fe, cache=cache, chunk_size=200, is_final=call["final"])
        assert res[0]["value"] == call["value"], (ci, res[0]["value"], call["value"])
    assert pos == total and len(cache["prev_samples"]) == 0 and cache["frames_done"] == 0      # re-initialised after the final call
