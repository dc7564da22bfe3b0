"""The streaming oracle (oracle/streaming_oracle.py) against the golden fixture produced by the REFERENCE's own
ParaformerStreaming / WavFrontendOnline classes (oracle/make_golden_streaming.py): per 600 ms chunk the online
features, the encoder window output, the carried CIF state and the token ids."""
import json
import os

import numpy as np
import pytest
import torch

from funasr_amd import synth
from oracle import paraformer_oracle as O
from oracle import streaming_oracle as S

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load():
    g = np.load(os.path.join(GOLD, "streaming.npz"), allow_pickle=False)
    cfg = json.loads(bytes(g["config"]).decode())
    sd = synth.paraformer_state_dict(cfg, seed=int(g["seed"]), cif_bias=float(g["cif_bias"]))
    wav = torch.from_numpy(g["pcm"].astype(np.float32) / 32768.0)
    cmvn = O.load_cmvn(os.path.join(GOLD, "am.mvn"))
    return g, cfg, sd, wav, cmvn


def test_online_lfr_matches_offline_lfr_when_fed_whole():
    """Feeding everything in one final call must reproduce the offline LFR (same frames, 3-frame left replicate)."""
    fb = torch.randn(61, 80)
    feats = torch.cat((fb[0:1].repeat(3, 1), fb), 0)
    out, rest = S.online_lfr(feats, 7, 6, True)
    assert torch.equal(out, O.apply_lfr(fb, 7, 6))


def test_streaming_oracle_matches_reference_chunks():
    g, cfg, sd, wav, cmvn = load()
    n1 = int(g["n1"])
    st = S.model_init(cfg)
    trace = []
    with torch.no_grad():
        t1 = S.streaming_inference(wav[:n1], st, sd, cfg, cmvn, False, trace)
        t2 = S.streaming_inference(wav[n1:], st, sd, cfg, cmvn, True, trace)
    n_chunks = int(g["n_chunks"])
    assert len(trace) == n_chunks
    toks = []
    for i, rec in enumerate(trace):
        fin, tail, start_idx = (int(v) for v in g[f"flags_{i}"])
        enc_ref = torch.from_numpy(g[f"enc_{i}"])
        assert rec["enc"].shape == enc_ref.shape, (i, rec["enc"].shape, enc_ref.shape)
        assert (rec["enc"] - enc_ref).abs().max().item() < 2e-5, i
        ids = [t for t in rec.get("raw_ids", []) if t not in (0, 1, 2)]
        assert ids == g[f"tokens_{i}"].tolist(), i
        toks += ids
    assert toks == t1 + t2
    # carried CIF state after the last chunk
    last = n_chunks - 1
    assert abs(float(st["cif_alphas"]) - float(g[f"cif_alphas_{last}"][0])) < 1e-5
    assert int(g[f"flags_{last}"][1]) == 1                       # the fixture ends with a tail chunk
    assert st["start_idx"] == int(g[f"flags_{last}"][2])


def test_streaming_frontend_features_match_reference():
    g, cfg, sd, wav, cmvn = load()
    n1 = int(g["n1"])
    fc = S.frontend_init()
    stride = 9600
    audio = wav[:n1]
    k = 0
    for i in range(len(audio) // stride):
        f = S.frontend_step(audio[i * stride:(i + 1) * stride], fc, cmvn, False)
        ref = torch.from_numpy(g[f"feats_{k}"])[0]
        assert f.shape == ref.shape and torch.equal(f, ref), k
        k += 1
    audio = torch.cat((audio[(len(audio) // stride) * stride:], wav[n1:]))
    n = len(audio) // stride + 1
    for i in range(n):
        piece = audio[i * stride:(i + 1) * stride]
        fin = i == n - 1
        if fin and len(piece) < 960:
            break
        f = S.frontend_step(piece, fc, cmvn, fin)
        ref = torch.from_numpy(g[f"feats_{k}"])[0]
        assert f.shape == ref.shape and torch.equal(f, ref), k
        k += 1


def test_streaming_oracle_matches_reference_in_other_chunk_geometries():
    """chunk sizes / look-back settings other than the main fixture's, from the reference's own ParaformerStreaming.inference
    on the same clip and weights (oracle/make_golden_streaming.py --geometries): the oracle, driven chunk by chunk with the
    recorded online features, returns the reference's token ids and position counter on every chunk incl. the tail chunk"""
    import json
    import numpy as np
    import os
    g, cfg, sd, wav, cmvn = load()
    gg = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "streaming_geometries.npz"), allow_pickle=False)
    sessions = json.loads(str(gg["sessions"]))
    assert [s["chunk"] for s in sessions] == [[5, 10, 5], [0, 8, 4], [0, 10, 5], [0, 20, 10], [0, 16, 8]]
    for si, s in enumerate(sessions):
        st = S.model_init(cfg, tuple(s["chunk"]), s["enc_lb"], s["dec_lb"])
        for i in range(s["n_chunks"]):
            fin, tail, start_idx = (int(v) for v in gg[f"s{si}_flags_{i}"])
            if tail:
                st["tail_chunk"] = True
                feats = st["feats"]
            else:
                feats = torch.from_numpy(gg[f"s{si}_feats_{i}"])
            with torch.no_grad():
                ids = S.generate_chunk(feats, st, sd, cfg, bool(fin))
            assert ids == gg[f"s{si}_tokens_{i}"].tolist(), (s, i)
            assert st["start_idx"] == start_idx, (s, i)


def test_streaming_oracle_reproduces_the_reference_when_chunk_right_is_zero():
    """chunk_size[2] == 0 (found by oracle/fuzz_streaming_vs_reference.py): the reference's K/V stride is k_h[:, :, :-(0)] = EMPTY
    (sanm/attention.py:345-346), so nothing is cached and the encoder look-back has no effect ([5, 11, 0], look-back 3); with
    chunk_size[0] == 0 as well its overlap window x[:, -0:] is the WHOLE history ([0, 12, 0]: 43 tokens from the final 9-frame
    chunk). Reference sessions: tests/golden/streaming_right0.npz (make_golden_streaming.py --right0)."""
    import json
    import numpy as np
    import os
    g, cfg, sd, wav, cmvn = load()
    gg = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "streaming_right0.npz"), allow_pickle=False)
    sessions = json.loads(str(gg["sessions"]))
    assert [s["chunk"] for s in sessions] == [[5, 11, 0], [0, 12, 0]]
    for si, s in enumerate(sessions):
        st = S.model_init(cfg, tuple(s["chunk"]), s["enc_lb"], s["dec_lb"])
        for i in range(s["n_chunks"]):
            fin, tail, start_idx = (int(v) for v in gg[f"s{si}_flags_{i}"])
            assert not tail
            trace = []
            with torch.no_grad():
                ids = S.generate_chunk(torch.from_numpy(gg[f"s{si}_feats_{i}"]), st, sd, cfg, bool(fin), trace)
            assert ids == gg[f"s{si}_tokens_{i}"].tolist(), (s, i)
            assert st["start_idx"] == start_idx, (s, i)
            assert (trace[0]["enc"] - torch.from_numpy(gg[f"s{si}_enc_{i}"])).abs().max().item() < 1e-4, (s, i)
    assert len(gg["s1_tokens_6"]) == 43 and gg["s1_feats_6"].shape[1] == 9          # the growing-window quirk at work


def test_stream_batch_refuses_the_geometry_whose_reference_window_is_the_whole_history():
    from funasr_amd.paraformer_streaming import ParaformerStreaming, StreamBatch
    g, cfg, sd, wav, cmvn = load()
    model = ParaformerStreaming.from_config(cfg)
    with pytest.raises(ValueError, match="chunk_size"):
        StreamBatch(model, 1, [0, 12, 0], 2, 1)
