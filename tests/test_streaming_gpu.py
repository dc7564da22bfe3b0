"""Streaming (chunked) Paraformer through the C ABI (pf_stream_*, pf_frontend_fbank / pf_frontend_lfr_cmvn) against the
golden fixture recorded from the REFERENCE's own ParaformerStreaming + WavFrontendOnline (tests/golden/streaming.npz,
oracle/make_golden_streaming.py): per 600 ms chunk the online features, the encoder window, the carried CIF state and
the token ids; incl. the final flush and the < 960-sample tail chunk. Token ids / counts bit-exact, activations 1e-3."""
import json
import os

import numpy as np
import pytest
import torch

from funasr_amd import synth

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load():
    g = np.load(os.path.join(GOLD, "streaming.npz"), allow_pickle=False)
    cfg = json.loads(bytes(g["config"]).decode())
    sd = synth.paraformer_state_dict(cfg, seed=int(g["seed"]), cif_bias=float(g["cif_bias"]))
    wav = torch.from_numpy(g["pcm"].astype(np.float32) / 32768.0)
    return g, cfg, sd, wav


def build(cfg, sd, dev):
    from funasr_amd.paraformer_streaming import ParaformerStreaming
    model = ParaformerStreaming.from_config(cfg)
    model.load_state_dict(sd, strict=False)
    return model.to(dev)


def frontend(dev):
    from funasr_amd.paraformer_streaming import WavFrontendOnline
    return WavFrontendOnline(cmvn_file=os.path.join(GOLD, "am.mvn"), lfr_m=7, lfr_n=6, dither=0.0, device=dev)


def test_online_frontend_matches_reference_chunks(cuda):
    g, cfg, sd, wav = load()
    fe = frontend(cuda)
    n1, stride = int(g["n1"]), 9600
    cache = {}
    k = 0
    audio = wav[:n1]
    for i in range(len(audio) // stride):
        f, lens = fe(audio[i * stride:(i + 1) * stride][None], None, cache=cache, is_final=False)
        ref = torch.from_numpy(g[f"feats_{k}"])
        assert tuple(f.shape) == tuple(ref.shape) and int(lens[0]) == ref.shape[1]
        assert (f.cpu() - ref).abs().max().item() < 5e-4, k
        k += 1
    audio = torch.cat((audio[(len(audio) // stride) * stride:], wav[n1:]))
    n = len(audio) // stride + 1
    for i in range(n):
        piece = audio[i * stride:(i + 1) * stride]
        fin = i == n - 1
        if fin and len(piece) < 960:
            break
        f, lens = fe(piece[None], None, cache=cache, is_final=fin)
        ref = torch.from_numpy(g[f"feats_{k}"])
        assert tuple(f.shape) == tuple(ref.shape)
        assert (f.cpu() - ref).abs().max().item() < 5e-4, k
        k += 1
    assert k == int(g["n_chunks"]) - 1                      # every chunk but the tail one goes through the frontend


@pytest.mark.parametrize("use_graph", [False, True])
def test_stream_steps_match_reference_given_reference_features(cuda, use_graph):
    from funasr_amd.paraformer_streaming import StreamBatch
    g, cfg, sd, wav = load()
    model = build(cfg, sd, cuda)
    sb = StreamBatch(model, 1, [0, 10, 5], 4, 1, use_graph=use_graph)
    for i in range(int(g["n_chunks"])):
        fin, tail, start_idx = (int(v) for v in g[f"flags_{i}"])
        feats = None if tail else torch.from_numpy(g[f"feats_{i}"]).to(cuda)
        ids, enc = sb.step(feats, is_final=bool(fin), tail_chunk=bool(tail), return_enc=True)
        ref = torch.from_numpy(g[f"enc_{i}"])
        assert tuple(enc.shape) == tuple(ref.shape), i
        err = (enc.cpu() - ref).abs().max().item()
        assert err < 1e-3, (i, err)
        got = [t for t in ids[0] if t not in (0, 1, 2)]
        assert got == g[f"tokens_{i}"].tolist(), (i, got, g[f"tokens_{i}"].tolist())
        st = sb.peek()
        assert st["start_idx"] == start_idx
        assert abs(st["cif_alphas"][0] - float(g[f"cif_alphas_{i}"][0])) < 1e-4, i
        assert (st["cif_hidden"][0] - torch.from_numpy(g[f"cif_hidden_{i}"])).abs().max().item() < 1e-3, i
    sb.close()


def test_graph_replay_equals_eager_bitwise_and_streams_are_independent(cuda):
    from funasr_amd.paraformer_streaming import StreamBatch
    g, cfg, sd, wav = load()
    model = build(cfg, sd, cuda)
    gen = torch.Generator().manual_seed(5)
    runs = {}
    for name, S, graph in (("eager1", 1, False), ("graph1", 1, True), ("graph3", 3, True)):
        sb = StreamBatch(model, S, [0, 10, 5], 4, 1, use_graph=graph)
        out = []
        for rep in range(2):                                 # second pass after reset(): state fully cleared
            gen.manual_seed(5)
            for i in range(int(g["n_chunks"])):
                fin, tail, _ = (int(v) for v in g[f"flags_{i}"])
                feats = None
                if not tail:
                    f0 = torch.from_numpy(g[f"feats_{i}"])
                    others = [f0 * 0.5 + 0.1 * torch.randn(f0.shape, generator=gen) for _ in range(S - 1)]
                    feats = torch.cat([f0] + others, 0).to(cuda)
                ids, enc = sb.step(feats, is_final=bool(fin), tail_chunk=bool(tail), return_enc=True)
                out.append((rep, ids[0], enc[0].cpu()))
            sb.reset()
        runs[name] = out
        sb.close()
    n = int(g["n_chunks"])
    for name in ("graph1", "graph3"):
        for a, b in zip(runs["eager1"], runs[name]):
            assert a[1] == b[1], name
            assert torch.equal(a[2], b[2]), name
    for a, b in zip(runs["eager1"][:n], runs["eager1"][n:]):  # reset() reproduces the session
        assert a[1] == b[1] and torch.equal(a[2], b[2])


def test_streaming_inference_api_two_calls_equals_reference_tokens(cuda):
    g, cfg, sd, wav = load()
    model = build(cfg, sd, cuda)
    fe = frontend(cuda)
    n1 = int(g["n1"])
    cache = {}
    kw = dict(chunk_size=[0, 10, 5], encoder_chunk_look_back=4, decoder_chunk_look_back=1)
    r1, meta = model.inference([wav[:n1]], key=["utt"], tokenizer=None, frontend=fe, cache=cache, is_final=False, **kw)
    assert "batch_data_time" in meta
    r2, _ = model.inference([wav[n1:]], key=["utt"], tokenizer=None, frontend=fe, cache=cache, is_final=True, **kw)
    ref = [g[f"tokens_{i}"].tolist() for i in range(int(g["n_chunks"]))]
    assert r1[0]["token_int"] == sum(ref[:5], [])
    assert r2[0]["token_int"] == sum(ref[5:], [])
    # the final call re-initialised the cache: the same audio again gives the same tokens
    r3, _ = model.inference([wav], key=["utt"], tokenizer=None, frontend=fe, cache=cache, is_final=True, **kw)
    assert r3[0]["token_int"] == sum(ref, [])


def test_streaming_inference_on_a_wav_path_inside_a_list_is_a_whole_utterance(cuda, tmp_path):
    """AutoModel.inference always hands `data_in` over as a list (auto_model.py:796-812): a file PATH in that list is a
    complete recording, so the look-ahead flush and the tail chunk must run without is_final being passed
    (paraformer_streaming/model.py:692-701) -- the trailing tokens of the reference session are all there."""
    import wave
    g, cfg, sd, wav = load()
    model = build(cfg, sd, cuda)
    fe = frontend(cuda)
    path = str(tmp_path / "utt.wav")
    with wave.open(path, "wb") as w:
        w.setnchannels(1); w.setsampwidth(2); w.setframerate(16000)
        w.writeframes(g["pcm"].astype("<i2").tobytes())
    kw = dict(chunk_size=[0, 10, 5], encoder_chunk_look_back=4, decoder_chunk_look_back=1)
    ref = sum((g[f"tokens_{i}"].tolist() for i in range(int(g["n_chunks"]))), [])
    for data_in in ([path], path):
        r, _ = model.inference(data_in, key=["utt"], tokenizer=None, frontend=fe, cache={}, **kw)
        assert r[0]["token_int"] == ref


@pytest.mark.parametrize("use_graph", [False, True])
def test_stream_other_chunk_geometries_match_reference_sessions(cuda, use_graph):
    """the reference's own ParaformerStreaming.inference at other chunk sizes / look-back settings (same clip and weights,
    tests/golden/streaming_geometries.npz): token ids and the position counter on every chunk incl. the tail chunk"""
    import json
    import numpy as np
    import os
    from funasr_amd.paraformer_streaming import StreamBatch
    g, cfg, sd, wav = load()
    model = build(cfg, sd, cuda)
    gg = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "streaming_geometries.npz"), allow_pickle=False)
    for si, s in enumerate(json.loads(str(gg["sessions"]))):
        sb = StreamBatch(model, 1, s["chunk"], s["enc_lb"], s["dec_lb"], use_graph=use_graph)
        for i in range(s["n_chunks"]):
            fin, tail, start_idx = (int(v) for v in gg[f"s{si}_flags_{i}"])
            feats = None if tail else torch.from_numpy(gg[f"s{si}_feats_{i}"]).to(cuda)
            ids = sb.step(feats, is_final=bool(fin), tail_chunk=bool(tail))
            got = [t for t in ids[0] if t not in (0, 1, 2)]
            assert got == gg[f"s{si}_tokens_{i}"].tolist(), (s, i, got)
            assert sb.peek()["start_idx"] == start_idx
        sb.close()


@pytest.mark.parametrize("chunk,enc_lb,dec_lb", [([5, 10, 5], 2, 2), ([0, 8, 4], 1, 0), ([0, 10, 5], 0, 1),
                                                 ([0, 20, 10], 2, 1), ([0, 16, 8], 1, 1)])      # > 24 possible fires per step
def test_stream_other_chunk_geometries_vs_streaming_oracle(cuda, chunk, enc_lb, dec_lb):
    """chunk_size / look-back settings other than the golden session's, against the (reference-pinned) streaming oracle
    on random online features: token ids and counts equal, encoder window within 1e-3."""
    from funasr_amd.paraformer_streaming import StreamBatch
    from oracle import streaming_oracle as S
    g, cfg, sd, wav = load()
    model = build(cfg, sd, cuda)
    sb = StreamBatch(model, 1, chunk, enc_lb, dec_lb)
    st = S.model_init(cfg, tuple(chunk), enc_lb, dec_lb)
    gen = torch.Generator().manual_seed(chunk[1] * 10 + enc_lb)
    for i in range(6):
        fin = i == 5
        n = chunk[1] if not fin else chunk[1] + 2           # the final flush of the online frontend brings extra frames
        feats = torch.randn(1, n, 560, generator=gen) * 0.7
        trace = []
        with torch.no_grad():
            oids = S.generate_chunk(feats.clone(), st, sd, cfg, fin, trace)
        ids, enc = sb.step(feats.to(cuda), is_final=fin, return_enc=True)
        assert (enc.cpu() - trace[0]["enc"]).abs().max().item() < 1e-3, i
        assert [t for t in ids[0] if t not in (0, 1, 2)] == oids, (i, ids[0], oids)
        assert len(ids[0]) == trace[0]["n"], i
    sb.close()


@pytest.mark.parametrize("precision", ["fp32", "f16x2"])
@pytest.mark.parametrize("chunk,enc_lb,sizes", [([5, 10, 5], 1, [10, 10, 10, 10, 7]),      # the first cache (15 rows) exceeds the trim (10)
                                                ([5, 10, 5], 1, [10, 3, 10, 2, 10, 12]),   # ... and short chunks follow it (API-level session)
                                                ([6, 4, 2], 1, [4, 4, 1, 4, 3]),           # the first cache is more than twice the trim
                                                ([3, 8, 4], 2, [8, 8, 8, 5])])             # fits: the plain ring
def test_stream_first_chunk_cache_is_not_trimmed(cuda, precision, chunk, enc_lb, sizes):
    """The reference trims the encoder's K / V cache to look_back * chunk_size[1] rows from the SECOND chunk on and leaves the first
    chunk's cache (chunk_size[0] + n rows) untrimmed (sanm/attention.py:353-361): with chunk_size[0] > (look_back - 1) * chunk_size[1]
    the second chunk attends to more cached rows than any later one. Found by tools/fuzz_gpu_streaming_vs_oracle.py (the ring used to drop
    those rows: encoder window off by 0.2 in the second chunk); against the reference-pinned streaming oracle."""
    from funasr_amd.paraformer_streaming import StreamBatch
    from oracle import streaming_oracle as S
    g, cfg, sd, wav = load()
    model = build(cfg, sd, cuda)
    sb = StreamBatch(model, 1, chunk, enc_lb, 1, precision=precision)
    st = S.model_init(cfg, tuple(chunk), enc_lb, 1)
    gen = torch.Generator().manual_seed(sum(sizes) + enc_lb)
    for i, n in enumerate(sizes):
        fin = i == len(sizes) - 1
        feats = torch.randn(1, n, 560, generator=gen) * 0.7
        trace = []
        with torch.no_grad():
            oids = S.generate_chunk(feats.clone(), st, sd, cfg, fin, trace)
        ids, enc = sb.step(feats.to(cuda), is_final=fin, return_enc=True)
        assert (enc.cpu() - trace[0]["enc"]).abs().max().item() < 1e-3, (i, n)
        assert [t for t in ids[0] if t not in (0, 1, 2)] == oids and len(ids[0]) == trace[0]["n"], (i, ids[0], oids)
    sb.close()


def test_graphs_survive_workspace_growth_by_an_offline_batch(cuda):
    """The captured step holds raw workspace pointers of the shared encoder/decoder handles; an offline batch that grows
    those workspaces in between must not leave the graph with stale pointers (it is re-captured)."""
    from funasr_amd.paraformer_streaming import StreamBatch
    g, cfg, sd, wav = load()
    model = build(cfg, sd, cuda)
    sb = StreamBatch(model, 1, [0, 10, 5], 4, 1, use_graph=True)
    ref = StreamBatch(model, 1, [0, 10, 5], 4, 1, use_graph=False)
    gen = torch.Generator().manual_seed(0)
    for i in range(6):
        f = torch.from_numpy(g[f"feats_{i}"]).to(cuda)
        a, ea = sb.step(f, return_enc=True)
        b, eb = ref.step(f, return_enc=True)
        assert a == b and torch.equal(ea, eb), i
        if i == 2:      # a big offline batch through the same handles: every workspace is re-allocated
            x = torch.randn(4, 300, 560, generator=gen).to(cuda)
            model.recognize_features(x, [300, 250, 100, 280])
    sb.close()
    ref.close()


@pytest.mark.parametrize("precision", ["fp32", "f16x2"])
def test_stream_chunk_right_zero_matches_reference_session(cuda, precision):
    """chunk_size [5, 11, 0] with encoder look-back 3: the reference's K/V stride `[: -chunk_size[2]]` is empty there, so it
    never caches anything (sanm/attention.py:345-346) -- the reference's own session (tests/golden/streaming_right0.npz):
    token ids and position counter on every chunk, encoder window within 1e-3"""
    from funasr_amd.paraformer_streaming import StreamBatch
    g, cfg, sd, wav = load()
    model = build(cfg, sd, cuda)
    gg = np.load(os.path.join(GOLD, "streaming_right0.npz"), allow_pickle=False)
    s = json.loads(str(gg["sessions"]))[0]
    assert s["chunk"] == [5, 11, 0]
    sb = StreamBatch(model, 1, s["chunk"], s["enc_lb"], s["dec_lb"], precision=precision)
    for i in range(s["n_chunks"]):
        fin, tail, start_idx = (int(v) for v in gg[f"s0_flags_{i}"])
        ids, enc = sb.step(torch.from_numpy(gg[f"s0_feats_{i}"]).to(cuda), is_final=bool(fin), return_enc=True)
        assert [t for t in ids[0] if t not in (0, 1, 2)] == gg[f"s0_tokens_{i}"].tolist(), i
        assert sb.peek()["start_idx"] == start_idx
        assert (enc.cpu() - torch.from_numpy(gg[f"s0_enc_{i}"])).abs().max().item() < 1e-3, i
    sb.close()


def test_step_launch_fusions_leave_the_session_unchanged(cuda):
    """pf_stream_set_option "fsmn_rides" (the encoder's FSMN memory inside the attention launch) and "kv_batched" (the decoder's
    sixteen key/value projections as one launch) and "wide_k" (four workgroups per tile for the long-K projections) are re-arrangements of the same arithmetic: bitwise equal encoder rows and the same
    tokens. "ln_carry" (LayerNorms applied on the fetch of the next small-M GEMM from block partials the GEMM before left) replaces a
    two-pass variance by a one-pass one: fp32-class agreement (1e-4 on rows of magnitude ~1) and the same tokens."""
    from funasr_amd.paraformer_streaming import StreamBatch
    g, cfg, sd, wav = load()
    model = build(cfg, sd, cuda)

    def session(opts, S=1, precision="fp32"):
        sb = StreamBatch(model, S, [0, 10, 5], 4, 1, use_graph=True, precision=precision)
        for k, v in opts.items():
            sb.set_option(k, v)
        out = []
        for i in range(int(g["n_chunks"])):
            fin, tail, _ = (int(v) for v in g[f"flags_{i}"])
            feats = None if tail else torch.from_numpy(g[f"feats_{i}"]).repeat(S, 1, 1).to(cuda)
            ids, enc = sb.step(feats, is_final=bool(fin), tail_chunk=bool(tail), return_enc=True)
            out.append((ids[0], enc[0].cpu()))
        sb.close()
        return out

    for S in (1, 3):
        plain = session({"ln_carry": 0, "fsmn_rides": 0, "kv_batched": 0, "wide_k": 0}, S)
        for opts in ({"ln_carry": 0, "fsmn_rides": 1, "kv_batched": 0, "wide_k": 0}, {"ln_carry": 0, "fsmn_rides": 0, "kv_batched": 1, "wide_k": 0},
                     {"ln_carry": 0, "fsmn_rides": 0, "kv_batched": 0, "wide_k": 1}):
            for a, b in zip(plain, session(opts, S)):
                assert a[0] == b[0] and torch.equal(a[1], b[1]), (S, opts)
        for a, b in zip(plain, session({"ln_carry": 1, "fsmn_rides": 1, "kv_batched": 1, "wide_k": 1}, S)):
            assert a[0] == b[0], S
            assert (a[1] - b[1]).abs().max().item() < 1e-4, S
        # the f16x2 step: "ln_folded" (LayerNorms in the second launch of the split-K projections, attention writing the operand
        # planes of the out-projection) and the FSMN rider repeat the separate launches' arithmetic: the same bits
        plain2 = session({"ln_folded": 0, "fsmn_rides": 0, "short_k": 0}, S, "f16x2")
        for a, b in zip(plain2, session({"ln_folded": 1, "fsmn_rides": 1, "short_k": 0}, S, "f16x2")):
            assert a[0] == b[0] and torch.equal(a[1], b[1]), S
        # "short_k" 3 (the default of small handles): linear_out in the four-slice split-K form, norm2 in its second launch -- another
        # summation order, fp32-class: the same tokens, encoder rows within 1e-4
        for a, b in zip(plain2, session({}, S, "f16x2")):
            assert a[0] == b[0], S
            assert (a[1] - b[1]).abs().max().item() < 1e-4, S


def test_two_handles_with_a_step_in_flight_each(cuda):
    """pf_stream_step_begin / _end: two StreamBatches over two model objects (own handles, own workspaces, own HIP streams) run
    their steps concurrently; each returns what it returns alone. A second begin on a handle before its end is refused."""
    from funasr_amd.paraformer_streaming import StreamBatch
    g, cfg, sd, wav = load()
    m1, m2 = build(cfg, sd, cuda), build(cfg, sd, cuda)
    alone = StreamBatch(m1, 2, [0, 10, 5], 4, 1, use_graph=True)
    feats = [torch.from_numpy(g[f"feats_{i}"]) for i in range(int(g["n_chunks"])) if not int(g[f"flags_{i}"][1])]
    pair = [torch.cat([f, 0.5 * f], 0).to(cuda) for f in feats]
    ref = [alone.step(f, return_enc=True) for f in pair]
    alone.close()
    a, b = StreamBatch(m1, 2, [0, 10, 5], 4, 1, use_graph=True), StreamBatch(m2, 2, [0, 10, 5], 4, 1, use_graph=True)
    for i, f in enumerate(pair):
        a.step_begin(f, return_enc=True)
        b.step_begin(f, return_enc=True)
        if i == 0:
            with pytest.raises(RuntimeError):
                a.step_begin(f)
        ra, rb = a.step_end(), b.step_end()
        for r in (ra, rb):
            assert r[0] == ref[i][0] and torch.equal(r[1], ref[i][1]), i
    a.close()
    b.close()
