"""Module-level parity of the HIP path (through the C ABI) against
  (a) the committed golden fixtures produced by the REFERENCE's own modules (tests/golden, oracle/make_golden.py) and
  (b) the CPU oracle (oracle/paraformer_oracle.py) on further seeded inputs.

Bars (north_star): integer results -- CIF fire positions, token counts, arg-max token ids -- bit-exact;
encoder / decoder activations within 1e-3 absolute of the fp32 CPU path (we assert 2e-4 or tighter where the
measured margin allows); CIF scan bit-exact given identical alphas.
"""
import json
import os

import numpy as np
import pytest
import torch

from funasr_amd import synth

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def gold(name):
    return np.load(os.path.join(GOLD, name + ".npz"), allow_pickle=False)


def t(a):
    return torch.from_numpy(np.asarray(a))


def fires_of(peaks):
    return (torch.floor(torch.as_tensor(peaks)) >= 1)


@pytest.fixture(params=["fp32", "bf16x3", "f16x2"])
def f32_mode(request):
    """the three fp32-accurate GEMM routes: v_mfma_f32_32x32x2_f32 (gemm_f32.hip), three-bf16-plane operands with six bf16
    MFMA products (gemm_split3.hip), two-fp16-plane operands with three fp16 MFMA products (gemm_f16x2.hip, which also
    moves the encoder's self-attention to attention_f16x2.hip). All must meet every fp32 parity bar."""
    return request.param


# ---------------------------------------------------------------------------------------------------- frontend
def test_frontend_vs_reference_features_and_oracle(cuda):
    from funasr_amd.wav_frontend import WavFrontend
    from oracle import paraformer_oracle as O
    g = gold("frontend")
    cmvn = t(g["cmvn"])
    fe = WavFrontend(cmvn=cmvn, lfr_m=7, lfr_n=6, dither=0.0)
    waves = [t(g["pcm_a"].astype(np.float32) / 32768.0), t(g["pcm_b"].astype(np.float32) / 32768.0)]
    lens = [w.numel() for w in waves]
    batch = torch.zeros(2, max(lens))
    for i, w in enumerate(waves):
        batch[i, : lens[i]] = w
    feats, flens, fb = fe(batch.to(cuda), lens, return_fbank=True)
    feats, fb = feats.cpu(), fb.cpu()
    assert flens.tolist() == [g["feats_a"].shape[0], g["feats_b"].shape[0]]
    # (a) reference features (kaldi-native-fbank + reference LFR/CMVN): float64-FFT vs float32-FFT level
    for i, k in enumerate("ab"):
        ref_fb = t(g[f"fbank_knf_{k}"])
        d = (fb[i, : ref_fb.shape[0]] - ref_fb).abs()
        assert d.max().item() <= 2e-3 and d.mean().item() <= 2e-5, (d.max().item(), d.mean().item())
        assert (feats[i, : flens[i]] - t(g[f"feats_{k}"])).abs().max().item() < 5e-4
    assert (feats[1, flens[1]:] == 0).all()            # pad_sequence zeros
    # (b) the oracle (torchaudio semantics in float32): same tables, only FFT / reduction order differ
    of, ol, ofb = O.wav_frontend(waves, cmvn, return_fbank=True)
    assert ol.tolist() == flens.tolist()
    for i in range(2):
        assert (fb[i, : ofb[i].shape[0]] - ofb[i]).abs().max().item() <= 2e-3
        assert (feats[i, : ol[i]] - of[i, : ol[i]]).abs().max().item() < 5e-4


@pytest.mark.parametrize("window,snip", [("povey", True), ("hanning", False), ("rectangular", True), ("blackman", False),
                                         ("hamming", False)])
def test_frontend_window_and_snip_edges_options(cuda, window, snip):
    """WavFrontend(window=, snip_edges=) -> kaldi.fbank(window_type=, snip_edges=) (wav_frontend.py:178-180): frame counts, the
    mirrored ends and the window against the reference-vendored kaldi-native-fbank run with the same options
    (tests/golden/fbank_options.npz) and against the oracle; a ragged batch, so the mirror uses each clip's own length."""
    from funasr_amd.wav_frontend import WavFrontend
    from oracle import paraformer_oracle as O
    g, go = gold("frontend"), gold("fbank_options")
    fe = WavFrontend(cmvn=None, lfr_m=1, lfr_n=1, dither=0.0, window=window, snip_edges=snip)
    waves = [t(g["pcm_a"].astype(np.float32) / 32768.0), t(g["pcm_b"].astype(np.float32) / 32768.0)]
    lens = [w.numel() for w in waves]
    batch = torch.zeros(2, max(lens))
    for i, w in enumerate(waves):
        batch[i, : lens[i]] = w
    feats, flens, fb = fe(batch.to(cuda), lens, return_fbank=True)
    fb = fb.cpu()
    bar = 5e-3 if window == "rectangular" else 2e-3
    for i, k in enumerate("ab"):
        ref = t(go[f"{k}_{window}_{'snip' if snip else 'nosnip'}"])
        assert int(flens[i]) == ref.shape[0] == fe.num_fbank_frames(lens[i])
        d = (fb[i, : ref.shape[0]] - ref).abs()
        assert d.max().item() <= bar and d.mean().item() <= 3e-5, (d.max().item(), d.mean().item())
        ofb = O.kaldi_fbank(waves[i] * 32768.0, window_type=window, snip_edges=snip)
        assert (fb[i, : ref.shape[0]] - ofb).abs().max().item() <= bar
        assert torch.equal(feats[i, : ref.shape[0]].cpu(), fb[i, : ref.shape[0]])          # lfr 1 / 1, no CMVN: the log-mel itself
    if not snip:        # fewer samples than 25 ms: the CLASS shrinks the window to the clip (wav_frontend.py:176), two mirrored frames
        short = t(go["short_pcm"].astype(np.float32) / 32768.0)
        _, sl, sfb = fe(short[None].to(cuda), [short.numel()], return_fbank=True)
        ofb = O.kaldi_fbank(short * 32768.0, 80, O.short_clip_frame_length_ms(short.numel()), window_type=window, snip_edges=False)
        assert int(sl[0]) == ofb.shape[0] == 2 and (sfb[0].cpu() - ofb).abs().max().item() <= bar


def test_frontend_clips_shorter_than_one_window(cuda):
    """wav_frontend.py:176: a clip shorter than 25 ms (a few-millisecond VAD segment) is analysed with one window of its own length, the
    FFT size following the window; in a batch only that clip is affected. Against the reference-vendored kaldi-native-fbank run with
    frame_length_ms = n / 16 (tests/golden/fbank_options.npz) and against the oracle's batch-level restatement, LFR + CMVN included."""
    from funasr_amd.wav_frontend import WavFrontend
    from oracle import paraformer_oracle as O
    g, go = gold("frontend"), gold("fbank_options")
    cmvn = t(g["cmvn"])
    fe = WavFrontend(cmvn=cmvn, lfr_m=7, lfr_n=6, dither=0.0)
    tiny = (399, 300, 257, 256, 200, 129, 100, 64, 33)
    waves = [t(go[f"tiny_{n}_pcm"].astype(np.float32) / 32768.0) for n in tiny] + [t(g["pcm_b"].astype(np.float32) / 32768.0)]
    lens = [w.numel() for w in waves]
    batch = torch.zeros(len(waves), max(lens))
    for i, w in enumerate(waves):
        batch[i, : lens[i]] = w
    feats, flens, fb = fe(batch.to(cuda), lens, return_fbank=True)
    feats, fb = feats.cpu(), fb.cpu()
    of, ol, ofb = O.wav_frontend(waves, cmvn, return_fbank=True)
    assert flens.tolist() == ol.tolist() == [1] * len(tiny) + [g["feats_b"].shape[0]]
    for i, n in enumerate(tiny):
        assert (fb[i, :1] - t(go[f"tiny_{n}_hamming_snip"])).abs().max().item() <= 2e-3, n
        assert (fb[i, 1:] == 0).all() and (feats[i, 1:] == 0).all()
    for i in range(len(waves)):
        assert (feats[i, : ol[i]] - of[i, : ol[i]]).abs().max().item() < 5e-4, lens[i]
    # the long clip of the mixed batch == the same clip alone, bitwise (utterance-level independence)
    alone, al = fe(waves[-1][None].to(cuda), [lens[-1]])
    assert torch.equal(alone[0, : al[0]].cpu(), feats[-1, : flens[-1]])
    assert (feats[-1, : flens[-1]] - t(g["feats_b"])).abs().max().item() < 5e-4
    with pytest.raises(ValueError, match="no analysis window"):
        fe(torch.zeros(1, 8).to(cuda), [1])


def test_frontend_window_by_name_at_the_c_abi(cuda):
    """pf_frontend_set_window evaluates the named window as kaldi-native-fbank does (float64 cosines): against the same goldens."""
    import ctypes as C
    from funasr_amd import _lib
    g, go = gold("frontend"), gold("fbank_options")
    lib = _lib.load()
    cfg = _lib.pf_frontend_config(16000, 400, 160, 80, 1, 1, 20.0, 0.0, 0.97, 32768.0)
    wave = t(g["pcm_b"].astype(np.float32) / 32768.0).to(cuda)
    with torch.cuda.device(cuda):
        h = _lib.check_handle(lib.pf_frontend_create(C.byref(cfg)), "pf_frontend_create")
        try:
            assert lib.pf_frontend_set_window(h, b"triangle", 0.42) != 0
            for window in ("povey", "blackman", "hanning", "rectangular", "hamming"):
                _lib.check(lib.pf_frontend_set_window(h, window.encode(), 0.42), "pf_frontend_set_window")
                for snip in (1, 0):
                    _lib.check(lib.pf_frontend_set_snip_edges(h, snip), "pf_frontend_set_snip_edges")
                    n = lib.pf_frontend_num_fbank_frames(h, wave.numel())
                    ref = t(g["fbank_knf_b"]) if (window == "hamming" and snip) else t(go[f"b_{window}_{'snip' if snip else 'nosnip'}"])
                    assert n == ref.shape[0]
                    out = torch.empty(n, 80, device=cuda)
                    _lib.check(lib.pf_frontend_fbank(h, wave.data_ptr(), wave.numel(), out.data_ptr(), None), "pf_frontend_fbank")
                    torch.cuda.synchronize()
                    d = (out.cpu() - ref).abs()
                    assert d.max().item() <= (5e-3 if window == "rectangular" else 2e-3), (window, snip, d.max().item())
            # fewer samples than the handle's window, snip_edges = 0: the ends mirror more than once (feature-window.cc:152-171)
            short = t(go["short_pcm"].astype(np.float32) / 32768.0).to(cuda)
            assert lib.pf_frontend_num_fbank_frames(h, short.numel()) == 2
            out = torch.empty(2, 80, device=cuda)
            _lib.check(lib.pf_frontend_fbank(h, short.data_ptr(), short.numel(), out.data_ptr(), None), "pf_frontend_fbank")
            torch.cuda.synchronize()
            assert (out.cpu() - t(go["short_hamming_nosnip"])).abs().max().item() <= 2e-3
        finally:
            lib.pf_frontend_destroy(h)


def test_frontend_ragged_batch_equals_single_utterance_bitwise(cuda):
    """Utterance-level data parallelism: a clip's features do not depend on its batch neighbours."""
    from funasr_amd.wav_frontend import WavFrontend
    sh, sc = synth.synthetic_cmvn()
    fe = WavFrontend(cmvn=torch.stack([sh, sc]), lfr_m=7, lfr_n=6, dither=0.0)
    lens = [16000, 7777, 400, 30011]
    waves = [synth.speech_like(n, seed=i) for i, n in enumerate(lens)]
    batch = torch.zeros(len(lens), max(lens))
    for i, w in enumerate(waves):
        batch[i, : lens[i]] = w
    feats, flens = fe(batch.to(cuda), lens)
    assert flens.tolist() == [fe.num_frames(n) for n in lens]
    for i, w in enumerate(waves):
        single, sl = fe(w[None].to(cuda), [lens[i]])
        assert torch.equal(single[0, : sl[0]].cpu(), feats[i, : flens[i]].cpu())
        assert (feats[i, flens[i]:] == 0).all()


# ----------------------------------------------------------------------------------------------------- encoder
def _encoder(cfg, sd, cuda, cls=None):
    from funasr_amd.sanm_encoder import SANMEncoder
    cls = cls or SANMEncoder
    kw = dict(cfg)
    enc = cls(input_layer="pe", **kw)
    enc.load_state_dict(sd, strict=True)
    return enc.to(cuda)


def test_encoder_vs_reference_golden(cuda, f32_mode):
    g = gold("encoder")
    cfg = json.loads(str(g["cfg"]))
    enc = _encoder(cfg, synth.encoder_state_dict(cfg, seed=int(g["seed"])), cuda).set_precision(f32_mode)
    out, olens, _ = enc(t(g["xs"]).to(cuda), t(g["lens"]))
    assert olens.tolist() == g["olens"].tolist()
    b1, _ = enc._run(t(g["xs"]).to(cuda), t(g["lens"]), run_blocks=1)
    b3, _ = enc._run(t(g["xs"]).to(cuda), t(g["lens"]), run_blocks=3)
    assert (b1.cpu() - t(g["block1"])).abs().max().item() < 1e-4
    assert (b3.cpu() - t(g["block3"])).abs().max().item() < 1e-4
    assert (out.cpu() - t(g["out"])).abs().max().item() < 1e-4     # north_star bar: 1e-3


def test_encoder_full_depth_vs_oracle(cuda, f32_mode):
    """All 50 blocks of Paraformer-large, ragged batch, every (also padded) frame compared."""
    from oracle import paraformer_oracle as O
    cfg = synth.PARAFORMER_LARGE["encoder"]
    sd = synth.encoder_state_dict(cfg, seed=3)
    enc = _encoder(cfg, sd, cuda).set_precision(f32_mode)
    g = torch.Generator().manual_seed(77)
    xs = torch.randn(3, 90, 560, generator=g) * 0.7
    lens = torch.tensor([90, 61, 8], dtype=torch.int32)
    for b in range(3):
        xs[b, lens[b]:] = 0
    ref, _ = O.sanm_encoder(xs, lens, sd, cfg)
    out, _, _ = enc(xs.to(cuda), lens)
    err = (out.cpu() - ref).abs().max().item()
    assert err < 1e-3, err


def test_encoder_row_packing_is_bitwise_the_padded_layout(cuda):
    """f16x2 mode, pf_encoder_set_row_packing: computing only a prefix of every sequence (len + k rows) in 16-row slots laid
    back to back gives BITWISE the rows the padded layout computes (every slot starts on the attention's tile boundary, so a
    sequence's key tiles do not move), zeros behind them; SANMEncoder and the SenseVoice encoder (its mid after_norm)."""
    from funasr_amd.sanm_encoder import SenseVoiceEncoderSmall
    g = torch.Generator().manual_seed(5)
    lens = torch.tensor([97, 40, 83, 1, 16, 33, 96, 15], dtype=torch.int32)
    B, T = lens.numel(), int(lens.max())
    xs = torch.randn(B, T, 560, generator=g) * 0.7
    for b in range(B):
        xs[b, lens[b]:] = 0
    pcfg = synth.tiny(synth.PARAFORMER_LARGE, enc_blocks=3)["encoder"]
    scfg = synth.tiny(synth.SENSEVOICE_SMALL, enc_blocks=2, tp_blocks=2)
    ssd = synth.sensevoice_state_dict(scfg, seed=4)
    encs = [_encoder(pcfg, synth.encoder_state_dict(pcfg, seed=9), cuda),
            _encoder(scfg["encoder"], {k[len("encoder."):]: v for k, v in ssd.items() if k.startswith("encoder.")}, cuda,
                     cls=SenseVoiceEncoderSmall)]
    for enc in encs:
        enc.set_precision("f16x2")
        full = enc.set_row_packing(None)(xs.to(cuda), lens)[0].cpu()
        assert torch.equal(enc.set_row_packing(enc.ALL_ROWS)(xs.to(cuda), lens)[0].cpu(), full)
        for k in (0, 1, 3):
            out = enc.set_row_packing(k)(xs.to(cuda), lens)[0].cpu()
            for b in range(B):
                n = min(int(lens[b]) + k, T)
                assert torch.equal(out[b, :n], full[b, :n]), (k, b)
                assert not out[b, n:].any(), (k, b)
        # one sequence alone (nothing to save: padded layout) == the same sequence inside the packed batch
        enc.set_row_packing(1)
        alone = enc(xs[1:2].to(cuda), lens[1:2])[0].cpu()
        assert torch.equal(alone[0, : int(lens[1]) + 1], full[1, : int(lens[1]) + 1])


def test_paraformer_prefix_rows_give_the_all_rows_result(cuda):
    """`recognize_features` computes len + 1 encoder rows per clip in the f16x2 mode (what CifPredictorV2 and the decoder
    read); token ids, counts and CIF peaks equal the all-rows run (`return_intermediate=True`) and the CPU oracle's."""
    from funasr_amd.paraformer import Paraformer
    from oracle import paraformer_oracle as O
    cfg = synth.tiny(synth.PARAFORMER_LARGE, enc_blocks=3, dec_blocks=2, vocab=8404)
    sd = synth.paraformer_state_dict(cfg, seed=21, cif_bias=0.5)
    model = Paraformer.from_config(cfg)
    model.load_state_dict(sd, strict=False)
    model = model.to(cuda).set_precision("f16x2")
    g = torch.Generator().manual_seed(8)
    lens = torch.tensor([120, 77, 119, 30, 64, 5], dtype=torch.int32)
    feats = torch.randn(lens.numel(), 120, 560, generator=g) * 0.7
    for b in range(lens.numel()):
        feats[b, lens[b]:] = 0
    packed = model.recognize_features(feats.to(cuda), lens)
    full = model.recognize_features(feats.to(cuda), lens, return_intermediate=True)
    assert packed["token_num"] == full["token_num"] and packed["raw_ids"] == full["raw_ids"]
    ref = O.paraformer_greedy(feats, lens, sd, cfg)
    assert packed["token_num"] == ref["token_num"].tolist()
    top2 = torch.topk(ref["logits"], 2, dim=-1).values
    for i, (a, b) in enumerate(zip(ref["raw_ids"], packed["raw_ids"])):
        for pos, (x, y) in enumerate(zip(a, b)):       # random-init output layer: only near-ties of the oracle itself may differ
            assert x == y or float(top2[i, pos, 0] - top2[i, pos, 1]) < 1e-4, (i, pos, x, y)


# --------------------------------------------------------------------------------------------------- predictor
def test_cif_bit_exact_given_reference_alphas(cuda):
    from funasr_amd import ops
    g = gold("cif")
    al, hid = t(g["alphas"]).to(cuda), t(g["hidden"]).to(cuda)
    n_max = g["frames"].shape[1]
    peaks, nf, emb = ops.cif(al, hid, n_max)
    assert np.array_equal(peaks.cpu().numpy(), g["fires"])                       # bit-exact float32 "fires"
    assert np.array_equal(fires_of(peaks.cpu()).numpy(), g["fire_idx"])          # fire positions
    assert nf.cpu().tolist() == g["fire_idx"].sum(1).tolist()
    assert np.array_equal(emb.cpu().numpy(), g["frames"])                        # bit-exact weighted sums


def test_predictor_vs_reference_golden(cuda):
    from funasr_amd.cif_predictor import CifPredictorV2
    g = gold("predictor")
    cfg = json.loads(str(g["cfg"]))
    p = CifPredictorV2(**cfg)
    p.load_state_dict(synth.predictor_state_dict(cfg, seed=int(g["seed"])), strict=True)
    p = p.to(cuda)
    hid, lens = t(g["hidden"]).to(cuda), t(g["lens"])
    mask = (torch.arange(hid.shape[1])[None, :] < lens[:, None]).float()[:, None, :]
    emb, tok, alphas, peaks = p(hid, None, mask.to(cuda))
    assert tok.cpu().tolist() == g["token_num"].tolist()
    assert (alphas.cpu() - t(g["alphas"])).abs().max().item() < 2e-6
    assert torch.equal(fires_of(peaks.cpu()), fires_of(g["peaks"]))
    assert emb.shape == g["embeds"].shape
    assert (emb.cpu() - t(g["embeds"])).abs().max().item() < 2e-5


# ----------------------------------------------------------------------------------------------------- decoder
def test_decoder_vs_reference_golden(cuda, f32_mode):
    from funasr_amd.paraformer_decoder import ParaformerSANMDecoder
    g = gold("decoder")
    cfg = json.loads(str(g["cfg"]))
    d = ParaformerSANMDecoder(**cfg)
    d.load_state_dict(synth.decoder_state_dict(cfg, seed=int(g["seed"]), with_embed=True), strict=True)
    d = d.to(cuda).set_precision(f32_mode)
    args = (t(g["memory"]).to(cuda), t(g["mem_lens"]), t(g["embeds"]).to(cuda), t(g["tok_lens"]))
    logits, olens = d(*args)
    assert olens.tolist() == g["tok_lens"].tolist()
    assert (logits.cpu() - t(g["logits"])).abs().max().item() < 2e-4
    ids, _ = d.greedy(*args)                                   # fused GEMM + arg-max
    ref_ids = t(g["logits"]).argmax(-1)
    for b in range(ids.shape[0]):
        n = int(g["tok_lens"][b])
        assert ids[b, :n].cpu().tolist() == ref_ids[b, :n].tolist()


@pytest.mark.parametrize("tag", ["a", "b"])
def test_decoder_with_decoders2_vs_reference_golden(cuda, f32_mode, tag):
    """att_layer_num < num_blocks (decoder.py:363-380, :436-437): pf_decoder_set_decoders2 -- blocks of FFN + FSMN without
    cross-attention after the attention blocks, their taps centred whatever sanm_shfit says (case b) -- against the reference's own
    ParaformerSANMDecoder (tests/golden/decoders2.npz), logits, hidden states and the fused arg-max route."""
    from funasr_amd.paraformer_decoder import ParaformerSANMDecoder
    g = gold("decoders2")
    cfg = json.loads(str(g[f"{tag}_cfg"]))
    d = ParaformerSANMDecoder(**cfg)
    assert d.decoders2 is not None and len(d.decoders2) == cfg["num_blocks"] - cfg["att_layer_num"]
    d.load_state_dict(synth.decoder_state_dict(cfg, seed=int(g[f"{tag}_seed"]), with_embed=True), strict=True)
    d = d.to(cuda).set_precision(f32_mode)
    args = (t(g[f"{tag}_memory"]).to(cuda), t(g[f"{tag}_mem_lens"]), t(g[f"{tag}_embeds"]).to(cuda), t(g[f"{tag}_tok_lens"]))
    logits, hidden, olens = d(*args, return_both=True)
    assert olens.tolist() == g[f"{tag}_tok_lens"].tolist()
    ref_logits, ref_hidden = t(g[f"{tag}_logits"]), t(g[f"{tag}_hidden"])
    for b in range(logits.shape[0]):
        n = int(g[f"{tag}_tok_lens"][b])                         # rows past a sequence's length are masked garbage on both sides
        assert (logits[b, :n].cpu() - ref_logits[b, :n]).abs().max().item() < 2e-4
        assert (hidden[b, :n].cpu() - ref_hidden[b, :n]).abs().max().item() < 2e-4
    ids, _ = d.greedy(*args)
    ref_ids = ref_logits.argmax(-1)
    for b in range(ids.shape[0]):
        n = int(g[f"{tag}_tok_lens"][b])
        assert ids[b, :n].cpu().tolist() == ref_ids[b, :n].tolist()


def test_decoder_with_decoders2_bf16_mode_close_to_reference(cuda):
    from funasr_amd.paraformer_decoder import ParaformerSANMDecoder
    g = gold("decoders2")
    cfg = json.loads(str(g["a_cfg"]))
    d = ParaformerSANMDecoder(**cfg)
    d.load_state_dict(synth.decoder_state_dict(cfg, seed=int(g["a_seed"]), with_embed=True), strict=True)
    d = d.to(cuda).set_precision("bf16")
    args = (t(g["a_memory"]).to(cuda), t(g["a_mem_lens"]), t(g["a_embeds"]).to(cuda), t(g["a_tok_lens"]))
    hid, _ = d(*args, return_hidden=True)
    ref = t(g["a_hidden"])
    for b in range(hid.shape[0]):
        n = int(g["a_tok_lens"][b])
        dd = (hid[b, :n].cpu() - ref[b, :n]).abs()
        assert dd.mean().item() < 0.03 * ref[b, :n].abs().mean().item()       # bf16 operands: the bar of the other bf16-mode tests


# ---------------------------------------------------------------------------------------------------- pipeline
def test_pipeline_token_ids_equal_reference(cuda, f32_mode):
    from funasr_amd.paraformer import Paraformer
    g = gold("pipeline")
    cfg = json.loads(str(g["cfg"]))
    model = Paraformer.from_config(cfg)
    model.load_state_dict(synth.paraformer_state_dict(cfg, seed=int(g["seed"])), strict=False)
    model = model.to(cuda).set_precision(f32_mode)
    res = model.recognize_features(t(g["feats"]).to(cuda), t(g["lens"]), return_intermediate=True)
    assert res["token_num"] == g["token_num"].tolist()
    assert torch.equal(fires_of(res["peaks"].cpu()), fires_of(g["peaks"]))       # CIF fire indices bit-exact
    assert (res["enc"].cpu() - t(g["enc"])).abs().max().item() < 1e-4
    for b, n in enumerate(g["token_num"].tolist()):
        assert res["raw_ids"][b] == g["raw_ids"][b, :n].tolist()


def test_one_call_forward_is_bitwise_the_module_chain(cuda, f32_mode):
    """include/paraformer_hip.h pf_paraformer_forward (the reference's export boundary, funasr/models/paraformer/export_meta.py:44-68:
    speech + speech_lengths -> token ids + token_num) against the module-by-module chain Paraformer.inference runs
    (funasr/models/paraformer/model.py:286-346, 614-616, 642): token counts, ids, alphas and peaks bit for bit -- through the
    Python route (`enqueue_features`) and through ctypes directly, ragged lengths, and a batch in which nothing fires."""
    import ctypes as C
    from funasr_amd import _lib
    from funasr_amd.hip_module import host_i32, stream_ptr
    from funasr_amd.paraformer import Paraformer
    cfg = synth.tiny(synth.PARAFORMER_LARGE, enc_blocks=3, dec_blocks=2, vocab=97)
    model = Paraformer.from_config(cfg)
    model.load_state_dict(synth.paraformer_state_dict(cfg, seed=5, cif_bias=-0.3), strict=False)
    model = model.to(cuda).set_precision(f32_mode)
    g = torch.Generator().manual_seed(8)
    feats = (torch.randn(5, 77, 560, generator=g) * 0.7).to(cuda)
    lens = torch.tensor([77, 60, 33, 77, 9], dtype=torch.int32)
    model._one_call = False
    chain = model.recognize_features(feats, lens, return_intermediate=False)
    inter = model.recognize_features(feats, lens, return_intermediate=True)
    model._one_call = True
    assert model._one_call_ok()
    one = model.recognize_features(feats, lens)
    assert one["token_num"] == chain["token_num"] == inter["token_num"] and max(one["token_num"]) >= 1
    assert one["raw_ids"] == chain["raw_ids"] and one["ids"] == chain["ids"]
    # ctypes, with the optional outputs
    lib, h = model._pipeline()
    B, T = feats.shape[:2]
    lens_c, _ = host_i32(lens, B)
    tok = (C.c_int32 * B)()
    ids = torch.full((B, T + 1), -7, device=cuda, dtype=torch.int32)
    alphas = torch.empty(B, T + 1, device=cuda)
    peaks = torch.empty(B, T + 1, device=cuda)
    pe = model.encoder._pe_table(T, feats.device)
    model.encoder.set_row_packing(model.encoder.ALL_ROWS)
    model.encoder._apply_settings(*model.encoder._ensure_handle())
    n = lib.pf_paraformer_forward(h, feats.data_ptr(), lens_c, B, T, pe.data_ptr(), ids.data_ptr(), T + 1, tok, alphas.data_ptr(), peaks.data_ptr(), stream_ptr())
    assert n == max(inter["token_num"]) and list(tok) == inter["token_num"]      # (the same row layout as `inter`: every row)
    torch.cuda.synchronize()
    for b, k in enumerate(inter["token_num"]):
        assert ids[b, :k].cpu().tolist() == inter["raw_ids"][b]
    assert torch.equal(alphas, inter["alphas"]) and torch.equal(peaks, inter["peaks"])
    enc_ptr = lib.pf_paraformer_encoder_out(h)
    assert enc_ptr
    # nothing fires: zero features of length 1 give alphas far below the threshold -> N = 0, ids untouched (model.py:615-616)
    model2 = Paraformer.from_config(cfg)
    model2.load_state_dict(synth.paraformer_state_dict(cfg, seed=5, cif_bias=-9.0), strict=False)
    model2 = model2.to(cuda).set_precision(f32_mode)
    none = model2.recognize_features(feats[:2, :4], torch.tensor([4, 3], dtype=torch.int32))
    assert none["token_num"] == [0, 0] and none["raw_ids"] == [[], []]
    # too small an id buffer is an error, not an overrun
    small = torch.empty(B, 1, device=cuda, dtype=torch.int32)
    rc = lib.pf_paraformer_forward(h, feats.data_ptr(), lens_c, B, T, pe.data_ptr(), small.data_ptr(), 1, tok, None, None, stream_ptr())
    assert (rc < 0 and "ids_ld" in _lib.last_error()) or max(tok) <= 1


def test_two_phase_forward_interleaved_across_batches_is_bitwise_the_one_call_forward(cuda, f32_mode):
    """include/paraformer_hip.h pf_paraformer_begin / _finish: batch i + 1's encoder + CIF scan are enqueued BEFORE batch i's token
    counts are read and its decoder launched (the serving loop of bench.py). The decoder is still sized by the exact count (the
    .item() at funasr/models/paraformer/cif_predictor.py:311), so ids and counts are those of the one-call forward, whatever the
    interleaving -- batches of different shapes, a batch in which nothing fires in the middle; misuse is an error, not a corruption."""
    from funasr_amd import _lib
    from funasr_amd.paraformer import Paraformer
    cfg = synth.tiny(synth.PARAFORMER_LARGE, enc_blocks=3, dec_blocks=2, vocab=97)
    model = Paraformer.from_config(cfg)
    model.load_state_dict(synth.paraformer_state_dict(cfg, seed=5, cif_bias=-0.3), strict=False)
    model = model.to(cuda).set_precision(f32_mode)
    g = torch.Generator().manual_seed(21)
    shapes = [(5, 77, [77, 60, 33, 77, 9]), (2, 120, [120, 64]), (3, 4, [1, 1, 1]), (7, 50, [50, 49, 48, 10, 50, 3, 25]), (1, 200, [200])]
    batches = []
    for B, T, lens in shapes:
        f = torch.randn(B, T, 560, generator=g) * 0.7
        if T == 4:
            f = torch.full((B, T, 560), -50.0)             # far below the threshold: nothing fires (model.py:615-616)
        batches.append((f.to(cuda), torch.tensor(lens, dtype=torch.int32)))
    want = [model.recognize_features(f, l) for f, l in batches]
    assert max(want[0]["token_num"]) >= 1
    # the loop of bench.py: begin(i + 1), finish(i), collect(i - 1)
    got, ticket, pending = [], model.begin_features(*batches[0]), None
    for i in range(1, len(batches)):
        nxt = model.begin_features(*batches[i])
        fin = model.finish_features(ticket)
        if pending is not None:
            got.append(model.collect(pending))
        pending, ticket = fin, nxt
    fin = model.finish_features(ticket)
    got.append(model.collect(pending))
    got.append(model.collect(fin))
    for w, r in zip(want, got):
        assert r["token_num"] == w["token_num"] and r["raw_ids"] == w["raw_ids"] and r["ids"] == w["ids"]
    # the second phase on ANOTHER stream (the decoder of batch i beside the encoder of batch i + 1): same results
    side = torch.cuda.Stream(device=cuda)
    got, ticket, pending = [], model.begin_features(*batches[0]), None
    for i in range(1, len(batches)):
        nxt = model.begin_features(*batches[i])
        fin = model.finish_features(ticket, stream=side)
        if pending is not None:
            got.append(model.collect(pending))
        pending, ticket = fin, nxt
    fin = model.finish_features(ticket, stream=side)
    got.append(model.collect(pending))
    got.append(model.collect(fin))
    torch.cuda.synchronize()
    for w, r in zip(want, got):
        assert r["token_num"] == w["token_num"] and r["raw_ids"] == w["raw_ids"], "decoder on a second stream"
    # three batches in flight: refused; the two begun ones still finish correctly
    t0 = model.begin_features(*batches[0])
    t1 = model.begin_features(*batches[1])
    with pytest.raises(RuntimeError, match="both slots are in flight"):
        model.begin_features(*batches[3])
    r0 = model.collect(model.finish_features(t0))
    r1 = model.collect(model.finish_features(t1))
    assert r0["raw_ids"] == want[0]["raw_ids"] and r1["raw_ids"] == want[1]["raw_ids"]
    with pytest.raises(RuntimeError, match="not in flight"):
        model.finish_features(t1)
    model.close()                                            # and the object can be dropped and rebuilt
    again = model.recognize_features(*batches[0])
    assert again["raw_ids"] == want[0]["raw_ids"]


def test_pred_timestamp_follows_the_reference_call(cuda):
    """Paraformer.inference(pred_timestamp=True) (paraformer/model.py:668-681): per utterance the reference calls
    ts_prediction_lfr6_standard(pre_peak_index[i], alphas[i], tokens, vad_offset=begin_time, upsample_rate=1) -- cif_peak
    in the `us_alphas` slot, alphas in the `us_peaks` slot -- and post-processes tokens and spans together. The two host
    functions are pinned to the reference in tests/test_timestamps.py; here the wiring is checked on the oracle's outputs."""
    from funasr_amd.paraformer import Paraformer
    from funasr_amd.timestamps import cif_timestamps
    from funasr_amd.tokenizer import CharTokenizer, sentence_postprocess
    from oracle import paraformer_oracle as O
    cfg = synth.tiny(synth.PARAFORMER_LARGE, enc_blocks=2, dec_blocks=1, vocab=64)
    sd = synth.paraformer_state_dict(cfg, seed=21, cif_bias=-0.5)
    model = Paraformer.from_config(cfg)
    model.load_state_dict(sd, strict=False)
    model = model.to(cuda)
    chars = ["<blank>", "<s>", "</s>"] + list("天地玄黄宇宙洪荒日月盈昃辰宿列张寒来暑往秋收冬藏") + ["he@@", "llo", "a", "b", "ok", "world"]
    chars += [f"w{i}" for i in range(64 - len(chars))]
    tok = CharTokenizer(token_list=chars)
    g = torch.Generator().manual_seed(4)
    x = torch.randn(2, 60, 560, generator=g) * 0.7
    lens = torch.tensor([60, 41], dtype=torch.int32)
    x[1, 41:] = 0
    res, meta = model.inference(x.to(cuda), data_lengths=lens, key=["u0", "u1"], tokenizer=tok, data_type="fbank",
                                pred_timestamp=True, begin_time=500)
    ref = O.paraformer_greedy(x, lens, sd, cfg)
    assert [r["key"] for r in res] == ["u0", "u1"]
    for i, r in enumerate(res):
        tokens = tok.ids2tokens(ref["ids"][i])
        _, stamps = cif_timestamps(ref["peaks"][i], ref["alphas"][i], list(tokens), vad_offset=500, upsample_rate=1)
        text, stamps, _ = sentence_postprocess(tokens, stamps)
        assert r["text"] == text and r["timestamp"] == stamps and len(stamps) > 0
        assert all(b >= 500 and e >= b for b, e in r["timestamp"])


# -------------------------------------------------------------------------------------------------- SenseVoice
def test_sensevoice_encoder_and_ctc_vs_reference_golden(cuda, f32_mode):
    from funasr_amd.ctc import CTC
    from funasr_amd.sanm_encoder import SenseVoiceEncoderSmall
    g = gold("sensevoice")
    cfg = json.loads(str(g["cfg"]))
    sd = synth.sensevoice_state_dict(cfg, seed=int(g["seed"]))
    enc = SenseVoiceEncoderSmall(input_layer="pe", **cfg["encoder"])
    enc.load_state_dict({k[len("encoder."):]: v for k, v in sd.items() if k.startswith("encoder.")}, strict=True)
    enc = enc.to(cuda).set_precision(f32_mode)
    out, olens = enc(t(g["xs"]).to(cuda), t(g["lens"]))
    assert olens.tolist() == g["olens"].tolist()
    assert (out.cpu() - t(g["out"])).abs().max().item() < 1e-4
    ctc = CTC(odim=cfg["vocab_size"], encoder_output_size=cfg["encoder"]["output_size"])
    ctc.load_state_dict({"ctc_lo.weight": sd["ctc.ctc_lo.weight"], "ctc_lo.bias": sd["ctc.ctc_lo.bias"]})
    ctc = ctc.to(cuda)
    assert np.array_equal(ctc.argmax(out).cpu().numpy(), g["frame_ids"])


def test_sensevoice_inference_equals_reference_inference(cuda, f32_mode):
    """`SenseVoiceSmall.inference` end to end (query frames by language / text-norm, encoder, fused CTC arg-max,
    unique_consecutive, blank removal, tokenizer.decode, keys) against the fixture recorded from the REFERENCE class's own
    `inference` (funasr/models/sense_voice/model.py:918-1030; oracle/make_golden_sensevoice.py), all six query cases."""
    from funasr_amd.sense_voice import SenseVoiceSmall
    g = gold("sensevoice_inference")
    cfg = json.loads(str(g["config"]))
    sd = synth.sensevoice_state_dict(cfg, seed=int(g["seed"]))
    sd["ctc.ctc_lo.bias"][0] += float(g["ctc_blank_bias_add"])
    model = SenseVoiceSmall.from_config(cfg)
    model.load_state_dict(sd, strict=False)
    model = model.to(cuda).set_precision(f32_mode)

    class IdTokenizer:
        def decode(self, ids):
            return " ".join(str(int(i)) for i in ids)

    feats, lens = t(g["feats"]).to(cuda), t(g["lens"])
    for ci, kw in enumerate(json.loads(str(g["cases"]))):
        res, meta = model.inference(feats, data_lengths=lens, key=[f"utt{i}" for i in range(feats.shape[0])],
                                    tokenizer=IdTokenizer(), frontend=None, device=cuda, data_type="fbank", **kw)
        assert [r["text"] for r in res] == json.loads(str(g[f"texts_{ci}"])), (ci, kw)
        assert [r["key"] for r in res] == json.loads(str(g[f"keys_{ci}"]))


def test_sensevoice_ban_emo_unk_equals_reference_inference(cuda, f32_mode):
    """`inference(..., ban_emo_unk=True)`: <|EMO_UNKNOWN|> (id 25009 of the real 25055 vocabulary) never wins a frame -- the
    ids the REFERENCE class returns with and without the ban (oracle/make_golden_sensevoice_ban.py)."""
    from funasr_amd.sense_voice import SenseVoiceSmall
    g = gold("sensevoice_ban")
    cfg = json.loads(str(g["config"]))
    sd = synth.sensevoice_state_dict(cfg, seed=int(g["seed"]))
    for k, v in json.loads(str(g["bias_add"])).items():
        sd["ctc.ctc_lo.bias"][int(k)] += float(v)
    model = SenseVoiceSmall.from_config(cfg)
    model.load_state_dict(sd, strict=False)
    model = model.to(cuda).set_precision(f32_mode)

    class IdTokenizer:
        def decode(self, ids):
            return " ".join(str(int(i)) for i in ids)

    feats, lens = t(g["feats"]).to(cuda), t(g["lens"])
    for name, kw in (("plain", {}), ("ban", dict(ban_emo_unk=True)), ("plain", {})):
        res, _ = model.inference(feats, data_lengths=lens, key=[f"utt{i}" for i in range(feats.shape[0])],
                                 tokenizer=IdTokenizer(), frontend=None, device=cuda, data_type="fbank", language="auto", **kw)
        assert [r["text"] for r in res] == json.loads(str(g[f"texts_{name}"])), name


def test_beam_search_with_ctc_rescoring_vs_oracle(cuda):
    """`inference(decoding_ctc_weight=0.5, beam_size=3)` on a Paraformer WITH a CTC head (model.py:554-562,629-637): the device
    supplies decoder and CTC log-probs (HIP GEMMs + the row log-softmax kernel), the host runs the beam search that
    tests/test_beam_search.py pins against the reference's BeamSearchPara. Expected n-best: the same search over the CPU
    oracle's scores."""
    import torch.nn.functional as F
    from funasr_amd.beam_search import BeamSearchPara
    from funasr_amd.paraformer import Paraformer
    from oracle import paraformer_oracle as O
    cfg = synth.tiny(synth.PARAFORMER_LARGE, enc_blocks=2, dec_blocks=2, vocab=97)
    sd = synth.paraformer_state_dict(cfg, seed=31, cif_bias=0.4)
    g = torch.Generator().manual_seed(8)
    sd["ctc.ctc_lo.weight"] = torch.randn(97, 512, generator=g) * 0.08
    sd["ctc.ctc_lo.bias"] = torch.randn(97, generator=g) * 0.1
    sd["ctc.ctc_lo.bias"][0] += 1.0
    ec = dict(cfg["encoder"]); input_size = ec.pop("input_size")
    dc = dict(cfg["decoder"]); vocab = dc.pop("vocab_size"); dc.pop("encoder_output_size", None)
    model = Paraformer(encoder="SANMEncoder", encoder_conf=dict(ec, input_layer="pe"), decoder="ParaformerSANMDecoder",
                       decoder_conf=dc, predictor="CifPredictorV2", predictor_conf=dict(cfg["predictor"]), ctc_weight=0.3,
                       input_size=input_size, vocab_size=vocab)
    missing, unexpected = model.load_state_dict(sd, strict=False)
    assert not unexpected and not [k for k in missing if "embed" not in k], (missing, unexpected)
    model = model.to(cuda)
    B, T = 3, 60
    feats = torch.randn(B, T, 560, generator=g) * 0.7
    lens = torch.tensor([60, 41, 25], dtype=torch.int32)
    for b in range(B):
        feats[b, lens[b]:] = 0
    kw = dict(decoding_ctc_weight=0.5, beam_size=3, penalty=0.1, nbest=2, token_list=[str(i) for i in range(97)],
              data_type="fbank", device=cuda)
    res, _ = model.inference(feats.to(cuda), data_lengths=lens, key=["a", "b", "c"], tokenizer=None, frontend=None, **kw)
    with torch.no_grad():
        r = O.paraformer_greedy(feats, lens, sd, cfg)
        am = torch.log_softmax(r["logits"], -1)
        logp = torch.log_softmax(F.linear(r["enc"], sd["ctc.ctc_lo.weight"], sd["ctc.ctc_lo.bias"]), -1).numpy()
    bs = BeamSearchPara(beam_size=3, vocab_size=97, sos=1, eos=2, ctc_weight=0.5, length_bonus_weight=0.1)
    want = []
    for i in range(B):
        n = int(r["token_num"][i])
        for h in bs(am[i, :n], logp[i, : int(r["olens"][i])])[:2]:
            want.append((["a", "b", "c"][i], [t for t in h.yseq[1:-1] if t not in (0, 1, 2)], h.score))
    assert [(x["key"], x["token_int"]) for x in res] == [(k, ids) for k, ids, _ in want]
    assert all(set(x) == {"key", "token_int"} for x in res)          # the reference's record carries no score (model.py:693)
    nb = model.recognize_features_beam(feats.to(cuda), lens)["nbest"]
    assert max(abs(h.score - w[2]) for h, w in zip([h for hyps in nb for h in hyps], want)) < 5e-3
    # without decoding_ctc_weight the same model object (beam search now initialised) still takes the beam route; a fresh
    # model without it is greedy
    plain = Paraformer(encoder="SANMEncoder", encoder_conf=dict(ec, input_layer="pe"), decoder="ParaformerSANMDecoder",
                       decoder_conf=dc, predictor="CifPredictorV2", predictor_conf=dict(cfg["predictor"]), ctc_weight=0.3,
                       input_size=input_size, vocab_size=vocab)
    plain.load_state_dict(sd, strict=False)
    g_res, _ = plain.to(cuda).inference(feats.to(cuda), data_lengths=lens, key=["a", "b", "c"], tokenizer=None, frontend=None,
                                        data_type="fbank", device=cuda)
    assert [x["token_int"] for x in g_res] == r["ids"]


# ------------------------------------------------------------------- full-size, size-independent properties
def test_full_size_batch_independence_and_fused_argmax(cuda, f32_mode):
    """BASELINE config 2 shapes (B=64 x 30 s -> T=500) on a shallow model: (1) every clip's encoder output and token
    ids are bitwise identical whether it is decoded alone or inside the 64-clip batch (utterance DP is exact),
    (2) fused arg-max == arg-max of the materialised logits."""
    from funasr_amd.paraformer import Paraformer
    cfg = synth.tiny(synth.PARAFORMER_LARGE, enc_blocks=2, dec_blocks=1, vocab=8404)
    model = Paraformer.from_config(cfg)
    model.load_state_dict(synth.paraformer_state_dict(cfg, seed=5), strict=False)
    model = model.to(cuda).set_precision(f32_mode)
    g = torch.Generator().manual_seed(1)
    B, T = 64, 500
    feats = torch.randn(B, T, 560, generator=g) * 0.7
    lens = torch.randint(100, T + 1, (B,), generator=g, dtype=torch.int32)
    lens[0] = T
    for b in range(B):
        feats[b, lens[b]:] = 0
    res = model.recognize_features(feats.to(cuda), lens, return_intermediate=True)
    for b in (0, 7, 63):
        L = int(lens[b])
        # same padded length: everything (also the padded frames the CIF conv peeks into, cif_predictor.py:275-277)
        # is bitwise identical, hence the same tokens
        one = model.recognize_features(feats[b:b + 1].to(cuda), lens[b:b + 1], return_intermediate=True)
        assert torch.equal(one["enc"][0].cpu(), res["enc"][b].cpu())
        assert one["raw_ids"][0] == res["raw_ids"][b]
        # truncated to its own length: valid encoder frames still bitwise identical
        cut, _, _ = model.encoder(feats[b:b + 1, :L].to(cuda), lens[b:b + 1])
        assert torch.equal(cut[0].cpu(), res["enc"][b, :L].cpu())
    logits, _ = model.decoder(res["enc"], lens, res["embeds"], torch.tensor(res["token_num"]))
    ids = logits.argmax(-1).cpu()
    for b in range(B):
        assert ids[b, : res["token_num"][b]].tolist() == res["raw_ids"][b]


def test_full_configuration_vs_reference_modules(cuda, f32_mode):
    """The HIP path against the REFERENCE's own SANMEncoder + CifPredictorV2 + ParaformerSANMDecoder at the headline
    configuration -- 50 + 16 blocks, T = 500, vocabulary 8404, the bench's weights, two 30 s clips (tests/golden/full_config.npz,
    oracle/make_golden_full.py; sanm/encoder.py:392-461, paraformer/cif_predictor.py:253-314,818-908, paraformer/decoder.py:
    397-449). Bars (north_star): CIF fire frames and token counts bit-exact, encoder within 1e-3 (asserted 2e-4), decoder
    hidden states within 1e-3, arg-max ids of the random-init output layer equal except where the reference's own top-2
    logits are closer than 1e-4."""
    from funasr_amd.paraformer import Paraformer
    from tests._full_config import check_against_full_config, full_config_fixture
    g, cfg, sd, feats, lens = full_config_fixture()
    model = Paraformer.from_config(cfg)
    model.load_state_dict(sd, strict=False)
    model = model.to(cuda).set_precision(f32_mode)
    x, xl = feats.to(cuda), lens.to(cuda)
    res = model.recognize_features(x, xl, return_intermediate=True)
    hid, _ = model.decoder(res["enc"], res["olens"], res["embeds"], torch.tensor(res["token_num"]), return_hidden=True)
    out = dict(enc=res["enc"].cpu(), alphas=res["alphas"].cpu(), peaks=res["peaks"].cpu(), token_num=res["token_num"],
               raw_ids=res["raw_ids"], hidden=hid.cpu())
    d = check_against_full_config(g, out, enc_tol=2e-4, hid_tol=1e-3, alpha_tol=5e-5)
    print(f"[{f32_mode}] vs reference modules at 50+16 blocks, T=500: encoder max|d| {d['enc']:.2e}, alpha {d['alpha']:.2e}, "
          f"decoder hidden {d['hidden']:.2e}, near-tie flips {d['flips']}")
    # the production call (row-packed encoder, no intermediates) returns the same ids and counts
    res2 = model.recognize_features(x, xl)
    assert res2["raw_ids"] == res["raw_ids"] and res2["token_num"] == res["token_num"]


def test_full_configuration_ragged_batch_vs_oracle(cuda, f32_mode):
    """The headline configuration itself -- Paraformer-large, 50 + 16 blocks, vocabulary 8404, T = 500 frames, B = 8 ragged
    30 s ... 9 s clips from waveforms -- against the CPU oracle clip by clip: encoder <= 1e-3 (north_star), CIF fire
    indices and token counts bit-exact, token ids equal except where the ORACLE's own top-2 logits are closer than 1e-4
    (random-init weights put arg-maxes over 8404 classes on near-ties that any fp32 summation order may flip; the
    fp32-MFMA mode flips such tokens too, tools/mode_parity.py / profiles/r02_mode_parity.json)."""
    from funasr_amd.paraformer import Paraformer
    from funasr_amd.wav_frontend import WavFrontend
    from oracle import paraformer_oracle as O
    cfg = synth.PARAFORMER_LARGE
    sd = synth.paraformer_state_dict(cfg, seed=0, cif_bias=synth.BENCH_CIF_BIAS)
    model = Paraformer.from_config(cfg)
    model.load_state_dict(sd, strict=False)
    model = model.to(cuda).set_precision(f32_mode)
    shift, scale = synth.synthetic_cmvn(560)
    cmvn = torch.stack([shift, scale])
    fe = WavFrontend(cmvn=cmvn, lfr_m=7, lfr_n=6, dither=0.0, device=cuda)
    secs = [30.0, 30.0, 27.3, 24.1, 19.9, 15.0, 12.2, 9.0]
    clips = [synth.speech_like(int(sec * 16000), seed=40 + i) for i, sec in enumerate(secs)]
    lens = [c.numel() for c in clips]
    wav = torch.zeros(len(clips), max(lens))
    for i, c in enumerate(clips):
        wav[i, : lens[i]] = c
    feats, flens = fe(wav.to(cuda), lens)
    assert int(flens.max()) == 500
    res = model.recognize_features(feats, flens, return_intermediate=True)
    enc, peaks = res["enc"].cpu(), res["peaks"].cpu()
    worst, margin, flips = 0.0, 1.0, 0
    with torch.no_grad():
        # the oracle decodes the SAME padded batch (the reference's batch semantics: the CIF convolution of a shorter clip's
        # last frames sees the encoder's output on its padding, cif_predictor.py:275-277, so a clip's tail token depends on
        # the batch it is padded in -- decoding it alone is a different, equally valid, reference result)
        f, fl = O.wav_frontend(clips, cmvn)
        r = O.paraformer_greedy(f, fl, sd, cfg)
        top2 = torch.topk(r["logits"], 2, dim=-1).values
        for i in range(len(clips)):
            T = int(r["olens"][i])
            assert T == int(flens[i])
            worst = max(worst, float((r["enc"][i, :T] - enc[i, :T]).abs().max()))
            fire_c = torch.floor(r["peaks"][i]) >= 1
            assert torch.equal(fire_c, torch.floor(peaks[i, : fire_c.numel()]) >= 1), f"clip {i}: CIF fire indices differ"
            assert int(r["token_num"][i]) == res["token_num"][i]
            ps = torch.cumsum(r["alphas"][i].double(), 0)[: T + 1]
            fr = ps - torch.floor(ps)
            margin = min(margin, float(torch.minimum(fr, 1 - fr)[ps > 0.5].min()))
            for pos, (x, y) in enumerate(zip(r["raw_ids"][i], res["raw_ids"][i])):
                if x != y:
                    flips += 1
                    assert float(top2[i, pos, 0] - top2[i, pos, 1]) < 1e-4, f"clip {i} token {pos}: {x} != {y} and not a near-tie"
    assert worst < 1e-3, worst
    print(f"[{f32_mode}] encoder max|d| {worst:.2e}, min prefix-sum margin {margin:.2e}, near-tie token flips {flips}")
    # ---- the same comparison with a CONFIDENT output layer (synth.confident_output_layer: the closed-form stand-in for
    #      training, calibrated on this batch's own hidden states): top-2 gaps far above any summation-order noise, so the ids
    #      must be identical on every position -- no near-tie waiver
    over, stats = synth.make_paraformer_confident(model, feats, flens)
    assert stats["frac_gap_above_1e3"] >= 0.999 and stats["distinct_classes"] >= 50, stats
    res2 = model.recognize_features(feats, flens)
    W, b = over["decoder.output_layer.weight"], over["decoder.output_layer.bias"]
    gaps = []
    for i in range(len(clips)):
        n = int(r["token_num"][i])
        lg = r["hidden"][i, :n] @ W.T + b
        assert lg.argmax(-1).tolist() == res2["raw_ids"][i], f"clip {i}: token ids differ with the confident output layer"
        t2 = lg.topk(2, dim=-1).values
        gaps.append(t2[:, 0] - t2[:, 1])
    gaps = torch.cat(gaps)
    assert float((gaps > 1e-3).float().mean()) >= 0.999
    print(f"[{f32_mode}] confident output layer: {gaps.numel()} tokens identical, oracle top-2 gap min {float(gaps.min()):.3f} "
          f"median {float(gaps.median()):.1f}, {stats['distinct_classes']} classes")


def test_bf16_operand_mode_stays_close_to_fp32_mode(cuda):
    """Throughput mode (bf16 operands for GEMMs + attention, fp32 accumulate / residual / LN / softmax) on the
    full-depth 50-block encoder: measured against the fp32 parity mode of the same weights and input. The reference's
    OWN bf16=True mode deviates from its fp32 output by mean 0.015 / max 0.138 on activations of magnitude 0.8
    (SURVEY.md section 7); operand-only bf16 must do at least as well."""
    from funasr_amd.sanm_encoder import SANMEncoder
    cfg = synth.PARAFORMER_LARGE["encoder"]
    sd = synth.encoder_state_dict(cfg, seed=3)
    enc = _encoder(cfg, sd, cuda).set_precision("fp32")
    g = torch.Generator().manual_seed(9)
    x = torch.randn(3, 120, 560, generator=g) * 0.6
    lens = torch.tensor([120, 77, 101], dtype=torch.int32)
    ref, _, _ = enc(x.to(cuda), lens)
    enc.set_precision("bf16")
    out, _, _ = enc(x.to(cuda), lens)
    enc.set_precision("fp32")
    again, _, _ = enc(x.to(cuda), lens)
    assert torch.equal(again, ref)                              # switching back restores the parity mode bit for bit
    m = (torch.arange(120)[None, :] < lens[:, None]).to(cuda)
    d = (out - ref).abs()[m]
    scale = ref.abs()[m].mean().item()
    assert d.mean().item() < 0.02 * scale and d.max().item() < 0.25 * max(scale, 1.0), (d.mean().item(), d.max().item(), scale)


def test_edge_cases_short_silent_and_zero_token_utterances(cuda, f32_mode):
    """Edge cases of the offline path: (1) an utterance shorter than one 25 ms window gets one window of its own length
    (wav_frontend.py:176; test_frontend_clips_shorter_than_one_window), a one-sample clip has none; (2) one-frame utterances and a batch where some clips fire no token at all: those
    clips come back empty, their neighbours are unaffected (bitwise) -- where the reference raises IndexError for a
    zero-token LAST utterance (cif_predictor.py:887) this path returns an empty hypothesis; (3) a 60 s clip (T = 1000)."""
    from funasr_amd.paraformer import Paraformer
    from funasr_amd.wav_frontend import WavFrontend
    from oracle import paraformer_oracle as O
    sh, sc = synth.synthetic_cmvn()
    fe = WavFrontend(cmvn=torch.stack([sh, sc]), lfr_m=7, lfr_n=6, dither=0.0)
    feats, flens = fe(torch.zeros(1, 399).to(cuda), [399])
    assert flens.tolist() == [1] and feats.shape == (1, 1, 560)
    with pytest.raises(Exception, match="window"):
        fe(torch.zeros(1, 399).to(cuda), [1])
    feats, flens = fe(torch.zeros(2, 400).to(cuda), [400, 400])           # exactly one window -> one LFR frame
    assert flens.tolist() == [1, 1] and feats.shape[1] == 1
    cfg = synth.tiny(synth.PARAFORMER_LARGE, enc_blocks=2, dec_blocks=1, vocab=97)
    sd = synth.paraformer_state_dict(cfg, seed=5, cif_bias=-6.0)          # alpha ~ 0.002: almost nothing fires
    model = Paraformer.from_config(cfg)
    model.load_state_dict(sd, strict=False)
    model = model.to(cuda).set_precision(f32_mode)
    g = torch.Generator().manual_seed(2)
    x = torch.randn(3, 40, 560, generator=g) * 0.7
    lens = torch.tensor([40, 3, 25], dtype=torch.int32)
    for b in range(3):
        x[b, lens[b]:] = 0
    res = model.recognize_features(x.to(cuda), lens, return_intermediate=True)
    ref = O.paraformer_greedy(x[:1], lens[:1], sd, cfg)
    assert res["token_num"][0] == int(ref["token_num"][0]) and res["raw_ids"][0] == ref["raw_ids"][0]
    assert all(len(r) == n for r, n in zip(res["raw_ids"], res["token_num"]))
    one = model.recognize_features(x[:1].to(cuda), lens[:1], return_intermediate=True)
    assert torch.equal(one["enc"][0], res["enc"][0]) and one["raw_ids"][0] == res["raw_ids"][0]
    # tail threshold 0.45 + tiny alphas: floor(sum) can be 0 -> empty hypothesis, no exception
    assert min(res["token_num"]) >= 0
    # long clip: T = 1000 frames
    xl = torch.randn(1, 1000, 560, generator=g) * 0.7
    rl = model.recognize_features(xl.to(cuda), [1000], return_intermediate=True)
    refl = O.paraformer_greedy(xl, torch.tensor([1000], dtype=torch.int32), sd, cfg)
    assert (rl["enc"].cpu() - refl["enc"]).abs().max().item() < 1e-3
    assert rl["raw_ids"] == refl["raw_ids"]


def test_bf16_operand_decoder_stays_close_to_fp32_decoder(cuda):
    """bf16-operand decoder (all GEMMs + cross-attention, fused arg-max route) against the fp32 decoder of the same
    weights: final hidden states within a few percent, most arg-max ids unchanged even with random-init weights."""
    from funasr_amd.paraformer_decoder import ParaformerSANMDecoder
    cfg = synth.PARAFORMER_LARGE["decoder"]
    sd = synth.decoder_state_dict(cfg, seed=4, with_embed=True)
    kw = {k: v for k, v in cfg.items()}
    dec = ParaformerSANMDecoder(**kw)
    dec.load_state_dict(sd, strict=True)
    dec = dec.to(cuda).set_precision("fp32")
    g = torch.Generator().manual_seed(11)
    B, T, N = 3, 90, 14
    mem = torch.randn(B, T, 512, generator=g) * 0.8
    emb = torch.randn(B, N, 512, generator=g) * 0.8
    mlen, tlen = [90, 61, 75], [14, 9, 11]
    _, ids32, hid32, _ = dec._run(mem.to(cuda), mlen, emb.to(cuda), tlen, want_logits=False, want_ids=True, want_hidden=True)
    dec.set_precision("bf16")
    _, ids16, hid16, _ = dec._run(mem.to(cuda), mlen, emb.to(cuda), tlen, want_logits=False, want_ids=True, want_hidden=True)
    dec.set_precision("fp32")
    _, ids32b, hid32b, _ = dec._run(mem.to(cuda), mlen, emb.to(cuda), tlen, want_logits=False, want_ids=True, want_hidden=True)
    assert torch.equal(hid32, hid32b) and torch.equal(ids32, ids32b)
    valid = torch.zeros(B, N, dtype=torch.bool)
    for b in range(B):
        valid[b, : tlen[b]] = True
    d = (hid16 - hid32).abs().cpu()[valid]
    scale = hid32.abs().cpu()[valid].mean().item()
    assert d.mean().item() < 0.03 * scale, (d.mean().item(), scale)
    same = (ids16.cpu()[valid] == ids32.cpu()[valid]).float().mean().item()
    assert same >= 0.7, same


def test_frontend_dither_is_seeded_noise_with_the_reference_statistics(cuda):
    """`dither` (kaldi.fbank, reference default 1.0: wav_frontend.py:106,171-181; torchaudio adds randn * dither to every sample of
    every frame before the DC removal). The HIP frontend draws it from a counter-based generator: (a) dither = 0 is the
    deterministic path, untouched; (b) one seed -> one feature sequence, another seed -> another; (c) parity is statistical: on
    silence the log-mel energies are those of pure noise -- per-bin mean and spread equal the oracle's (torch.randn) within
    sampling error; on speech-like audio the dithered features stay as close to the clean ones as the oracle's do."""
    from funasr_amd.wav_frontend import WavFrontend
    from oracle import paraformer_oracle as O
    n = 16000 * 20
    silence = torch.zeros(1, n)
    speech = synth.speech_like(n, seed=7)[None]
    def feats(w, dither, seed=11, calls=1):
        fe = WavFrontend(cmvn=None, lfr_m=1, lfr_n=1, dither=dither, dither_seed=seed, device=cuda)
        out = [fe(w.to(cuda), [n])[0][0].cpu() for _ in range(calls)]
        return out if calls > 1 else out[0]
    clean = feats(speech, 0.0)
    assert torch.equal(clean, feats(speech, 0.0, seed=99))                          # (a)
    a1, a2 = feats(speech, 1.0, seed=11, calls=2)
    b1 = feats(speech, 1.0, seed=11)
    c1 = feats(speech, 1.0, seed=12)
    assert torch.equal(a1, b1) and not torch.equal(a1, a2) and not torch.equal(a1, c1)   # (b): seed + call index
    g = torch.Generator().manual_seed(5)
    ref_sil = O.kaldi_fbank(silence[0] * 32768, dither=1.0, generator=g)
    got_sil = feats(silence, 1.0)
    assert got_sil.shape == ref_sil.shape and got_sil.shape[0] > 1900
    # (c) ~2000 frames: the per-bin mean of a log chi-square-like variable has a standard error of ~0.01-0.03
    assert (got_sil.mean(0) - ref_sil.mean(0)).abs().max().item() < 0.08, (got_sil.mean(0) - ref_sil.mean(0)).abs().max().item()
    assert (got_sil.std(0) / ref_sil.std(0) - 1).abs().max().item() < 0.15
    ref_noisy = O.kaldi_fbank(speech[0] * 32768, dither=1.0, generator=g)
    ref_clean = O.kaldi_fbank(speech[0] * 32768)
    d_ref = (ref_noisy - ref_clean).abs().mean().item()
    d_got = (a1 - clean).abs().mean().item()
    assert d_got < 2.0 * d_ref + 1e-3 and d_got > 0.3 * d_ref, (d_got, d_ref)



def test_frontend_cross_check_returns_the_same_features(cuda):
    """WavFrontend(verify=True): fbank_kernel evaluates every frame twice from its samples in registers and repeats until two runs
    agree (pf_frontend_set_verify; for GPUs shared between processes). Alone on the GPU nothing disagrees and the features are the
    bits of the plain kernel."""
    from funasr_amd import synth
    from funasr_amd.wav_frontend import WavFrontend
    sh, sc = synth.synthetic_cmvn(560)
    wav = torch.stack([synth.speech_like(48000, seed=3 + i) for i in range(4)]).to(cuda)
    lens = [48000, 31000, 16000, 400]
    plain = WavFrontend(cmvn=torch.stack([sh, sc]), lfr_m=7, lfr_n=6, dither=0.0, device=cuda, verify=False)
    check = WavFrontend(cmvn=torch.stack([sh, sc]), lfr_m=7, lfr_n=6, dither=0.0, device=cuda, verify=True)
    a, la = plain(wav, lens)
    b, lb = check(wav, lens)
    assert torch.equal(a, b) and torch.equal(torch.as_tensor(la), torch.as_tensor(lb))
    assert check.faults() == 0 and plain.faults() == 0


def test_frontend_next_to_the_small_block_f16x2_gemm_on_a_second_stream(cuda):
    """The configuration that made fbank_kernel fault: this library's 128 x 128 f16x2 GEMM on a second HIP stream of the
    same process while the frontend runs. With the non-matrix kernels built without packed-fp32 instructions every frontend call
    returns the bits it returns alone, and the cross-check sees nothing (before: thousands of disagreements per second)."""
    import time
    from funasr_amd import ops, synth
    from funasr_amd.wav_frontend import WavFrontend
    g = torch.Generator().manual_seed(0)
    sh, sc = synth.synthetic_cmvn(560)
    fe = WavFrontend(cmvn=torch.stack([sh, sc]), lfr_m=7, lfr_n=6, dither=0.0, device=cuda, verify=True)
    wav = synth.speech_like(235000, seed=7).to(cuda)[None]
    ref = fe(wav, [235000])[0].clone()
    side = torch.cuda.Stream(device=cuda)
    with torch.cuda.stream(side):
        a = ops.split2(torch.randn(1024, 512, generator=g).to(cuda), 8)
        w = ops.split2((torch.randn(2048, 512, generator=g) * 512 ** -0.5).to(cuda), 12)
        b = torch.zeros(2048, device=cuda)
    torch.cuda.synchronize()
    t0, calls = time.time(), 0
    while time.time() - t0 < 3.0:
        with torch.cuda.stream(side):
            for _ in range(40):
                ops.gemm_f16x2(a, w, b, scale_exp=20, tile=3)
        for _ in range(4):
            assert torch.equal(fe(wav, [235000])[0], ref)
            calls += 1
    torch.cuda.synchronize()
    assert calls > 1000 and fe.faults() == 0, (calls, fe.faults())
