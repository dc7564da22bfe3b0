"""CPU suite: pins the oracle (oracle/paraformer_oracle.py) to the reference.

Fixtures under tests/golden/ were produced by oracle/make_golden.py from the reference's own modules
(/root/reference, imported read-only) and from the reference-vendored kaldi-native-fbank; weights are rebuilt from
the stored seed. Bars: bit-exact for indices / integer results and pure data movement; float32-roundoff for
tensors that go through ATen matmuls with a different blocking (stated per test).
"""
import json
import os

import numpy as np
import pytest
import torch

from funasr_amd import synth
from oracle import paraformer_oracle as O

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def gold(name):
    return np.load(os.path.join(GOLD, name + ".npz"), allow_pickle=False)


def t(a):
    return torch.from_numpy(np.asarray(a))


def test_lfr_bit_exact():
    g = gold("lfr")
    for T in (1, 2, 3, 5, 6, 7, 8, 13, 100):
        out = O.apply_lfr(t(g[f"in_{T}"]), 7, 6)
        assert np.array_equal(out.numpy(), g[f"out_{T}"]), T


def test_load_cmvn_matches_reference_parser():
    g = gold("frontend")
    cmvn = O.load_cmvn(os.path.join(GOLD, "am.mvn"))
    assert cmvn.shape == (2, 560)
    assert np.array_equal(cmvn.numpy(), g["cmvn"])


def test_fbank_matches_kaldi_native_fbank():
    """kaldi-native-fbank runs its FFT in float64 and builds its tables with float scalars, torchaudio (restated by
    the oracle) works in float32: agreement is float32-roundoff in the power domain, i.e. ~1e-4 in log-mel except
    where a mel energy is nearly cancelled. Bar: max |d| <= 2e-3, mean |d| <= 2e-5 (log-mel values are ~5..20)."""
    g = gold("frontend")
    for k in ("a", "b"):
        wave = t(g[f"pcm_{k}"].astype(np.float32) / 32768.0) * 32768.0
        fb = O.kaldi_fbank(wave)
        ref = t(g[f"fbank_knf_{k}"])
        assert fb.shape == ref.shape
        d = (fb - ref).abs()
        assert d.max().item() <= 2e-3, d.max().item()
        assert d.mean().item() <= 2e-5, d.mean().item()


FBANK_OPTION_CASES = [(w, snip) for w in ("hamming", "hanning", "povey", "rectangular", "blackman") for snip in (True, False)
                      if not (w == "hamming" and snip)]


@pytest.mark.parametrize("window,snip", FBANK_OPTION_CASES)
def test_fbank_window_and_snip_edges_options_match_kaldi_native_fbank(window, snip):
    """WavFrontend's `window` / `snip_edges` (wav_frontend.py:178-180) against the reference-vendored kaldi-native-fbank with the same
    FrameExtractionOptions (oracle/make_golden_fbank_options.py). Same bar as the default options; the rectangular window leaks the
    residual DC / pre-emphasis edge into every bin, so its near-cancelled mel energies sit at a looser maximum (5e-3)."""
    g, go = gold("frontend"), gold("fbank_options")
    for k in ("a", "b"):
        wave = t(g[f"pcm_{k}"].astype(np.float32) / 32768.0) * 32768.0
        fb = O.kaldi_fbank(wave, window_type=window, snip_edges=snip)
        ref = t(go[f"{k}_{window}_{'snip' if snip else 'nosnip'}"])
        assert fb.shape == ref.shape
        d = (fb - ref).abs()
        assert d.max().item() <= (5e-3 if window == "rectangular" else 2e-3), d.max().item()
        assert d.mean().item() <= 3e-5, d.mean().item()


def test_fbank_snip_edges_false_on_less_than_one_window_mirrors_repeatedly():
    go = gold("fbank_options")
    fb = O.kaldi_fbank(t(go["short_pcm"].astype(np.float32)), snip_edges=False)
    ref = t(go["short_hamming_nosnip"])
    assert fb.shape == ref.shape == (2, 80)
    assert (fb - ref).abs().max().item() <= 2e-3


TINY_CLIPS = (399, 300, 257, 256, 200, 129, 100, 64, 33)


def test_fbank_of_clips_shorter_than_one_window_matches_kaldi_native_fbank():
    """wav_frontend.py:176: `frame_length=min(self.frame_length, waveform_length / self.fs * 1000)` -- a clip shorter than 25 ms gets ONE
    window of its own length (float32 tensor arithmetic: n samples for every n in 2..399 at 16 kHz), and the FFT size follows (512 down to
    64 points here). Against the reference-vendored kaldi-native-fbank with frame_length_ms = n / 16."""
    for n in range(2, 401):
        assert O.short_clip_window_size(n) == n
    assert O.short_clip_frame_length_ms(400) == 25 and O.short_clip_frame_length_ms(16000) == 25
    go = gold("fbank_options")
    for n in TINY_CLIPS:
        fb = O.kaldi_fbank(t(go[f"tiny_{n}_pcm"].astype(np.float32)), 80, O.short_clip_frame_length_ms(n))
        ref = t(go[f"tiny_{n}_hamming_snip"])
        assert fb.shape == ref.shape == (1, 80)
        assert (fb - ref).abs().max().item() <= 2e-3, n
    fb = O.kaldi_fbank(t(go["tiny_300_pcm"].astype(np.float32)), 80, O.short_clip_frame_length_ms(300), window_type="povey")
    assert (fb - t(go["tiny_300_povey_snip"])).abs().max().item() <= 2e-3
    # through the batch-level restatement: one LFR row of seven copies of the single frame
    feats, lens = O.wav_frontend([t(go["tiny_200_pcm"].astype(np.float32) / 32768.0)], None)
    assert lens.tolist() == [1] and feats.shape == (1, 1, 560) and torch.equal(feats[0, 0, :80], feats[0, 0, 480:])


def test_frontend_lfr_cmvn_on_reference_fbank_bit_exact():
    g = gold("frontend")
    cmvn = t(g["cmvn"])
    for k in ("a", "b"):
        feats = O.apply_cmvn(O.apply_lfr(t(g[f"fbank_knf_{k}"]), 7, 6), cmvn)
        assert np.array_equal(feats.numpy(), g[f"feats_{k}"])


def test_wav_frontend_end_to_end_close_to_reference_features():
    g = gold("frontend")
    waves = [t(g["pcm_a"].astype(np.float32) / 32768.0), t(g["pcm_b"].astype(np.float32) / 32768.0)]
    feats, lens = O.wav_frontend(waves, t(g["cmvn"]))
    assert lens.tolist() == [g["feats_a"].shape[0], g["feats_b"].shape[0]]
    assert (feats[0, : lens[0]] - t(g["feats_a"])).abs().max().item() < 5e-4
    assert (feats[1, : lens[1]] - t(g["feats_b"])).abs().max().item() < 5e-4
    assert (feats[1, lens[1]:] == 0).all()


def test_positional_encoding_bit_exact():
    g = gold("pe")
    pe = O.sinusoidal_pe(600, 560)
    assert np.array_equal(pe[:40].numpy(), g["head"])
    assert np.array_equal(pe[560:600].numpy(), g["tail"])


def test_encoder_matches_reference():
    g = gold("encoder")
    cfg = json.loads(str(g["cfg"]))
    sd = synth.encoder_state_dict(cfg, seed=int(g["seed"]))
    assert abs(sum(v.double().abs().sum() for v in sd.values()).item() - float(g["checksum"])) < 1e-6 * float(g["checksum"])
    inter = []
    out, olens = O.sanm_encoder(t(g["xs"]), t(g["lens"]), sd, cfg, collect=inter)
    assert olens.tolist() == g["olens"].tolist()
    # same ATen kernels, same op order -> expect (near) bit equality; bar 1e-5 absolute on O(1) activations
    assert (inter[0] - t(g["block1"])).abs().max().item() < 1e-5
    assert (inter[2] - t(g["block3"])).abs().max().item() < 1e-5
    assert (out - t(g["out"])).abs().max().item() < 1e-5


def test_cif_bit_exact():
    g = gold("cif")
    al, hid = t(g["alphas"]), t(g["hidden"])
    fires, fire = O.cif_fires(al)
    assert np.array_equal(fire.numpy(), g["fire_idx"])
    assert np.array_equal(fires.numpy(), g["fires"])
    frames, fires2, n = O.cif_frames(hid, al)
    assert frames.shape == g["frames"].shape
    assert np.array_equal(frames.numpy(), g["frames"])


def test_predictor_matches_reference():
    g = gold("predictor")
    cfg = json.loads(str(g["cfg"]))
    sd = synth.predictor_state_dict(cfg, seed=int(g["seed"]))
    emb, tok, alphas, peaks = O.cif_predictor(t(g["hidden"]), t(g["lens"]), sd, cfg)
    assert np.array_equal(tok.numpy(), g["token_num"])
    # conv1d/linear blocking depends on the ATen thread count: alphas agree to 1 ulp, the integer results exactly
    assert (alphas - t(g["alphas"])).abs().max().item() <= 2e-7
    assert np.array_equal(np.floor(peaks.numpy()) >= 1, np.floor(g["peaks"]) >= 1)
    assert (peaks - t(g["peaks"])).abs().max().item() <= 2e-6
    assert emb.shape == g["embeds"].shape
    assert (emb - t(g["embeds"])).abs().max().item() <= 1e-5


def test_decoder_matches_reference():
    g = gold("decoder")
    cfg = json.loads(str(g["cfg"]))
    sd = synth.decoder_state_dict(cfg, seed=int(g["seed"]))
    logits = O.paraformer_decoder(t(g["memory"]), t(g["mem_lens"]), t(g["embeds"]), t(g["tok_lens"]), sd, cfg)
    assert (logits - t(g["logits"])).abs().max().item() < 2e-5


@pytest.mark.parametrize("tag", ["a", "b"])
def test_decoder_with_decoders2_matches_reference(tag):
    """att_layer_num < num_blocks: the blocks without cross-attention (decoder.py:363-380, :436-437; their FSMN is built at
    sanm_shfit 0 even when the attention blocks are shifted, case b) against the reference's own class."""
    g = gold("decoders2")
    cfg = json.loads(str(g[f"{tag}_cfg"]))
    sd = synth.decoder_state_dict(cfg, seed=int(g[f"{tag}_seed"]))
    logits, hidden = O.paraformer_decoder(t(g[f"{tag}_memory"]), t(g[f"{tag}_mem_lens"]), t(g[f"{tag}_embeds"]), t(g[f"{tag}_tok_lens"]),
                                          sd, cfg, return_hidden=True)
    assert (logits - t(g[f"{tag}_logits"])).abs().max().item() < 2e-5
    assert (hidden - t(g[f"{tag}_hidden"])).abs().max().item() < 2e-5


def test_pipeline_token_ids_equal_reference():
    g = gold("pipeline")
    cfg = json.loads(str(g["cfg"]))
    sd = synth.paraformer_state_dict(cfg, seed=int(g["seed"]))
    res = O.paraformer_greedy(t(g["feats"]), t(g["lens"]), sd, cfg)
    assert res["token_num"].tolist() == g["token_num"].tolist()
    assert np.array_equal(res["alphas"].numpy(), g["alphas"]) or (res["alphas"] - t(g["alphas"])).abs().max() < 1e-6
    fire_ref = np.floor(g["peaks"]) >= 1            # cif_peak >= 1 marks a fire
    assert np.array_equal(np.floor(res["peaks"].numpy()) >= 1, fire_ref)
    for b in range(len(res["raw_ids"])):
        n = int(g["token_num"][b])
        assert res["raw_ids"][b] == g["raw_ids"][b, :n].tolist()
    assert (res["enc"] - t(g["enc"])).abs().max().item() < 1e-5


from tests._full_config import check_against_full_config, full_config_fixture  # noqa: E402


def test_oracle_equals_reference_modules_at_the_headline_configuration():
    """The restatement against the reference's own modules where it matters: 50 + 16 blocks, T = 500, V = 8404
    (sanm/encoder.py:392-461, paraformer/cif_predictor.py:253-314,818-908, paraformer/decoder.py:397-449)."""
    g, cfg, sd, feats, lens = full_config_fixture()
    torch.set_num_threads(max(1, min(8, len(os.sched_getaffinity(0)))))
    with torch.no_grad():
        res = O.paraformer_greedy(feats, lens, sd, cfg)
    d = check_against_full_config(g, res, enc_tol=2e-5, hid_tol=1e-4, alpha_tol=1e-6)      # measured: 0.0, 3.2e-5, 3.6e-7
    assert d["flips"] == []          # same ATen kernels, same operation order: not even near-ties move
    lg = res["logits"][:, ::32, ::7]
    assert float((lg[:, : g["logit_rows"].shape[1]] - t(g["logit_rows"])).abs().max()) < 5e-5


def test_sensevoice_encoder_and_ctc_match_reference():
    g = gold("sensevoice")
    cfg = json.loads(str(g["cfg"]))
    sd = synth.sensevoice_state_dict(cfg, seed=int(g["seed"]))
    out, olens = O.sanm_encoder(t(g["xs"]), t(g["lens"]), sd, cfg["encoder"], "encoder.", eps=1e-5)
    assert olens.tolist() == g["olens"].tolist()
    assert (out - t(g["out"])).abs().max().item() < 1e-5
    logp = torch.log_softmax(torch.nn.functional.linear(out, sd["ctc.ctc_lo.weight"], sd["ctc.ctc_lo.bias"]), -1)
    assert np.array_equal(logp.argmax(-1).numpy(), g["frame_ids"])


def test_oracle_sensevoice_glue_reproduces_reference_inference():
    """the oracle's SenseVoice path (query frames, encoder, CTC arg-max, unique_consecutive, blank removal) against the
    fixture recorded from the reference class's own `inference` (oracle/make_golden_sensevoice.py)"""
    import json
    from funasr_amd import synth
    from oracle import paraformer_oracle as O
    g = np.load(os.path.join(GOLD, "sensevoice_inference.npz"), allow_pickle=False)
    cfg = json.loads(str(g["config"]))
    sd = synth.sensevoice_state_dict(cfg, seed=int(g["seed"]))
    sd["ctc.ctc_lo.bias"][0] += float(g["ctc_blank_bias_add"])
    feats, lens = torch.from_numpy(g["feats"]), torch.from_numpy(g["lens"])
    lid_dict = {"auto": 0, "zh": 3, "en": 4, "yue": 7, "ja": 11, "ko": 12, "nospeech": 13}
    for ci, kw in enumerate(json.loads(str(g["cases"]))):
        tn = kw.get("text_norm") or ("withitn" if kw.get("use_itn", False) else "woitn")
        with torch.no_grad():
            r = O.sensevoice_greedy(feats, lens, sd, cfg, language_id=lid_dict.get(kw.get("language", "auto"), 0),
                                    textnorm_id={"withitn": 14, "woitn": 15}[tn])
        assert [" ".join(str(i) for i in ids) for ids in r["ids"]] == json.loads(str(g[f"texts_{ci}"])), ci
