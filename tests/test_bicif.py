"""BiCifParaformer / CifPredictorV3 (SURVEY §8 f rank 2): the CPU oracle against goldens made by the reference's own classes
(oracle/make_golden_bicif.py), and the HIP path against both."""
import json
import os

import numpy as np
import pytest
import torch

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "bicif.npz")


def _gold():
    return np.load(GOLD, allow_pickle=False)


def _t(g, key):
    return torch.from_numpy(g[key])


def _fires(peaks, thr=1.0 - 1e-4):
    return [torch.nonzero(row >= thr).flatten().tolist() for row in peaks]


def test_oracle_predictor_equals_reference_golden():
    from oracle import bicif_oracle as BO
    g = _gold()
    for name in json.loads(str(g["variants"])):
        cfg = json.loads(str(g[f"{name}_cfg"]))
        sd = BO.predictor_v3_state_dict(cfg, seed=int(g[f"{name}_seed"]), cif_bias=-0.6)
        hidden, lens = _t(g, f"{name}_hidden"), _t(g, f"{name}_lens")
        emb, tok, alphas, peaks = BO.predictor_v3(hidden, lens, sd, cfg)
        assert tok.tolist() == g[f"{name}_token_num"].tolist()
        assert emb.shape == g[f"{name}_embeds"].shape and (emb - _t(g, f"{name}_embeds")).abs().max().item() < 2e-5
        assert (alphas - _t(g, f"{name}_alphas")).abs().max().item() < 2e-6
        assert _fires(peaks, 1.0) == _fires(_t(g, f"{name}_peaks"), 1.0)
        usa, usp = BO.upsample_timestamp(hidden, lens, tok.round().long(), sd, cfg)
        assert (usa - _t(g, f"{name}_us_alphas")).abs().max().item() < 2e-6
        assert _fires(usp) == _fires(_t(g, f"{name}_us_peaks"))


def test_oracle_lstm_equals_torch():
    from oracle import bicif_oracle as BO
    torch.manual_seed(3)
    for layers, bid in ((1, True), (2, False)):
        ref = torch.nn.LSTM(24, 16, layers, batch_first=True, bidirectional=bid)
        x = torch.randn(3, 11, 24)
        with torch.no_grad():
            want, _ = ref(x)
        got = BO.lstm(x, {k: v.detach() for k, v in ref.state_dict().items()}, "", layers=layers, bidirectional=bid)
        assert (got - want).abs().max().item() < 1e-6


def _e2e_setup(g):
    from oracle import bicif_oracle as BO
    from funasr_amd import synth
    cfg = json.loads(str(g["e2e_cfg"]))
    seed = int(g["e2e_seed"])
    sd = synth.paraformer_state_dict(cfg, seed=seed, cif_bias=-0.3)
    sd.update(BO.predictor_v3_state_dict(cfg["predictor"], seed=seed + 5, prefix="predictor.", cif_bias=-0.3))
    return cfg, sd, json.loads(str(g["e2e_vocab"])), json.loads(str(g["e2e_results"]))


def _texts_and_stamps(ids, us_alphas, us_peaks, olens, vocab):
    from funasr_amd.timestamps import ts_prediction_lfr6_standard
    from funasr_amd.tokenizer import CharTokenizer, sentence_postprocess
    tok = CharTokenizer(token_list=vocab, unk_symbol="<unk>")
    out = []
    for b, y in enumerate(ids):
        tokens = tok.ids2tokens(y)
        n = int(olens[b]) * 3
        _, stamps = ts_prediction_lfr6_standard(us_alphas[b][:n], us_peaks[b][:n], list(tokens))
        text, stamps, _ = sentence_postprocess(tokens, stamps)
        out.append((text, stamps))
    return out


def test_oracle_end_to_end_equals_reference_inference():
    from oracle import bicif_oracle as BO
    g = _gold()
    cfg, sd, vocab, want = _e2e_setup(g)
    res = BO.bicif_greedy(_t(g, "e2e_feats"), _t(g, "e2e_lens"), sd, cfg)
    assert res["token_num"].tolist() == g["e2e_token_num"].tolist()
    assert (res["us_alphas"] - _t(g, "e2e_us_alphas")).abs().max().item() < 5e-6
    got = _texts_and_stamps(res["ids"], res["us_alphas"], res["us_peaks"], res["olens"], vocab)
    for (text, stamps), w in zip(got, want):
        assert text == w["text"] and stamps == w["timestamp"]


# ------------------------------------------------------------------------------------------------------------ GPU
@pytest.mark.gpu
def test_lstm_kernel_equals_torch(cuda):
    from funasr_amd import ops
    torch.manual_seed(5)
    for B, T, D, H, bid in ((3, 11, 64, 32, True), (70, 5, 32, 64, False), (1, 40, 64, 128, True), (64, 3, 32, 48, True)):
        ref = torch.nn.LSTM(D, H, 1, batch_first=True, bidirectional=bid)
        x = torch.randn(B, T, D)
        with torch.no_grad():
            want, _ = ref(x)
        sd = ref.state_dict()
        sfx = ("", "_reverse") if bid else ("",)
        stack = lambda n: torch.stack([sd[f"{n}_l0{s}"] for s in sfx]).to(cuda)
        got = ops.lstm(x.to(cuda), stack("weight_ih"), stack("weight_hh"), stack("bias_ih"), stack("bias_hh")).cpu()
        assert got.shape == want.shape and (got - want).abs().max().item() < 2e-5, (B, T, D, H, bid)


@pytest.mark.gpu
def test_predictor_v3_on_the_gpu_equals_reference_golden(cuda):
    from funasr_amd.cif_predictor import CifPredictorV3
    from oracle import bicif_oracle as BO
    g = _gold()
    for name in json.loads(str(g["variants"])):
        cfg = json.loads(str(g[f"{name}_cfg"]))
        sd = BO.predictor_v3_state_dict(cfg, seed=int(g[f"{name}_seed"]), cif_bias=-0.6)
        pred = CifPredictorV3(**cfg)
        pred.load_state_dict(sd, strict=True)
        pred = pred.to(cuda)
        hidden, lens = _t(g, f"{name}_hidden").to(cuda), _t(g, f"{name}_lens")
        emb, tok, alphas, peaks, _ = pred(hidden, lengths=lens)
        assert tok.tolist() == g[f"{name}_token_num"].tolist(), name
        want = _t(g, f"{name}_embeds")
        assert emb.shape == want.shape and (emb.cpu() - want).abs().max().item() < 5e-5, name
        assert (alphas.cpu() - _t(g, f"{name}_alphas")).abs().max().item() < 5e-6
        assert _fires(peaks.cpu(), 1.0) == _fires(_t(g, f"{name}_peaks"), 1.0)
        _, _, usa, usp = pred.get_upsample_timestamp(hidden, None, tok.round().long(), lengths=lens)
        assert (usa.cpu() - _t(g, f"{name}_us_alphas")).abs().max().item() < 1e-5, name
        assert _fires(usp.cpu()) == _fires(_t(g, f"{name}_us_peaks")), name
        assert (usp.cpu() - _t(g, f"{name}_us_peaks")).abs().max().item() < 1e-4


@pytest.mark.gpu
def test_predictor_halves_with_two_batches_in_flight_are_bitwise_forward(cuda):
    """pf_predictor_alphas_begin / pf_predictor_embeds_slot (CifPredictorV2.forward_begin / forward_finish, V3 through the same
    entry points): begin(A), begin(B), finish(A), finish(B) -- both scan states alive at once -- give what forward(A), forward(B) give"""
    from funasr_amd.cif_predictor import CifPredictorV2, CifPredictorV3
    from oracle import bicif_oracle as BO
    g = _gold()
    name = json.loads(str(g["variants"]))[0]
    cfg = json.loads(str(g[f"{name}_cfg"]))
    sd = BO.predictor_v3_state_dict(cfg, seed=int(g[f"{name}_seed"]), cif_bias=-0.6)
    v3 = CifPredictorV3(**cfg)
    v3.load_state_dict(sd, strict=True)
    v2 = CifPredictorV2(idim=cfg["idim"], l_order=cfg["l_order"], r_order=cfg["r_order"], threshold=cfg.get("threshold", 1.0),
                        tail_threshold=cfg.get("tail_threshold", 0.0))
    v2.load_state_dict({k: v for k, v in sd.items() if k.startswith(("cif_conv1d", "cif_output."))}, strict=True)
    hidden, lens = _t(g, f"{name}_hidden").to(cuda), _t(g, f"{name}_lens")
    A = (hidden, lens)
    Bh = torch.cat([hidden.flip(0) * 1.25, hidden[:1]], 0)[:, : hidden.shape[1] - 2].contiguous()
    Bb = (Bh, [max(1, min(int(n), Bh.shape[1])) for n in list(lens.flip(0).tolist()) + [int(lens[0])]])
    for pred in (v2.to(cuda), v3.to(cuda)):
        want = [pred(h, lengths=n)[:4] for h, n in (A, Bb)]
        for rounds in range(2):                                      # twice: the slots alternate, buffers are reused
            sa, sb = pred.forward_begin(*A), pred.forward_begin(*Bb)
            got = [pred.forward_finish(sa), pred.forward_finish(sb)]
            for w, o in zip(want, got):
                assert w[1].tolist() == o[1].tolist() and o[1].device.type == "cpu"
                for x, y in ((w[0], o[0]), (w[2], o[2]), (w[3], o[3])):
                    assert torch.equal(x, y), type(pred).__name__


@pytest.mark.gpu
def test_bicif_paraformer_text_and_timestamps_equal_reference_inference(cuda):
    """the HIP BiCifParaformer on the features the reference model decoded (oracle/make_golden_bicif.py): same text, same
    token timestamps, in fp32 and in the bf16x3 mode; a single utterance decodes like its row of the batch"""
    from funasr_amd.bicif_paraformer import BiCifParaformer
    from funasr_amd.tokenizer import CharTokenizer
    g = _gold()
    cfg, sd, vocab, want = _e2e_setup(g)
    model = BiCifParaformer.from_config(cfg)
    missing, unexpected = model.load_state_dict(sd, strict=False)
    assert not unexpected and not [k for k in missing if not k.startswith("decoder.embed")], (missing, unexpected)
    model = model.to(cuda)
    tok = CharTokenizer(token_list=vocab, unk_symbol="<unk>")
    feats, lens = _t(g, "e2e_feats").to(cuda), _t(g, "e2e_lens")
    for mode in ("fp32", "bf16x3", "f16x2"):
        model.set_precision(mode)
        res, _ = model.inference(feats, data_lengths=lens, key=[w["key"] for w in want], tokenizer=tok, data_type="fbank")
        for r, w in zip(res, want):
            assert r["key"] == w["key"] and r["text"] == w["text"] and r["timestamp"] == w["timestamp"], (mode, r, w)
    # the chain's halves interleaved across two batches (what AutoModel.inference does with them) == one batch after the other
    model.set_precision("f16x2")
    fa, la, fb, lb = feats, lens, feats[:2].flip(0).contiguous(), lens[:2].flip(0)
    seq = [model.collect(model.enqueue_features(fa, la)), model.collect(model.enqueue_features(fb, lb))]
    ha, hb = model.enqueue_begin(fa, la), model.enqueue_begin(fb, lb)
    pa, pb = model.enqueue_finish(ha), model.enqueue_finish(hb)
    for w, o in zip(seq, (model.collect(pa), model.collect(pb))):
        assert w["ids"] == o["ids"] and torch.equal(w["us_alphas"], o["us_alphas"]) and torch.equal(w["us_peaks"], o["us_peaks"])
    model.set_precision("fp32")
    one, _ = model.inference(feats[:1], data_lengths=lens[:1], key=["utt0"], tokenizer=tok, data_type="fbank")
    assert one[0]["text"] == want[0]["text"] and one[0]["timestamp"] == want[0]["timestamp"]
