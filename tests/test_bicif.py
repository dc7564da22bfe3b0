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
