"""ContextualParaformer (CLAS hotword biasing; funasr_amd/contextual_paraformer.py) against the fixture recorded from the
REFERENCE classes' own `inference` (tests/golden/contextual.npz, oracle/make_golden_contextual.py): without hotwords, with
hotwords, with clas_scale 0.6; plus the state_dict layout of the reference model."""
import json
import os

import numpy as np
import pytest
import torch

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "contextual.npz")


class _Frontend:
    fs, frame_shift, lfr_n = 16000, 10, 6
    cmvn_file = None


def _setup():
    from oracle.make_golden_contextual import contextual_state_dict
    g = np.load(GOLD, allow_pickle=False)
    cfg = json.loads(str(g["cfg"]))
    return g, cfg, contextual_state_dict(cfg, int(g["seed"])), json.loads(str(g["vocab"]))


def _build(cfg):
    from funasr_amd.contextual_paraformer import ContextualParaformer
    ec = dict(cfg["encoder"])
    input_size = ec.pop("input_size")
    dc = dict(cfg["decoder"])
    vocab = dc.pop("vocab_size")
    dc.pop("encoder_output_size", None)
    return ContextualParaformer(encoder="SANMEncoder", encoder_conf=dict(ec, input_layer="pe"), decoder="ContextualParaformerDecoder",
                                decoder_conf=dc, predictor="CifPredictorV2", predictor_conf=dict(cfg["predictor"]), ctc_weight=0.0,
                                input_size=input_size, vocab_size=vocab, inner_dim=512, bias_encoder_type="lstm")


def test_state_dict_layout_matches_the_reference_model():
    g, cfg, sd, vocab = _setup()
    model = _build(cfg)
    missing, unexpected = model.load_state_dict(sd, strict=False)
    assert not missing and not unexpected, (missing, unexpected)
    ref = json.loads(str(g["ref_state_dict"]))
    mine = {k: list(v.shape) for k, v in model.state_dict().items()}
    assert mine == ref, (sorted(set(mine) ^ set(ref))[:10], [k for k in mine if k in ref and mine[k] != ref[k]][:10])


def test_registered_under_the_reference_keys():
    from funasr_amd import install as inst
    pairs = {(t, k) for t, k, _ in inst.hip_classes()}
    assert ("model_classes", "ContextualParaformer") in pairs and ("decoder_classes", "ContextualParaformerDecoder") in pairs


@pytest.mark.gpu
def test_contextual_paraformer_on_the_gpu_equals_reference_inference(cuda, tmp_path):
    from funasr_amd.tokenizer import CharTokenizer
    g, cfg, sd, vocab = _setup()
    model = _build(cfg)
    model.load_state_dict(sd, strict=True)
    model = model.to(cuda)
    tok = CharTokenizer(token_list=vocab, unk_symbol="<unk>")
    feats, lens = torch.from_numpy(g["feats"]).to(cuda), torch.from_numpy(g["lens"])
    seg = {ch: ch for ch in vocab[3:-10]}
    seg.update({"hello": "hel@@ lo", "world": "wor@@ ld", "the": "the"})
    with open(tmp_path / "seg_dict", "w", encoding="utf-8") as f:
        f.write("".join(f"{k} {v}\n" for k, v in seg.items()))
    fe = _Frontend()
    fe.cmvn_file = str(tmp_path / "am.mvn")
    keys = [f"utt{b}" for b in range(3)]
    assert model.generate_hotwords_list(str(g["hotwords"]), tokenizer=tok, frontend=fe) == json.loads(str(g["hw_list"]))
    for name, kw in (("plain", dict()), ("hot", dict(hotword=str(g["hotwords"]))),
                     ("hot_scaled", dict(hotword=str(g["hotwords"]), clas_scale=0.6))):
        res, _ = model.inference(feats, data_lengths=lens, key=keys, tokenizer=tok, frontend=fe, data_type="fbank", **kw)
        want = json.loads(str(g[name]))
        assert [r["text"] for r in res] == [w["text"] for w in want], (name, res, want)
        assert [r["key"] for r in res] == [w["key"] for w in want]


@pytest.mark.gpu
def test_contextual_beam_search_on_the_gpu_equals_reference_inference(cuda, tmp_path):
    """The CTC-rescored beam search over the hotword-biased decoder scores (contextual_paraformer/model.py:408-415,467-494;
    init_beam_search paraformer/model.py:482-532) against the n-best texts the REFERENCE classes' own `inference` produced
    (tests/golden/contextual_beam.npz, oracle/make_golden_contextual_beam.py): with and without hotwords, 2-best of beam 3."""
    from funasr_amd.contextual_paraformer import ContextualParaformer
    from funasr_amd.tokenizer import CharTokenizer
    from oracle.make_golden_contextual import contextual_state_dict
    from oracle.make_golden_contextual_beam import ctc_state_dict
    g = np.load(os.path.join(os.path.dirname(GOLD), "contextual_beam.npz"), allow_pickle=False)
    cfg, vocab = json.loads(str(g["cfg"])), json.loads(str(g["vocab"]))
    sd = contextual_state_dict(cfg, int(g["seed"]))
    sd.update(ctc_state_dict(cfg, int(g["seed"])))
    ec = dict(cfg["encoder"]); input_size = ec.pop("input_size")
    dc = dict(cfg["decoder"]); V = dc.pop("vocab_size"); dc.pop("encoder_output_size", None)
    tok = CharTokenizer(token_list=vocab, unk_symbol="<unk>")
    feats, lens = torch.from_numpy(g["feats"]).to(cuda), torch.from_numpy(g["lens"])
    seg = {ch: ch for ch in vocab[3:-10]}
    seg.update({"hello": "hel@@ lo", "world": "wor@@ ld", "the": "the"})
    with open(tmp_path / "seg_dict", "w", encoding="utf-8") as f:
        f.write("".join(f"{k} {v}\n" for k, v in seg.items()))
    fe = _Frontend()
    fe.cmvn_file = str(tmp_path / "am.mvn")
    keys = [f"utt{b}" for b in range(3)]
    beam_kw = json.loads(str(g["beam_kw"]))
    for name, kw in (("beam_hot", dict(hotword=str(g["hotwords"]))), ("beam_plain", dict())):
        model = ContextualParaformer(encoder="SANMEncoder", encoder_conf=dict(ec, input_layer="pe"), decoder="ContextualParaformerDecoder",
                                     decoder_conf=dc, predictor="CifPredictorV2", predictor_conf=dict(cfg["predictor"]), ctc_weight=0.3,
                                     input_size=input_size, vocab_size=V, inner_dim=512, bias_encoder_type="lstm")
        model.load_state_dict(sd, strict=True)
        model = model.to(cuda)
        res, _ = model.inference(feats, data_lengths=lens, key=keys, tokenizer=tok, frontend=fe, data_type="fbank", token_list=vocab,
                                 **beam_kw, **kw)
        want = json.loads(str(g[name]))
        assert [(r["key"], r["text"]) for r in res] == [(w["key"], w["text"]) for w in want], (name, res, want)


def _decoder_case(tag):
    from oracle.make_golden_contextual_decoder import decoder_weights
    g = np.load(os.path.join(os.path.dirname(GOLD), "contextual_decoder.npz"), allow_pickle=False)
    dc = json.loads(str(g[f"{tag}_cfg"]))
    t = lambda k: torch.from_numpy(g[f"{tag}_{k}"])                                  # noqa: E731
    return dc, decoder_weights(dc, int(g[f"{tag}_seed"])), t, float(g[f"{tag}_scale"])


@pytest.mark.parametrize("tag", ["a", "b", "c"])
def test_oracle_contextual_decoder_equals_reference_logits(tag):
    """oracle/paraformer_oracle.py contextual_decoder against logits of the reference's own ContextualParaformerDecoder
    (contextual_paraformer/decoder.py:293-352; oracle/make_golden_contextual_decoder.py): hotword rows, clas_scale 0.6, decoders2 behind the fusion."""
    from oracle import paraformer_oracle as O
    dc, sd, t, scale = _decoder_case(tag)
    logits = O.contextual_decoder(t("memory"), t("mem_lens"), t("embeds"), t("tok_lens"), t("hot"), sd, dc, clas_scale=scale)
    ref = t("logits")
    for b, n in enumerate(t("tok_lens").tolist()):
        assert (logits[b, :n] - ref[b, :n]).abs().max().item() < 2e-5, (tag, b)


@pytest.mark.gpu
@pytest.mark.parametrize("tag", ["a", "b", "c"])
def test_contextual_decoder_on_the_gpu_equals_reference_logits(cuda, tag):
    """pf_decoder_forward_contextual (last_decoder without its cross-attention residual, hotword branch, 1 x 1 fusion, decoders2 behind it
    in case c) against the same reference logits, and the fused arg-max route against their arg-max."""
    from funasr_amd.contextual_paraformer import ContextualParaformerDecoder
    dc, sd, t, scale = _decoder_case(tag)
    d = ContextualParaformerDecoder(**dc)
    missing, unexpected = d.load_state_dict(sd, strict=False)
    assert not missing and not unexpected, (missing, unexpected)
    d = d.to(cuda)
    args = (t("memory").to(cuda), t("mem_lens"), t("embeds").to(cuda), t("tok_lens"))
    logits, olens = d(*args, contextual_info=t("hot").to(cuda), clas_scale=scale)
    ref = t("logits")
    assert olens.tolist() == t("tok_lens").tolist()
    for b, n in enumerate(t("tok_lens").tolist()):
        assert (logits[b, :n].cpu() - ref[b, :n]).abs().max().item() < 2e-4, (tag, b)
    ids, _ = d.greedy(*args, contextual_info=t("hot").to(cuda), clas_scale=scale)
    for b, n in enumerate(t("tok_lens").tolist()):
        assert ids[b, :n].cpu().tolist() == ref[b, :n].argmax(-1).tolist()
