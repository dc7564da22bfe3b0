"""SeACo-Paraformer (SURVEY §8 f rank 2, the `paraformer-zh` model): the CPU oracle against goldens made by the reference
class (oracle/make_golden_seaco.py), and the HIP path against both."""
import json
import os

import numpy as np
import pytest
import torch

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "seaco.npz")


def _setup():
    from oracle import seaco_oracle as SO
    g = np.load(GOLD, allow_pickle=False)
    cfg = json.loads(str(g["cfg"]))
    sd = SO.seaco_state_dict(cfg, int(g["seed"]), int(g["no_bias"]))
    return g, cfg, sd, json.loads(str(g["vocab"])), json.loads(str(g["hw_list"]))


def test_oracle_equals_reference_inference_with_and_without_hotwords():
    from oracle import seaco_oracle as SO
    from tests.test_bicif import _texts_and_stamps
    g, cfg, sd, vocab, hw_list = _setup()
    feats, lens = torch.from_numpy(g["feats"]), torch.from_numpy(g["lens"])
    for name, hw in (("plain", None), ("hot", hw_list)):
        res = SO.seaco_greedy(feats, lens, hw, sd, cfg, int(g["no_bias"]))
        got = _texts_and_stamps(res["ids"], res["us_alphas"], res["us_peaks"], res["olens"], vocab)
        for (text, stamps), w in zip(got, json.loads(str(g[name]))):
            assert text == w["text"] and stamps == w["timestamp"], name
    assert json.loads(str(g["plain"])) != json.loads(str(g["hot"]))


class _Frontend:
    cmvn_file = None


def test_hotword_list_parsing(tmp_path):
    from funasr_amd.seaco_paraformer import SeacoParaformer, seg_tokenize
    from funasr_amd.tokenizer import CharTokenizer
    g, cfg, sd, vocab, hw_list = _setup()
    tok = CharTokenizer(token_list=vocab, unk_symbol="<unk>")
    seg = {ch: ch for ch in vocab[3:-10]}
    seg.update({"hello": "hel@@ lo", "world": "wor@@ ld", "the": "the"})
    assert seg_tokenize(["我们"], seg) == ["我", "们"] and seg_tokenize(["Hello"], seg) == ["hel@@", "lo"]
    assert seg_tokenize(["x1"], seg) == ["<unk>"] and seg_tokenize(["我x"], seg) == ["<unk>"]
    m = SeacoParaformer.__new__(SeacoParaformer)
    m.sos = 1
    fe = _Frontend()
    assert m.generate_hotwords_list(None, tokenizer=tok, frontend=fe) is None
    with open(tmp_path / "seg_dict", "w", encoding="utf-8") as f:
        f.write("".join(f"{k} {v}\n" for k, v in seg.items()))
    fe.cmvn_file = str(tmp_path / "am.mvn")
    assert m.generate_hotwords_list(str(g["hotwords"]), tokenizer=tok, frontend=fe) == hw_list
    (tmp_path / "hot.txt").write_text("\n".join(str(g["hotwords"]).split()) + "\n", encoding="utf-8")
    assert m.generate_hotwords_list(str(tmp_path / "hot.txt"), tokenizer=tok, frontend=fe) == hw_list
    fe.cmvn_file = None                                           # no seg_dict: a hotword is one token (or <unk>)
    assert m.generate_hotwords_list("我 hello", tokenizer=tok, frontend=fe) == [[tok.tokens2ids(["我"])[0]], tok.tokens2ids(["hello"]), [1]]


def _build(cfg):
    from funasr_amd.seaco_paraformer import SeacoParaformer
    ec = dict(cfg["encoder"])
    input_size = ec.pop("input_size")
    dc = dict(cfg["decoder"])
    vocab = dc.pop("vocab_size")
    dc.pop("encoder_output_size", None)
    sc = dict(cfg["seaco_decoder"])
    for k in ("vocab_size", "encoder_output_size", "att_layer_num"):
        sc.pop(k, None)
    return SeacoParaformer(encoder="SANMEncoder", encoder_conf=dict(ec, input_layer="pe"), decoder="ParaformerSANMDecoder",
                           decoder_conf=dc, seaco_decoder="ParaformerSANMDecoder",
                           seaco_decoder_conf=dict(sc, use_output_layer=False, wo_input_layer=True), predictor="CifPredictorV3",
                           predictor_conf=dict(cfg["predictor"]), ctc_weight=0.0, input_size=input_size, vocab_size=vocab,
                           inner_dim=512, bias_encoder_type="lstm", NO_BIAS=5)


def test_state_dict_layout_matches_the_reference_checkpoint():
    """every parameter of the reference SeacoParaformer (key and shape, dumped by oracle/make_golden_seaco.py) exists here
    under the same name, and nothing else: a published model.pt loads with strict=True"""
    g, cfg, sd, vocab, hw_list = _setup()
    model = _build(cfg)
    missing, unexpected = model.load_state_dict(sd, strict=False)
    assert not missing and not unexpected, (missing, unexpected)
    ref = json.loads(str(g["ref_state_dict"]))
    mine = {k: list(v.shape) for k, v in model.state_dict().items()}
    assert mine == ref, (sorted(set(mine) ^ set(ref))[:10], [k for k in mine if k in ref and mine[k] != ref[k]][:10])


@pytest.mark.gpu
def test_seaco_on_the_gpu_equals_reference_inference(cuda, tmp_path):
    from funasr_amd.tokenizer import CharTokenizer
    g, cfg, sd, vocab, hw_list = _setup()
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
    for mode in ("fp32", "bf16x3", "f16x2"):
        model.set_precision(mode)
        for name, hw in (("plain", None), ("hot", str(g["hotwords"]))):
            res, _ = model.inference(feats, data_lengths=lens, key=keys, tokenizer=tok, frontend=fe, data_type="fbank", hotword=hw)
            for r, w in zip(res, json.loads(str(g[name]))):
                assert r["text"] == w["text"] and r["timestamp"] == w["timestamp"], (mode, name, r, w)


@pytest.mark.gpu
def test_attention_score_filter_for_more_than_50_hotwords_equals_reference(cuda, tmp_path):
    """58 hotwords + the no-bias entry: `_seaco_decode_with_ASF` first ranks them by the bias decoder's block-5 attention
    (sequence 0, summed over heads and token positions; seaco_paraformer/model.py:323-335, decoder.py:485-513) and keeps the
    top 50. Fixture from the REFERENCE class (oracle/make_golden_seaco_asf.py): the filter's scores and the final texts /
    timestamps."""
    from funasr_amd.tokenizer import CharTokenizer
    from oracle import seaco_oracle as SO
    g = np.load(os.path.join(os.path.dirname(GOLD), "seaco_asf.npz"), allow_pickle=False)
    cfg = json.loads(str(g["cfg"]))
    vocab = json.loads(str(g["vocab"]))
    sd = SO.seaco_state_dict(cfg, int(g["seed"]), int(g["no_bias"]))
    model = _build(cfg)
    model.load_state_dict(sd, strict=True)
    model = model.to(cuda)
    tok = CharTokenizer(token_list=vocab, unk_symbol="<unk>")
    feats, lens = torch.from_numpy(g["feats"]).to(cuda), torch.from_numpy(g["lens"])
    with open(tmp_path / "seg_dict", "w", encoding="utf-8") as f:
        f.write("".join(f"{ch} {ch}\n" for ch in vocab[3:-10]))
    fe = _Frontend()
    fe.cmvn_file = str(tmp_path / "am.mvn")
    keys = [f"utt{b}" for b in range(3)]
    # (1) the filter's scores themselves
    hw_list = model.generate_hotwords_list(str(g["hotwords"]), tokenizer=tok, frontend=fe)
    assert hw_list == json.loads(str(g["hw_list"])) and len(hw_list) == 59
    enc, olens = model.encode(feats, lens)
    embeds, token_num, _, _ = model.calc_predictor(enc, olens)
    tk = [int(round(v)) for v in token_num.tolist()]
    _, _, dec_hidden, _ = model.decoder._run(enc, olens, embeds, tk, want_logits=False, want_ids=False, want_hidden=True)
    sel = model._hotword_representation(hw_list)
    ctx = sel[None].expand(3, -1, -1).contiguous()
    scores = model.seaco_decoder.forward_asf6(ctx, [59] * 3, dec_hidden, tk).cpu()
    ref = torch.from_numpy(g["asf_scores"])
    assert (scores - ref).abs().max().item() < 2e-3 * ref.abs().max().item()
    assert torch.topk(scores, 50)[1].tolist() == torch.topk(ref, 50)[1].tolist()
    # (2) end to end through inference()
    res, _ = model.inference(feats, data_lengths=lens, key=keys, tokenizer=tok, frontend=fe, data_type="fbank", hotword=str(g["hotwords"]))
    for r, w in zip(res, json.loads(str(g["hot"]))):
        assert r["text"] == w["text"] and r["timestamp"] == w["timestamp"], (r, w)


def test_model_directory_builds_through_automodel(tmp_path):
    from funasr_amd.auto_model import AutoModel
    from tests._model_dir import make_seaco_model_dir
    info = make_seaco_model_dir(str(tmp_path / "seaco"))
    am = AutoModel(model=str(tmp_path / "seaco"), device="cpu")
    assert type(am.model).__name__ == "SeacoParaformer" and type(am.model.predictor).__name__ == "CifPredictorV3"
    assert am.model.NO_BIAS == info["no_bias"] and am.model.seaco_decoder.kernel_size == 21 and am.model.seaco_decoder.output_layer is None
    assert am.kwargs["frontend"].cmvn_file.endswith("am.mvn")


@pytest.mark.gpu
def test_automodel_wav_to_text_with_hotwords_equals_the_oracle(cuda, tmp_path):
    """wav -> AutoModel(model=<SeACo dir>).generate(hotword=...) -> text + timestamps, against the CPU oracle run on the same
    waveforms (frontend + encoder + CifPredictorV3 + both decoders + timestamp head), batch of ragged clips"""
    from funasr_amd import synth
    from funasr_amd.auto_model import AutoModel
    from oracle import seaco_oracle as SO
    from tests._model_dir import VOCAB, make_seaco_model_dir
    from tests.test_bicif import _texts_and_stamps
    d = str(tmp_path / "seaco")
    info = make_seaco_model_dir(d)
    am = AutoModel(model=d, device="cuda:0")
    waves = [synth.speech_like(n, seed=90 + i) for i, n in enumerate((40000, 31000, 22000))]
    hot = "我们 world 地"
    res = am.generate(input=[w.numpy() for w in waves], batch_size=3, hotword=hot)
    plain = am.generate(input=[w.numpy() for w in waves], batch_size=3)
    # the oracle's network on the features of the HIP frontend (the frontends differ by fp32 FFT round-off, pinned elsewhere)
    wav = torch.nn.utils.rnn.pad_sequence(waves, batch_first=True).to(cuda)
    feats, flens = am.kwargs["frontend"](wav, [w.numel() for w in waves])
    feats, flens = feats.cpu(), flens.cpu()
    hw_list = am.model.generate_hotwords_list(hot, tokenizer=am.kwargs["tokenizer"], frontend=am.kwargs["frontend"])
    assert [len(h) for h in hw_list] == [2, 2, 1, 1]
    for out, hw in ((res, hw_list), (plain, None)):
        ref = SO.seaco_greedy(feats, flens, hw, info["sd"], info["cfg"], info["no_bias"])
        want = _texts_and_stamps(ref["ids"], ref["us_alphas"], ref["us_peaks"], ref["olens"], VOCAB)
        for r, (text, stamps) in zip(out, want):
            assert r["text"] == text and r["timestamp"] == stamps
    assert [r["text"] for r in res] != [r["text"] for r in plain]
    # more inputs than one batch: the overlapped loop (two parts for this class: everything enqueued, text later) == the plain loop
    many = [w.numpy() for w in waves] + [waves[0][:20000].numpy(), waves[1][:27000].numpy()]
    strip = lambda out: [(r["text"], r["timestamp"]) for r in out]
    assert strip(am.generate(input=many, batch_size=2, hotword=hot)) == strip(am.generate(input=many, batch_size=2, hotword=hot, pipeline=False))
