"""Differential test of `SeacoParaformer.inference` (the `paraformer-zh` model): this package's class against the
REFERENCE's own (funasr/models/seaco_paraformer/model.py, run for real on the CPU) on random ragged batches with random hotword
strings (none, single- and multi-token words, unknown words, many words). The product's device half is stood in by the CPU
oracle (oracle/seaco_oracle.py) fed with the hotword id list THIS package's `generate_hotwords_list` made, so what is compared
is the host side: hotword parsing / seg_dict expansion, token filtering, timestamps, the records, the empty return.
tests/golden/seaco.npz pins two calls of the same class; this sweeps. Build container only."""
import copy
import json
import os

import pytest
import torch

from oracle import make_golden_bicif as MB
from oracle import make_golden_seaco as GS
from oracle import ref_import
from oracle import seaco_oracle as SO

pytestmark = pytest.mark.skipif(not ref_import.available(), reason="reference checkout not present (GPU box)")
NO_BIAS = GS.NO_BIAS


def _reference_model(cfg, sd):
    ref_import.install()
    import funasr.models.sanm.encoder  # noqa: F401
    import funasr.models.paraformer.decoder  # noqa: F401
    import funasr.models.bicif_paraformer.cif_predictor  # noqa: F401
    from funasr.models.seaco_paraformer.model import SeacoParaformer
    ec, dc, sc = cfg["encoder"], cfg["decoder"], cfg["seaco_decoder"]
    model = SeacoParaformer(
        encoder="SANMEncoder",
        encoder_conf=dict(output_size=512, attention_heads=4, linear_units=ec["linear_units"], num_blocks=ec["num_blocks"],
                          input_layer="pe", pos_enc_class="SinusoidalPositionEncoder", normalize_before=True,
                          kernel_size=11, sanm_shfit=0, selfattention_layer_type="sanm"),
        decoder="ParaformerSANMDecoder",
        decoder_conf=dict(attention_heads=4, linear_units=dc["linear_units"], num_blocks=dc["num_blocks"],
                          att_layer_num=dc["att_layer_num"], kernel_size=11, sanm_shfit=0),
        seaco_decoder="ParaformerSANMDecoder",
        seaco_decoder_conf=dict(attention_heads=4, linear_units=sc["linear_units"], num_blocks=sc["num_blocks"], kernel_size=21,
                                sanm_shfit=0, use_output_layer=False, wo_input_layer=True),
        predictor="CifPredictorV3", predictor_conf=dict(MB.V3), input_size=560, vocab_size=len(MB.VOCAB), ctc_weight=0.0,
        inner_dim=512, bias_encoder_type="lstm", NO_BIAS=NO_BIAS).eval()
    missing, unexpected = model.load_state_dict(sd, strict=False)
    assert not unexpected and all(k.startswith("criterion") for k in missing), (missing, unexpected)
    return model


def test_seaco_inference_equals_the_reference(monkeypatch, tmp_path):
    ref_import.install()
    from funasr.tokenizer.char_tokenizer import CharTokenizer as RefTok
    from funasr_amd.tokenizer import CharTokenizer
    from tests.test_seaco import _build
    cfg = GS.model_config()
    rtok, tok = RefTok(token_list=MB.VOCAB, unk_symbol="<unk>"), CharTokenizer(token_list=MB.VOCAB, unk_symbol="<unk>")
    with open(tmp_path / "seg_dict", "w", encoding="utf-8") as f:       # next to the cmvn file: multi-token hotwords (:616-623)
        for ch in MB.VOCAB[3:-10]:
            f.write(f"{ch} {ch}\n")
        f.write("hello hel@@ lo\nworld wor@@ ld\nthe the\n")
    words = [w for w in MB.VOCAB[3:-10]] + ["hello", "world", "the", "我们", "大地", "zzz", "国家大"]
    g = torch.Generator().manual_seed(5)
    compared = empty = with_hot = 0
    for trial in range(20):
        sd = SO.seaco_state_dict(cfg, 400 + trial, NO_BIAS)
        if trial % 10 == 9:
            sd["predictor.cif_output.bias"] = sd["predictor.cif_output.bias"] - 12.0          # no token at all
        ref = _reference_model(cfg, sd)
        ours = _build(cfg)
        ours.load_state_dict(sd, strict=False)
        assert ours.NO_BIAS == NO_BIAS

        def recognize_features(speech, speech_lengths, return_intermediate=False, _sd=sd, _m=ours):
            from oracle import bicif_oracle as BO
            from oracle import paraformer_oracle as O
            lens = torch.as_tensor(speech_lengths, dtype=torch.int32).reshape(-1)
            enc, olens = O.sanm_encoder(speech.float(), lens, _sd, cfg["encoder"], "encoder.")
            tok0 = BO.predictor_v3(enc, olens, _sd, cfg["predictor"], "predictor.")[1].round().long()
            if int(tok0.max()) < 1:
                B0 = speech.shape[0]
                return dict(token_num=[0] * B0, raw_ids=[[] for _ in range(B0)], ids=[[] for _ in range(B0)])
            r = SO.seaco_greedy(speech.float(), lens, _m.hotword_list, _sd, cfg, NO_BIAS)
            out = dict(token_num=[int(v) for v in r["token_num"].tolist()], raw_ids=r["raw_ids"], ids=r["ids"], olens=r["olens"],
                       us_alphas=r["us_alphas"], us_peaks=r["us_peaks"], enc=r["enc"])
            if max(out["token_num"]) >= 1:
                out.update(us_alphas_host=r["us_alphas"], us_peaks_host=r["us_peaks"], olens_host=[int(v) for v in r["olens"].tolist()])
            return out

        monkeypatch.setattr(ours, "recognize_features", recognize_features)
        B = int(torch.randint(1, 4, (1,), generator=g))
        T = int(torch.randint(8, 45, (1,), generator=g))
        lens = torch.randint(4, T + 1, (B,), generator=g, dtype=torch.int32)
        lens[int(torch.randint(0, B, (1,), generator=g))] = T
        feats = torch.randn(B, T, 560, generator=g) * 0.7
        for b in range(B):
            feats[b, lens[b]:] = 0
        kw = dict(device="cpu")
        n_hot = [0, 1, 3, 6][trial % 4]
        if n_hot:
            kw["hotword"] = " ".join(words[int(i)] for i in torch.randint(0, len(words), (n_hot,), generator=g))
            with_hot += 1
        keys = [f"u{b}" for b in range(B)]
        waves = [torch.zeros(int(l) * 960) for l in lens]
        r_fe, o_fe = MB._Frontend(feats.clone(), lens.clone().long()), MB._Frontend(feats.clone(), lens.clone())
        r_fe.cmvn_file = o_fe.cmvn_file = os.path.join(str(tmp_path), "am.mvn")
        with torch.no_grad():
            want = ref.inference([w.clone() for w in waves], key=copy.deepcopy(keys), tokenizer=rtok, frontend=r_fe, **copy.deepcopy(kw))
        got = ours.inference([w.clone() for w in waves], key=copy.deepcopy(keys), tokenizer=tok, frontend=o_fe, **copy.deepcopy(kw))
        assert ours.hotword_list == ref.hotword_list, (trial, kw, ours.hotword_list, ref.hotword_list)
        if len(want) == 1:                                              # nothing decoded: the reference returns ([],)
            assert want == ([],) and got == ([],), (trial, got)
            empty += 1
            continue
        assert json.loads(json.dumps(got[0])) == json.loads(json.dumps(want[0])), (trial, kw, got[0], want[0])
        compared += len(want[0])
    assert compared > 20 and empty >= 1 and with_hot >= 10
