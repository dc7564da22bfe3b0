"""Differential test of `BiCifParaformer.inference` (text + token timestamps on every call): this package's
class against the REFERENCE's own (funasr/models/bicif_paraformer/model.py:275-400, run for real on the CPU) on random ragged
batches. The product's device half is stood in by the CPU oracle (oracle/bicif_oracle.py), so what is compared is the host
glue: token filtering, the upsampled-CIF -> token-span conversion with `begin_time`, `sentence_postprocess` with stamps, key
handling, the records, and what comes back when nothing was decoded. tests/golden/bicif.npz pins one batch of the same class;
this sweeps. Build container only."""
import copy
import json

import pytest
import torch

from oracle import bicif_oracle as BO
from oracle import make_golden_bicif as G
from oracle import ref_import

pytestmark = pytest.mark.skipif(not ref_import.available(), reason="reference checkout not present (GPU box)")


def _reference_model(cfg, sd):
    ref_import.install()
    import funasr.models.sanm.encoder  # noqa: F401
    import funasr.models.paraformer.decoder  # noqa: F401
    import funasr.models.bicif_paraformer.cif_predictor  # noqa: F401
    from funasr.models.bicif_paraformer.model import BiCifParaformer
    ec, dc = cfg["encoder"], cfg["decoder"]
    model = BiCifParaformer(
        encoder="SANMEncoder",
        encoder_conf=dict(output_size=ec["output_size"], attention_heads=ec["attention_heads"], linear_units=ec["linear_units"],
                          num_blocks=ec["num_blocks"], input_layer="pe", pos_enc_class="SinusoidalPositionEncoder",
                          normalize_before=True, kernel_size=ec["kernel_size"], sanm_shfit=ec["sanm_shfit"],
                          selfattention_layer_type="sanm"),
        decoder="ParaformerSANMDecoder",
        decoder_conf=dict(attention_heads=dc["attention_heads"], linear_units=dc["linear_units"], num_blocks=dc["num_blocks"],
                          att_layer_num=dc["att_layer_num"], kernel_size=dc["kernel_size"], sanm_shfit=dc["sanm_shfit"]),
        predictor="CifPredictorV3", predictor_conf=dict(cfg["predictor"]), input_size=ec["input_size"], vocab_size=len(G.VOCAB),
        ctc_weight=0.0).eval()
    missing, unexpected = model.load_state_dict(sd, strict=False)
    assert not unexpected and all(k.startswith(("decoder.embed", "criterion")) for k in missing), (missing, unexpected)
    with torch.no_grad():
        model.decoder.embed[0].weight.zero_()
    return model


def test_bicif_inference_equals_the_reference(monkeypatch):
    ref_import.install()
    from funasr.tokenizer.char_tokenizer import CharTokenizer as RefTok
    from funasr_amd.bicif_paraformer import BiCifParaformer
    from funasr_amd.tokenizer import CharTokenizer
    cfg = G.model_config(enc_blocks=2, dec_blocks=1)
    rtok, tok = RefTok(token_list=G.VOCAB, unk_symbol="<unk>"), CharTokenizer(token_list=G.VOCAB, unk_symbol="<unk>")
    g = torch.Generator().manual_seed(3)
    compared = empty = 0
    for trial in range(24):
        sd = G.model_state_dict(cfg, 200 + trial)
        if trial % 8 == 7:                                              # a batch that predicts no token
            sd["predictor.cif_output.bias"] = sd["predictor.cif_output.bias"] - 12.0
        ref = _reference_model(cfg, sd)
        ours = BiCifParaformer.from_config(cfg)
        ours.load_state_dict(sd, strict=False)

        def recognize_features(speech, speech_lengths, return_intermediate=False, _sd=sd):
            lens = torch.as_tensor(speech_lengths, dtype=torch.int32).reshape(-1)
            r = BO.bicif_greedy(speech.float(), lens, _sd, cfg)
            out = dict(token_num=[int(v) for v in r["token_num"].tolist()], raw_ids=r["raw_ids"], ids=r["ids"], olens=r["olens"],
                       us_alphas=r["us_alphas"], us_peaks=r["us_peaks"], alphas=r["alphas"], peaks=r["peaks"], enc=r["enc"])
            if r["us_alphas"] is not None:
                out.update(us_alphas_host=r["us_alphas"], us_peaks_host=r["us_peaks"], olens_host=[int(v) for v in r["olens"].tolist()])
            return out

        monkeypatch.setattr(ours, "recognize_features", recognize_features)
        B = int(torch.randint(1, 4, (1,), generator=g))
        T = int(torch.randint(8, 50, (1,), generator=g))
        lens = torch.randint(4, T + 1, (B,), generator=g, dtype=torch.int32)
        lens[int(torch.randint(0, B, (1,), generator=g))] = T
        feats = torch.randn(B, T, 560, generator=g) * 0.7
        for b in range(B):
            feats[b, lens[b]:] = 0
        kw = dict(device="cpu")
        if trial % 3 == 1:
            kw["begin_time"] = int(torch.randint(0, 5000, (1,), generator=g))
        keys = [f"u{b}" for b in range(B)]          # (this class does not unwrap a list-of-lists key like Paraformer.inference does)
        waves = [torch.zeros(int(l) * 960) for l in lens]
        with torch.no_grad():
            want = ref.inference([w.clone() for w in waves], key=copy.deepcopy(keys), tokenizer=rtok, frontend=G._Frontend(feats.clone(), lens.clone().long()), **copy.deepcopy(kw))
        got = ours.inference([w.clone() for w in waves], key=copy.deepcopy(keys), tokenizer=tok, frontend=G._Frontend(feats.clone(), lens.clone()), **copy.deepcopy(kw))
        if isinstance(want, list):
            assert want == [] and got == [], (trial, got)
            empty += 1
            continue
        assert json.loads(json.dumps(got[0])) == json.loads(json.dumps(want[0])), (trial, kw, got[0], want[0])
        compared += len(want[0])
    assert compared > 25 and empty >= 2
