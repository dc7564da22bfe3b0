"""Differential test of the flagship class's own `inference` glue: this package's `Paraformer.inference` against
the REFERENCE's `Paraformer.inference` (funasr/models/paraformer/model.py:534-697, imported from /root/reference and run for
real on the CPU) on random batches -- greedy and CTC-rescored beam routes, with / without a tokenizer, `pred_timestamp`,
feature and sound inputs, key forms, batches that predict no token at all. The product's device half
(`recognize_features` / `recognize_features_beam`) is stood in by the CPU oracle (tests may use it), so what is compared is
everything the host does around it: hypothesis assembly, sos / eos / blank filtering, tokenizer calls, `sentence_postprocess`,
token timestamps, the records and what is returned when nothing was decoded. Build container only."""
import copy

import pytest
import torch

from funasr_amd import synth
from oracle import paraformer_oracle as O
from oracle import ref_import

pytestmark = pytest.mark.skipif(not ref_import.available(), reason="reference checkout not present (GPU box)")

VOCAB = ["<blank>", "<s>", "</s>"] + list("今天气真不错我们一起去公园散步吧欢迎大家") + ["hello", "world", "a", "b", "ok", "the", "<unk>"]


class _Frontend:
    """hands the given LFR features to extract_fbank (funasr/utils/load_utils.py:381-419) / to this package's inference"""
    fs, frame_shift, lfr_n = 16000, 10, 6

    def __init__(self, feats, lens):
        self.feats, self.lens = feats, lens

    def __call__(self, data, data_len, **kwargs):
        return self.feats, self.lens


def _config(with_ctc):
    cfg = synth.tiny(synth.PARAFORMER_LARGE, enc_blocks=2, dec_blocks=1, vocab=len(VOCAB))
    return cfg


def _reference_model(cfg, sd, with_ctc):
    ref_import.install()
    import funasr.models.sanm.encoder  # noqa: F401
    import funasr.models.paraformer.decoder  # noqa: F401
    import funasr.models.paraformer.cif_predictor  # noqa: F401
    from funasr.models.paraformer.model import Paraformer
    ec, dc, pc = cfg["encoder"], cfg["decoder"], cfg["predictor"]
    model = Paraformer(
        encoder="SANMEncoder",
        encoder_conf=dict(output_size=ec["output_size"], attention_heads=ec["attention_heads"], linear_units=ec["linear_units"],
                          num_blocks=ec["num_blocks"], input_layer="pe", pos_enc_class="SinusoidalPositionEncoder",
                          normalize_before=True, kernel_size=ec["kernel_size"], sanm_shfit=ec["sanm_shfit"],
                          selfattention_layer_type="sanm"),
        decoder="ParaformerSANMDecoder",
        decoder_conf=dict(attention_heads=dc["attention_heads"], linear_units=dc["linear_units"], num_blocks=dc["num_blocks"],
                          att_layer_num=dc["att_layer_num"], kernel_size=dc["kernel_size"], sanm_shfit=dc["sanm_shfit"]),
        predictor="CifPredictorV2",
        predictor_conf=dict(idim=pc["idim"], threshold=pc["threshold"], l_order=pc["l_order"], r_order=pc["r_order"],
                            tail_threshold=pc["tail_threshold"]),
        ctc_weight=0.3 if with_ctc else 0.0, input_size=560, vocab_size=len(VOCAB), predictor_bias=1, sampling_ratio=0.75).eval()
    sdr = dict(sd)
    sdr["decoder.embed.0.weight"] = torch.zeros(len(VOCAB), 512)
    missing, unexpected = model.load_state_dict(sdr, strict=False)
    assert not unexpected and all(k.startswith("criterion") or "sampler" in k for k in missing), (missing, unexpected)
    return model


def _our_model(cfg, sd, with_ctc, monkeypatch):
    """this package's Paraformer with its device half computed by the oracle on the CPU"""
    from funasr_amd.paraformer import Paraformer
    ec = dict(cfg["encoder"])
    input_size = ec.pop("input_size")
    dc = dict(cfg["decoder"])
    vocab = dc.pop("vocab_size")
    dc.pop("encoder_output_size", None)
    model = Paraformer(encoder="SANMEncoder", encoder_conf=dict(ec, input_layer="pe"), decoder="ParaformerSANMDecoder", decoder_conf=dc,
                       predictor="CifPredictorV2", predictor_conf=dict(cfg["predictor"]), ctc_weight=0.3 if with_ctc else 0.0,
                       input_size=input_size, vocab_size=vocab)
    model.load_state_dict({k: v for k, v in sd.items()}, strict=False)

    def recognize_features(speech, speech_lengths, return_intermediate=False):
        lens = torch.as_tensor(speech_lengths, dtype=torch.int32).reshape(-1)
        r = O.paraformer_greedy(speech.float(), lens, sd, cfg)
        tok = [int(v) for v in r["token_num"].tolist()]
        out = dict(token_num=tok, raw_ids=r["raw_ids"], ids=r["ids"])
        if return_intermediate:
            out.update(enc=r["enc"], olens=r["olens"], embeds=r["embeds"], alphas=r["alphas"], peaks=r["peaks"])
        return out

    def recognize_features_beam(speech, speech_lengths, maxlenratio=0.0, minlenratio=0.0, return_intermediate=False):
        lens = torch.as_tensor(speech_lengths, dtype=torch.int32).reshape(-1)
        r = O.paraformer_greedy(speech.float(), lens, sd, cfg)
        tok = [int(v) for v in r["token_num"].tolist()]
        B = speech.shape[0]
        out = dict(token_num=tok, nbest=[[] for _ in range(B)])
        if return_intermediate:
            out.update(enc=r["enc"], olens=r["olens"], embeds=r["embeds"], alphas=r["alphas"], peaks=r["peaks"])
        if max(tok) < 1:
            return out
        am = torch.log_softmax(r["logits"], dim=-1)
        ctc_logp = None
        if model.ctc is not None and model.beam_search.w_ctc != 0:
            ctc_logp = torch.log_softmax(torch.nn.functional.linear(r["enc"], sd["ctc.ctc_lo.weight"], sd["ctc.ctc_lo.bias"]), dim=-1).numpy()
        for i in range(B):
            if tok[i] < 1:
                continue
            lp = ctc_logp[i, : int(r["olens"][i])] if ctc_logp is not None else None
            out["nbest"][i] = model.beam_search(am[i, : tok[i]], lp, maxlenratio=maxlenratio, minlenratio=minlenratio)[: model.nbest]
        return out

    monkeypatch.setattr(model, "recognize_features", recognize_features)
    monkeypatch.setattr(model, "recognize_features_beam", recognize_features_beam)
    return model


def _state(cfg, seed, with_ctc, cif_bias):
    sd = synth.paraformer_state_dict(cfg, seed=seed, cif_bias=cif_bias)
    if with_ctc:
        g = torch.Generator().manual_seed(seed + 9)
        sd["ctc.ctc_lo.weight"] = torch.randn(len(VOCAB), 512, generator=g) * 0.05
        sd["ctc.ctc_lo.bias"] = torch.randn(len(VOCAB), generator=g) * 0.1
    return sd


@pytest.mark.parametrize("with_ctc", [False, True])
def test_paraformer_inference_equals_the_reference(with_ctc, monkeypatch):
    ref_import.install()
    from funasr.tokenizer.char_tokenizer import CharTokenizer as RefTok
    from funasr_amd.tokenizer import CharTokenizer
    cfg = _config(with_ctc)
    g = torch.Generator().manual_seed(31 + int(with_ctc))
    rtok, tok = RefTok(token_list=VOCAB, unk_symbol="<unk>"), CharTokenizer(token_list=VOCAB, unk_symbol="<unk>")
    compared = empty = stamped = 0
    for trial in range(36):
        seed = 70 + trial
        silent = trial % 9 == 8                                        # a batch that predicts no token at all
        sd = _state(cfg, seed, with_ctc, cif_bias=-9.0 if silent else float(torch.rand(1, generator=g)) * 1.5 - 0.5)
        ref = _reference_model(cfg, sd, with_ctc)
        ours = _our_model(cfg, sd, with_ctc, monkeypatch)
        B = int(torch.randint(1, 4, (1,), generator=g))
        T = int(torch.randint(8, 40, (1,), generator=g))
        lens = torch.randint(4, T + 1, (B,), generator=g, dtype=torch.int32)
        lens[int(torch.randint(0, B, (1,), generator=g))] = T
        feats = torch.randn(B, T, 560, generator=g) * 0.7
        for b in range(B):
            feats[b, lens[b]:] = 0
        kw = dict(device="cpu")
        if trial % 3 == 1:
            kw["pred_timestamp"] = True
        if with_ctc and trial % 2 == 0:
            kw.update(decoding_ctc_weight=[0.3, 0.5, 1.0][trial % 3], beam_size=int(torch.randint(1, 4, (1,), generator=g)),
                      token_list=VOCAB, nbest=int(torch.randint(1, 3, (1,), generator=g)), penalty=[0.0, 0.5][trial % 4 == 0])
        use_tok = trial % 5 != 4
        keys = [f"u{b}" for b in range(B)] if trial % 4 else [[f"u{b}" for b in range(B)]]     # AutoModel hands lists of lists over for list inputs
        if trial % 2:                                                   # fbank tensors (data_type="fbank") ...
            r_in = dict(data_in=feats.clone(), data_lengths=lens.clone().long()[:, None], data_type="fbank")
            o_in = dict(data_in=feats.clone(), data_lengths=lens.clone(), data_type="fbank")
            r_fe = o_fe = None
        else:                                                           # ... or waveforms through a frontend
            waves = [torch.zeros(int(l) * 960) for l in lens]
            r_in, o_in = dict(data_in=[w.clone() for w in waves]), dict(data_in=[w.clone() for w in waves])
            r_fe, o_fe = _Frontend(feats.clone(), lens.clone().long()), _Frontend(feats.clone(), lens.clone())
        with torch.no_grad():
            want = ref.inference(key=copy.deepcopy(keys), tokenizer=rtok if use_tok else None, frontend=r_fe, **r_in, **copy.deepcopy(kw))
        got = ours.inference(key=copy.deepcopy(keys), tokenizer=tok if use_tok else None, frontend=o_fe, **o_in, **copy.deepcopy(kw))
        if isinstance(want, list):                                      # nothing decoded: the reference returns a bare [] (:617)
            assert want == [] and isinstance(got, list) and got == [], (trial, got)
            empty += 1
            continue
        w_res, g_res = want[0], got[0]
        assert len(w_res) == len(g_res), (trial, kw, w_res, g_res)
        for a, b in zip(g_res, w_res):
            assert set(a) == set(b), (trial, kw, a, b)
            for k in b:
                if k == "score":
                    assert abs(float(a[k]) - float(b[k])) < 1e-3, (trial, a, b)
                else:
                    assert a[k] == b[k], (trial, kw, k, a, b)
            compared += 1
            stamped += int("timestamp" in b)
    assert compared > 30 and empty >= 2 and stamped > 5
