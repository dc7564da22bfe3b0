"""Golden for ContextualParaformer's beam-search route (CTC-rescored n-best over the hotword-biased decoder scores;
funasr/models/contextual_paraformer/model.py:408-415,483-494 with Paraformer.init_beam_search, paraformer/model.py:482-532), made by
the REFERENCE classes' own `inference` (build container only; TEST INFRASTRUCTURE). Same tiny model as make_golden_contextual.py plus a
CTC head (ctc_weight 0.3). Writes tests/golden/contextual_beam.npz.

    python oracle/make_golden_contextual_beam.py
"""
import json
import os
import sys
import tempfile

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
from oracle import make_golden_bicif as MB  # noqa: E402
from oracle import make_golden_contextual as MC  # noqa: E402
from oracle import ref_import  # noqa: E402

BEAM_KW = dict(decoding_ctc_weight=0.5, beam_size=3, nbest=2, penalty=0.1, maxlenratio=0.0, minlenratio=0.0)


def ctc_state_dict(cfg: dict, seed: int):
    g = torch.Generator().manual_seed(seed + 11)
    V = cfg["decoder"]["vocab_size"]
    w = torch.randn(V, 512, generator=g) * 0.2
    b = torch.randn(V, generator=g) * 0.3
    b[0] += 1.5                                                 # blank is frequent, as in a trained CTC head
    return {"ctc.ctc_lo.weight": w, "ctc.ctc_lo.bias": b}


def main():
    ref_import.install()
    import funasr.models.sanm.encoder  # noqa: F401
    import funasr.models.paraformer.cif_predictor  # noqa: F401
    import funasr.models.contextual_paraformer.decoder  # noqa: F401
    from funasr.models.contextual_paraformer.model import ContextualParaformer
    from funasr.tokenizer.char_tokenizer import CharTokenizer
    cfg = MC.model_config()
    ec, dc = cfg["encoder"], cfg["decoder"]
    seed = 61
    sd = MC.contextual_state_dict(cfg, seed)
    sd.update(ctc_state_dict(cfg, seed))
    model = ContextualParaformer(
        encoder="SANMEncoder",
        encoder_conf=dict(output_size=512, attention_heads=4, linear_units=ec["linear_units"], num_blocks=ec["num_blocks"],
                          input_layer="pe", pos_enc_class="SinusoidalPositionEncoder", normalize_before=True,
                          kernel_size=11, sanm_shfit=0, selfattention_layer_type="sanm"),
        decoder="ContextualParaformerDecoder",
        decoder_conf=dict(attention_heads=4, linear_units=dc["linear_units"], num_blocks=dc["num_blocks"],
                          att_layer_num=dc["att_layer_num"], kernel_size=11, sanm_shfit=0),
        predictor="CifPredictorV2", predictor_conf=dict(cfg["predictor"]), input_size=560, vocab_size=len(MB.VOCAB),
        ctc_weight=0.3, inner_dim=512, bias_encoder_type="lstm",
    ).eval()
    missing, unexpected = model.load_state_dict(sd, strict=False)
    assert not unexpected and all(k.startswith("criterion") for k in missing), (missing, unexpected)
    tok = CharTokenizer(token_list=MB.VOCAB, unk_symbol="<unk>")
    g = torch.Generator().manual_seed(31)
    B, T = 3, 46
    lens = torch.tensor([46, 30, 39], dtype=torch.int32)
    feats = torch.randn(B, T, 560, generator=g) * 0.7
    for b in range(B):
        feats[b, lens[b]:] = 0
    fe = MB._Frontend(feats, lens)
    tmp = tempfile.mkdtemp()
    with open(os.path.join(tmp, "seg_dict"), "w", encoding="utf-8") as f:
        for ch in MB.VOCAB[3:-10]:
            f.write(f"{ch} {ch}\n")
        f.write("hello hel@@ lo\nworld wor@@ ld\nthe the\n")
    fe.cmvn_file = os.path.join(tmp, "am.mvn")
    keys = [f"utt{b}" for b in range(B)]
    out = {}
    with torch.no_grad():
        greedy, _ = model.inference([torch.zeros(1600)] * B, key=keys, tokenizer=tok, frontend=fe, device="cpu", hotword=MC.HOTWORDS)
        for name, kw in (("beam_hot", dict(hotword=MC.HOTWORDS)), ("beam_plain", dict())):
            res, _ = model.inference([torch.zeros(1600)] * B, key=keys, tokenizer=tok, frontend=fe, device="cpu", token_list=MB.VOCAB,
                                     **BEAM_KW, **kw)
            out[name] = res
            print(name, [(r["key"], r["text"][:30]) for r in res])
    print("greedy ", [(r["key"], r["text"][:30]) for r in greedy])
    path = os.path.join(os.path.dirname(HERE), "tests", "golden", "contextual_beam.npz")
    np.savez_compressed(path, cfg=json.dumps(cfg), seed=seed, vocab=json.dumps(MB.VOCAB, ensure_ascii=False), hotwords=MC.HOTWORDS,
                        beam_kw=json.dumps(BEAM_KW), feats=feats.numpy(), lens=lens.numpy(), greedy_hot=json.dumps(greedy, ensure_ascii=False),
                        **{k: json.dumps(v, ensure_ascii=False) for k, v in out.items()})
    print("wrote", path)


if __name__ == "__main__":
    torch.set_num_threads(8)
    main()
