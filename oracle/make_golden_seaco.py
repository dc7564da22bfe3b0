"""Golden vectors for SeACo-Paraformer's hotword path, made by the REFERENCE class (build container only; TEST
INFRASTRUCTURE). Builds `SeacoParaformer` (funasr/models/seaco_paraformer/model.py) from a tiny config with seeded weights,
calls its `inference()` with and without hotwords on given LFR features (stand-in frontend) and stores the results in
tests/golden/seaco.npz.

    python oracle/make_golden_seaco.py
"""
import json
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
from oracle import make_golden_bicif as MB  # noqa: E402
from oracle import ref_import  # noqa: E402
from oracle import seaco_oracle as SO  # noqa: E402

NO_BIAS = 5
HOTWORDS = "我们 大地 hello 国"


def model_config():
    cfg = MB.model_config()
    cfg["seaco_decoder"] = dict(SO.SEACO_DECODER)
    return cfg


def main():
    ref_import.install()
    import funasr.models.sanm.encoder  # noqa: F401
    import funasr.models.paraformer.decoder  # noqa: F401
    import funasr.models.bicif_paraformer.cif_predictor  # noqa: F401
    from funasr.models.seaco_paraformer.model import SeacoParaformer
    from funasr.tokenizer.char_tokenizer import CharTokenizer
    cfg = model_config()
    ec, dc, sc = cfg["encoder"], cfg["decoder"], cfg["seaco_decoder"]
    seed = 83
    sd = SO.seaco_state_dict(cfg, seed, NO_BIAS)
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
        inner_dim=512, bias_encoder_type="lstm", NO_BIAS=NO_BIAS,
    ).eval()
    missing, unexpected = model.load_state_dict(sd, strict=False)
    assert not unexpected, unexpected
    assert all(k.startswith("criterion") for k in missing), missing
    tok = CharTokenizer(token_list=MB.VOCAB, unk_symbol="<unk>")
    g = torch.Generator().manual_seed(17)
    B, T = 3, 52
    lens = torch.tensor([52, 40, 19], dtype=torch.int32)
    feats = torch.randn(B, T, 560, generator=g) * 0.7
    for b in range(B):
        feats[b, lens[b]:] = 0
    fe = MB._Frontend(feats, lens)
    # a seg_dict next to the cmvn file makes multi-token hotwords (seaco_paraformer/model.py:616-623)
    import tempfile
    tmp = tempfile.mkdtemp()
    with open(os.path.join(tmp, "seg_dict"), "w", encoding="utf-8") as f:
        for ch in MB.VOCAB[3:-10]:
            f.write(f"{ch} {ch}\n")
        f.write("hello hel@@ lo\nworld wor@@ ld\nthe the\n")
    fe.cmvn_file = os.path.join(tmp, "am.mvn")
    keys = [f"utt{b}" for b in range(B)]
    out = {}
    with torch.no_grad():
        for name, hw in (("plain", None), ("hot", HOTWORDS)):
            res, _ = model.inference([torch.zeros(1600)] * B, key=keys, tokenizer=tok, frontend=fe, device="cpu", hotword=hw)
            out[name] = res
            print(name, [r["text"][:40] for r in res])
    hw_list = model.generate_hotwords_list(HOTWORDS, tokenizer=tok, frontend=fe)
    o = SO.seaco_greedy(feats, lens, hw_list, sd, cfg, NO_BIAS)
    n_nb = int((o["dha_ids"] == NO_BIAS).sum())
    print("hotword ids", hw_list, "| NO_BIAS positions", n_nb, "of", o["dha_ids"].numel(), "| merged differs from decoder at",
          int((o["dec_ids"] != torch.where(o["dha_ids"] == NO_BIAS, o["dec_ids"], o["dha_ids"])).sum()))
    ref_keys = {k: list(v.shape) for k, v in model.state_dict().items()}
    path = os.path.join(os.path.dirname(HERE), "tests", "golden", "seaco.npz")
    np.savez_compressed(path, cfg=json.dumps(cfg), seed=seed, no_bias=NO_BIAS, vocab=json.dumps(MB.VOCAB, ensure_ascii=False),
                        hotwords=HOTWORDS, hw_list=json.dumps(hw_list), ref_state_dict=json.dumps(ref_keys), feats=feats.numpy(), lens=lens.numpy(),
                        plain=json.dumps(out["plain"], ensure_ascii=False), hot=json.dumps(out["hot"], ensure_ascii=False))
    print("wrote", path)


if __name__ == "__main__":
    torch.set_num_threads(8)
    main()
