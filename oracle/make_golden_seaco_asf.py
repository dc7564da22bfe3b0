"""Golden vectors for SeACo's attention-score filter (more hotwords than `nfilter` = 50), made by the REFERENCE class (build
container only; TEST INFRASTRUCTURE). Same construction as make_golden_seaco.py but with a six-block bias decoder (the
filter reads block 5's attention, funasr/models/paraformer/decoder.py:485-513) and 58 hotwords + the no-bias entry:
`SeacoParaformer.inference(hotword=...)` then goes through `_seaco_decode_with_ASF`'s filter branch
(funasr/models/seaco_paraformer/model.py:323-349). Writes tests/golden/seaco_asf.npz.

    python oracle/make_golden_seaco_asf.py
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
from oracle import make_golden_seaco as MS  # noqa: E402
from oracle import ref_import  # noqa: E402
from oracle import seaco_oracle as SO  # noqa: E402

NO_BIAS = MS.NO_BIAS


def hotword_string():
    chars = MB.VOCAB[3:-10]
    words = [chars[i] + chars[(i * 7 + 3) % len(chars)] for i in range(40)] + [chars[i] for i in range(18)]
    return " ".join(words)


def main():
    ref_import.install()
    import funasr.models.sanm.encoder  # noqa: F401
    import funasr.models.paraformer.decoder  # noqa: F401
    import funasr.models.bicif_paraformer.cif_predictor  # noqa: F401
    from funasr.models.seaco_paraformer.model import SeacoParaformer
    from funasr.tokenizer.char_tokenizer import CharTokenizer
    cfg = MS.model_config()
    cfg["seaco_decoder"] = dict(cfg["seaco_decoder"], num_blocks=6, att_layer_num=6)
    ec, dc, sc = cfg["encoder"], cfg["decoder"], cfg["seaco_decoder"]
    seed = 91
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
        seaco_decoder_conf=dict(attention_heads=4, linear_units=sc["linear_units"], num_blocks=6, kernel_size=21,
                                sanm_shfit=0, use_output_layer=False, wo_input_layer=True),
        predictor="CifPredictorV3", predictor_conf=dict(MB.V3), input_size=560, vocab_size=len(MB.VOCAB), ctc_weight=0.0,
        inner_dim=512, bias_encoder_type="lstm", NO_BIAS=NO_BIAS,
    ).eval()
    missing, unexpected = model.load_state_dict(sd, strict=False)
    assert not unexpected and all(k.startswith("criterion") for k in missing), (missing, unexpected)
    tok = CharTokenizer(token_list=MB.VOCAB, unk_symbol="<unk>")
    g = torch.Generator().manual_seed(23)
    B, T = 3, 52
    lens = torch.tensor([52, 44, 21], dtype=torch.int32)
    feats = torch.randn(B, T, 560, generator=g) * 0.7
    for b in range(B):
        feats[b, lens[b]:] = 0
    fe = MB._Frontend(feats, lens)
    tmp = tempfile.mkdtemp()
    with open(os.path.join(tmp, "seg_dict"), "w", encoding="utf-8") as f:
        for ch in MB.VOCAB[3:-10]:
            f.write(f"{ch} {ch}\n")
    fe.cmvn_file = os.path.join(tmp, "am.mvn")
    keys = [f"utt{b}" for b in range(B)]
    hot = hotword_string()
    kept = {}
    orig = model.seaco_decoder.forward_asf6

    def spy(*a, **k):                                            # record the filter's scores for the oracle-side check
        out = orig(*a, **k)
        kept["scores"] = out[0].sum(0).sum(0).clone()
        return out

    model.seaco_decoder.forward_asf6 = spy
    with torch.no_grad():
        res, _ = model.inference([torch.zeros(1600)] * B, key=keys, tokenizer=tok, frontend=fe, device="cpu", hotword=hot)
    hw_list = model.generate_hotwords_list(hot, tokenizer=tok, frontend=fe)
    assert len(hw_list) == 59 and "scores" in kept
    order = torch.topk(kept["scores"], 50)[1].tolist()
    print("hotwords", len(hw_list), "| kept by the filter (first 10)", order[:10], "| texts", [r["text"][:30] for r in res])
    path = os.path.join(os.path.dirname(HERE), "tests", "golden", "seaco_asf.npz")
    np.savez_compressed(path, cfg=json.dumps(cfg), seed=seed, no_bias=NO_BIAS, vocab=json.dumps(MB.VOCAB, ensure_ascii=False),
                        hotwords=hot, hw_list=json.dumps(hw_list), feats=feats.numpy(), lens=lens.numpy(),
                        asf_scores=kept["scores"].numpy(), hot=json.dumps(res, ensure_ascii=False))
    print("wrote", path)


if __name__ == "__main__":
    torch.set_num_threads(8)
    main()
