"""Golden vectors for ContextualParaformer (CLAS hotword biasing), made by the REFERENCE classes (build container only; TEST
INFRASTRUCTURE). Builds `ContextualParaformer` + `ContextualParaformerDecoder` (funasr/models/contextual_paraformer/) from a
tiny config with seeded weights and calls `inference()` on given LFR features without hotwords, with hotwords and with
clas_scale 0.6. Writes tests/golden/contextual.npz (features, settings, texts, the reference's state_dict layout).

    python oracle/make_golden_contextual.py
"""
import json
import os
import sys
import tempfile

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
from funasr_amd import synth  # noqa: E402
from oracle import make_golden_bicif as MB  # noqa: E402
from oracle import ref_import  # noqa: E402

HOTWORDS = "我们 大地 hello 国 时"


def model_config():
    cfg = synth.tiny(synth.PARAFORMER_LARGE, enc_blocks=2, dec_blocks=3, vocab=len(MB.VOCAB))
    cfg["decoder"].update(kernel_size=11, sanm_shfit=0)
    return cfg


def contextual_state_dict(cfg: dict, seed: int):
    """Paraformer weights with the last attention block renamed to decoder.last_decoder.*, plus the hotword branch"""
    sd = synth.paraformer_state_dict(cfg, seed=seed, cif_bias=-0.3)
    L = cfg["decoder"]["att_layer_num"]
    out = {}
    for k, v in sd.items():
        out[k.replace(f"decoder.decoders.{L - 1}.", "decoder.last_decoder.")] = v
    g = torch.Generator().manual_seed(seed + 3)
    D, V = 512, cfg["decoder"]["vocab_size"]
    out["decoder.embed.0.weight"] = torch.randn(V, D, generator=g) * 0.5
    out["decoder.bias_decoder.norm3.weight"] = 1.0 + 0.1 * torch.randn(D, generator=g)
    out["decoder.bias_decoder.norm3.bias"] = 0.1 * torch.randn(D, generator=g)
    for name, o in (("linear_q", D), ("linear_k_v", 2 * D), ("linear_out", D)):
        out[f"decoder.bias_decoder.src_attn.{name}.weight"] = torch.randn(o, D, generator=g) / D ** 0.5
        out[f"decoder.bias_decoder.src_attn.{name}.bias"] = 0.1 * torch.randn(o, generator=g)
    out["decoder.bias_output.weight"] = torch.randn(D, 2 * D, 1, generator=g) * 1.5 / (2 * D) ** 0.5
    out["bias_embed.weight"] = torch.randn(V, D, generator=g) * 0.5
    out["bias_encoder.weight_ih_l0"] = torch.randn(4 * D, D, generator=g) / D ** 0.5
    out["bias_encoder.weight_hh_l0"] = torch.randn(4 * D, D, generator=g) * 0.7 / D ** 0.5
    out["bias_encoder.bias_ih_l0"] = torch.randn(4 * D, generator=g) * 0.1
    out["bias_encoder.bias_hh_l0"] = torch.randn(4 * D, generator=g) * 0.1
    return out


def main():
    ref_import.install()
    import funasr.models.sanm.encoder  # noqa: F401
    import funasr.models.paraformer.cif_predictor  # noqa: F401
    import funasr.models.contextual_paraformer.decoder  # noqa: F401
    from funasr.models.contextual_paraformer.model import ContextualParaformer
    from funasr.tokenizer.char_tokenizer import CharTokenizer
    cfg = model_config()
    ec, dc = cfg["encoder"], cfg["decoder"]
    seed = 57
    sd = contextual_state_dict(cfg, seed)
    model = ContextualParaformer(
        encoder="SANMEncoder",
        encoder_conf=dict(output_size=512, attention_heads=4, linear_units=ec["linear_units"], num_blocks=ec["num_blocks"],
                          input_layer="pe", pos_enc_class="SinusoidalPositionEncoder", normalize_before=True,
                          kernel_size=11, sanm_shfit=0, selfattention_layer_type="sanm"),
        decoder="ContextualParaformerDecoder",
        decoder_conf=dict(attention_heads=4, linear_units=dc["linear_units"], num_blocks=dc["num_blocks"],
                          att_layer_num=dc["att_layer_num"], kernel_size=11, sanm_shfit=0),
        predictor="CifPredictorV2", predictor_conf=dict(cfg["predictor"]), input_size=560, vocab_size=len(MB.VOCAB),
        ctc_weight=0.0, inner_dim=512, bias_encoder_type="lstm",
    ).eval()
    missing, unexpected = model.load_state_dict(sd, strict=False)
    assert not unexpected and all(k.startswith("criterion") for k in missing), (missing, unexpected)
    tok = CharTokenizer(token_list=MB.VOCAB, unk_symbol="<unk>")
    g = torch.Generator().manual_seed(29)
    B, T = 3, 50
    lens = torch.tensor([50, 37, 22], dtype=torch.int32)
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
        for name, kw in (("plain", dict()), ("hot", dict(hotword=HOTWORDS)), ("hot_scaled", dict(hotword=HOTWORDS, clas_scale=0.6))):
            res, _ = model.inference([torch.zeros(1600)] * B, key=keys, tokenizer=tok, frontend=fe, device="cpu", **kw)
            out[name] = res
            print(name, [r["text"][:40] for r in res])
    hw_list = model.generate_hotwords_list(HOTWORDS, tokenizer=tok, frontend=fe)
    ref_keys = {k: list(v.shape) for k, v in model.state_dict().items()}
    path = os.path.join(os.path.dirname(HERE), "tests", "golden", "contextual.npz")
    np.savez_compressed(path, cfg=json.dumps(cfg), seed=seed, vocab=json.dumps(MB.VOCAB, ensure_ascii=False), hotwords=HOTWORDS,
                        hw_list=json.dumps(hw_list), ref_state_dict=json.dumps(ref_keys), feats=feats.numpy(), lens=lens.numpy(),
                        **{k: json.dumps(v, ensure_ascii=False) for k, v in out.items()})
    print("wrote", path, "| differing texts hot vs plain:", sum(a["text"] != b["text"] for a, b in zip(out["plain"], out["hot"])))


if __name__ == "__main__":
    torch.set_num_threads(8)
    main()
