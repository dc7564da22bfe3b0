"""Tensor-level golden for the contextual decoder: the reference's OWN ContextualParaformerDecoder
(funasr/models/contextual_paraformer/decoder.py:133-352), imported from /root/reference, on seeded inputs -- logits with hotword rows,
with clas_scale 0.6, and with num_blocks > att_layer_num (decoders2 behind the hotword fusion). Weights are rebuilt from the seed by
oracle.make_golden_contextual.contextual_state_dict. TEST INFRASTRUCTURE ONLY (build container only).

    python oracle/make_golden_contextual_decoder.py        # writes tests/golden/contextual_decoder.npz
"""
import json
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, ROOT)

from funasr_amd import synth  # noqa: E402
from oracle import make_golden_contextual as MC  # noqa: E402
from oracle import ref_import  # noqa: E402


def decoder_weights(dc: dict, seed: int):
    """`decoder.*` of contextual_state_dict for a decoder config (att_layer_num / num_blocks free)"""
    cfg = synth.tiny(synth.PARAFORMER_LARGE, enc_blocks=1, dec_blocks=dc["att_layer_num"], vocab=dc["vocab_size"])
    cfg["decoder"].update(num_blocks=dc["num_blocks"], att_layer_num=dc["att_layer_num"], kernel_size=dc["kernel_size"], sanm_shfit=dc["sanm_shfit"])
    sd = MC.contextual_state_dict(cfg, seed)
    return {k[len("decoder."):]: v for k, v in sd.items() if k.startswith("decoder.")}


def main():
    ref_import.install()
    from funasr.models.contextual_paraformer.decoder import ContextualParaformerDecoder
    torch.manual_seed(0)
    torch.set_num_threads(4)
    out = {}
    for tag, n_att, n_all, scale, n_hot in (("a", 3, 3, 1.0, 6), ("b", 2, 2, 0.6, 1), ("c", 2, 4, 1.0, 9)):
        dc = dict(vocab_size=83, encoder_output_size=512, attention_heads=4, linear_units=2048, num_blocks=n_all, att_layer_num=n_att,
                  kernel_size=11, sanm_shfit=0)
        dec = ContextualParaformerDecoder(**dc).eval()
        sd = decoder_weights(dc, 170 + n_all)
        missing, unexpected = dec.load_state_dict(sd, strict=False)
        assert not unexpected and not missing, (missing, unexpected)
        g = torch.Generator().manual_seed(50 + n_hot)
        memory = torch.randn(3, 33, 512, generator=g)
        mlens = torch.tensor([33, 18, 27], dtype=torch.int32)
        embeds = torch.randn(3, 12, 512, generator=g)
        tlens = torch.tensor([12, 5, 8], dtype=torch.int64)
        hot = torch.randn(1, n_hot, 512, generator=g)
        with torch.no_grad():
            logits, _ = dec(memory, mlens, embeds, tlens, contextual_info=hot.repeat(3, 1, 1), clas_scale=scale)
        out.update({f"{tag}_memory": memory.numpy(), f"{tag}_mem_lens": mlens.numpy(), f"{tag}_embeds": embeds.numpy(),
                    f"{tag}_tok_lens": tlens.numpy(), f"{tag}_hot": hot.numpy(), f"{tag}_logits": logits.numpy(),
                    f"{tag}_seed": np.int64(170 + n_all), f"{tag}_scale": np.float32(scale), f"{tag}_cfg": json.dumps(dc)})
    path = os.path.join(ROOT, "tests", "golden", "contextual_decoder.npz")
    np.savez_compressed(path, **out)
    print(f"wrote {path} ({os.path.getsize(path) / 1024:.1f} KB)")


if __name__ == "__main__":
    main()
