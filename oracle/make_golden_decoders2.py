"""Golden for the decoder's `decoders2` blocks (att_layer_num < num_blocks; funasr/models/paraformer/decoder.py:363-380,436-437):
the reference's OWN ParaformerSANMDecoder, imported from /root/reference, on seeded inputs; weights are rebuilt from the stored
seed by funasr_amd.synth.decoder_state_dict. TEST INFRASTRUCTURE ONLY (build container only: needs /root/reference).

    python oracle/make_golden_decoders2.py        # writes tests/golden/decoders2.npz
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
from oracle import ref_import  # noqa: E402


def main():
    R = ref_import.modules()
    torch.manual_seed(0)
    torch.set_num_threads(4)
    out = {}
    # (a) the offline geometry: kernel 11, sanm_shfit 0, 2 attention blocks + 2 blocks without cross-attention
    # (b) attention blocks shifted (sanm_shfit 5, the streaming model's causal FSMN) while decoders2 stays centred (decoder.py:371)
    for tag, shift, n_att, n_all in (("a", 0, 2, 4), ("b", 5, 1, 2)):
        dc = dict(vocab_size=97, encoder_output_size=512, attention_heads=4, linear_units=2048, num_blocks=n_all,
                  att_layer_num=n_att, kernel_size=11, sanm_shfit=shift)
        dec = R["ParaformerSANMDecoder"](**dc).eval()
        sd = synth.decoder_state_dict(dc, seed=140 + shift, with_embed=True)
        dec.load_state_dict(sd, strict=True)
        g = torch.Generator().manual_seed(31 + shift)
        memory = torch.randn(3, 37, 512, generator=g)
        mlens = torch.tensor([37, 20, 29], dtype=torch.int32)
        embeds = torch.randn(3, 14, 512, generator=g)
        tlens = torch.tensor([14, 3, 9], dtype=torch.int64)
        with torch.no_grad():
            logits, hidden, _ = dec(memory, mlens, embeds, tlens, return_hidden=True, return_both=True)   # decoder.py:443-449
        out.update({f"{tag}_memory": memory.numpy(), f"{tag}_mem_lens": mlens.numpy(), f"{tag}_embeds": embeds.numpy(),
                    f"{tag}_tok_lens": tlens.numpy(), f"{tag}_logits": logits.numpy(), f"{tag}_hidden": hidden.numpy(),
                    f"{tag}_seed": np.int64(140 + shift), f"{tag}_cfg": json.dumps(dc)})
    path = os.path.join(ROOT, "tests", "golden", "decoders2.npz")
    np.savez_compressed(path, **out)
    print(f"wrote {path} ({os.path.getsize(path) / 1024:.1f} KB)")


if __name__ == "__main__":
    main()
