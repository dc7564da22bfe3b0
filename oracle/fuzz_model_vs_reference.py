"""Fuzz the CPU oracle's network restatements against the REFERENCE modules (build container only; TEST INFRASTRUCTURE):
random batch sizes, lengths and seeds through SANMEncoder, CifPredictorV2 (+ cif_v1) and ParaformerSANMDecoder loaded
with the same seeded state_dicts. Prints the largest differences; the committed goldens (make_golden.py) pin a few fixed
cases, this sweeps the shape space the GPU parity tests rely on the oracle for."""
import os
import sys

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
from funasr_amd import synth  # noqa: E402
from oracle import paraformer_oracle as O  # noqa: E402
from oracle import ref_import  # noqa: E402


def main(n_cases=24):
    R = ref_import.modules()
    torch.set_num_threads(8)
    worst = dict(encoder=0.0, alphas=0.0, embeds=0.0, logits=0.0)
    idx_bad = 0
    g = torch.Generator().manual_seed(99)
    for ci in range(n_cases):
        cfg = synth.tiny(synth.PARAFORMER_LARGE, enc_blocks=int(torch.randint(1, 4, (1,), generator=g)),
                         dec_blocks=int(torch.randint(1, 3, (1,), generator=g)), vocab=int(torch.randint(20, 200, (1,), generator=g)))
        ec, pc, dc = cfg["encoder"], cfg["predictor"], cfg["decoder"]
        seed = 1000 + ci
        sd = synth.paraformer_state_dict(cfg, seed=seed, cif_bias=float(torch.rand(1, generator=g)) - 0.3)
        enc = R["SANMEncoder"](input_size=ec["input_size"], output_size=ec["output_size"], attention_heads=ec["attention_heads"],
                               linear_units=ec["linear_units"], num_blocks=ec["num_blocks"], dropout_rate=0.1, input_layer="pe",
                               pos_enc_class=R["SinusoidalPositionEncoder"], normalize_before=True,
                               kernel_size=ec["kernel_size"], sanm_shfit=ec["sanm_shfit"], selfattention_layer_type="sanm").eval()
        enc.load_state_dict({k[len("encoder."):]: v for k, v in sd.items() if k.startswith("encoder.")}, strict=True)
        pred = R["CifPredictorV2"](idim=pc["idim"], l_order=pc["l_order"], r_order=pc["r_order"], threshold=pc["threshold"],
                                   tail_threshold=pc["tail_threshold"]).eval()
        pred.load_state_dict({k[len("predictor."):]: v for k, v in sd.items() if k.startswith("predictor.")}, strict=True)
        dec = R["ParaformerSANMDecoder"](vocab_size=dc["vocab_size"], encoder_output_size=dc["encoder_output_size"],
                                         attention_heads=dc["attention_heads"], linear_units=dc["linear_units"],
                                         num_blocks=dc["num_blocks"], att_layer_num=dc["att_layer_num"],
                                         kernel_size=dc["kernel_size"], sanm_shfit=dc["sanm_shfit"]).eval()
        dsd = {k[len("decoder."):]: v for k, v in sd.items() if k.startswith("decoder.")}
        dsd["embed.0.weight"] = torch.zeros(dc["vocab_size"], 512)
        dec.load_state_dict(dsd, strict=True)
        B = int(torch.randint(1, 5, (1,), generator=g))
        T = int(torch.randint(4, 70, (1,), generator=g))
        lens = torch.randint(2, T + 1, (B,), generator=g, dtype=torch.int32)
        lens[int(torch.randint(0, B, (1,), generator=g))] = T
        xs = torch.randn(B, T, 560, generator=g) * 0.7
        for b in range(B):
            xs[b, lens[b]:] = 0
        with torch.no_grad():
            r_enc, r_olens, _ = enc(xs.clone(), lens)
            mask = (torch.arange(T)[None, :] < lens[:, None]).float()[:, None, :]
            try:
                r_emb, r_tok, r_alpha, r_peak = pred(r_enc, None, mask)
            except IndexError:                                  # zero-token last utterance (cif_predictor.py:887)
                continue
        o_enc, o_olens = O.sanm_encoder(xs, lens, sd, ec, "encoder.")
        o_emb, o_tok, o_alpha, o_peak = O.cif_predictor(r_enc, lens, sd, pc, "predictor.")
        worst["encoder"] = max(worst["encoder"], (o_enc - r_enc).abs().max().item())
        worst["alphas"] = max(worst["alphas"], (o_alpha - r_alpha).abs().max().item())
        n = min(o_emb.shape[1], r_emb.shape[1])
        worst["embeds"] = max(worst["embeds"], (o_emb[:, :n] - r_emb[:, :n]).abs().max().item())
        if o_emb.shape != r_emb.shape or not torch.equal(o_tok, r_tok) or \
                not torch.equal(torch.floor(o_peak) >= 1, torch.floor(r_peak) >= 1):
            idx_bad += 1
            print("case", ci, "integer results differ", o_emb.shape, r_emb.shape, o_tok.tolist(), r_tok.tolist())
        tok = r_tok.round().long()
        if int(tok.max()) < 1:
            continue
        with torch.no_grad():
            r_logits, _ = dec(r_enc, lens, r_emb, tok)
        o_logits = O.paraformer_decoder(r_enc, lens, r_emb, tok, sd, dc, "decoder.")
        if r_logits.shape[-1] == o_logits.shape[-1]:
            ref_l = r_logits if r_logits.max() > 0 else None      # the reference returns log-softmax or raw logits by call path
            cmp_r = torch.log_softmax(r_logits, -1)
            cmp_o = torch.log_softmax(o_logits, -1)
            for b in range(B):
                k = int(tok[b])
                if k:
                    worst["logits"] = max(worst["logits"], (cmp_o[b, :k] - cmp_r[b, :k]).abs().max().item())
    print(f"cases {n_cases}: max |oracle - reference|: {worst}; cases with different integer results: {idx_bad}")


if __name__ == "__main__":
    main(int(sys.argv[1]) if len(sys.argv) > 1 else 24)
