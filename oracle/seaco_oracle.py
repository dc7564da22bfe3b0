"""CPU restatement of SeACo-Paraformer's hotword path.  TEST INFRASTRUCTURE: only tests/, __graft_entry__.smoke() and
bench.py's cpu_baseline leg may import this; the product path never does.

Follows `SeacoParaformer._seaco_decode_with_ASF` / `_hotword_representation` / `generate_hotwords_list`
(funasr/models/seaco_paraformer/model.py:270-385, :388-424, :583-690) for hotword lists that stay under the attention
filter's size (nfilter = 50: the ASF branch :323-349 only runs for more hotwords than that). Pinned to the reference class
by oracle/make_golden_seaco.py -> tests/golden/seaco.npz.
"""
from __future__ import annotations

from typing import Dict, List

import torch
import torch.nn.functional as F

from oracle import bicif_oracle as BO
from oracle import paraformer_oracle as O

Tensor = torch.Tensor
SD = Dict[str, Tensor]


def hotword_representation(hw_list: List[List[int]], sd: SD) -> Tensor:
    """embed (the decoder's token table) -> 2-layer LSTM -> the output at each hotword's last token (:388-424)"""
    lens = [len(h) for h in hw_list]
    pad = torch.zeros(len(hw_list), max(lens), dtype=torch.long)
    for i, h in enumerate(hw_list):
        pad[i, : len(h)] = torch.tensor(h)
    emb = F.embedding(pad, sd["decoder.embed.0.weight"])
    out = BO.lstm(emb, sd, "bias_encoder.", layers=2, bidirectional=False)     # uni-directional: padding cannot leak back
    return out[torch.arange(len(hw_list)), torch.tensor(lens) - 1]


def seaco_ids(enc: Tensor, olens: Tensor, embeds: Tensor, tok: Tensor, hw_list, sd: SD, cfg: dict, no_bias: int):
    """-> (ids [B, N] after the merge, decoder ids, bias-decoder ids): :293-385 with seaco_weight 1: a position takes the
    bias decoder's token unless that token is NO_BIAS"""
    logits, hidden = O.paraformer_decoder(enc, olens, embeds, tok, sd, cfg["decoder"], "decoder.", return_hidden=True)
    dec_ids = torch.log_softmax(logits, -1).argmax(-1)
    if hw_list is None:
        return dec_ids, dec_ids, None
    sel = hotword_representation(hw_list, sd)
    B = enc.shape[0]
    ctx = sel[None].repeat(B, 1, 1)
    clen = torch.full((B,), sel.shape[0], dtype=torch.int32)
    cif_att = O.paraformer_decoder(ctx, clen, embeds, tok, sd, cfg["seaco_decoder"], "seaco_decoder.")
    dec_att = O.paraformer_decoder(ctx, clen, hidden, tok, sd, cfg["seaco_decoder"], "seaco_decoder.")
    dha = F.linear(cif_att + dec_att, sd["hotword_output_layer.weight"], sd["hotword_output_layer.bias"])
    dha_ids = torch.log_softmax(dha, -1).argmax(-1)
    return torch.where(dha_ids == no_bias, dec_ids, dha_ids), dec_ids, dha_ids


def seaco_greedy(feats: Tensor, lens: Tensor, hw_list, sd: SD, cfg: dict, no_bias: int, sos: int = 1, eos: int = 2, blank: int = 0):
    enc, olens = O.sanm_encoder(feats, lens, sd, cfg["encoder"], "encoder.")
    embeds, token_num, alphas, peaks = BO.predictor_v3(enc, olens, sd, cfg["predictor"], "predictor.")
    tok = token_num.round().long()
    ids, dec_ids, dha_ids = seaco_ids(enc, olens, embeds, tok, hw_list, sd, cfg, no_bias)
    usa, usp = BO.upsample_timestamp(enc, olens, tok, sd, cfg["predictor"], "predictor.")
    B = feats.shape[0]
    raw = [ids[b, : int(tok[b])].tolist() for b in range(B)]
    return dict(enc=enc, olens=olens, embeds=embeds, token_num=tok, raw_ids=raw, dec_ids=dec_ids, dha_ids=dha_ids,
                ids=[[t for t in r if t not in (sos, eos, blank)] for r in raw], us_alphas=usa, us_peaks=usp)


SEACO_DECODER = dict(vocab_size=0, encoder_output_size=512, attention_heads=4, linear_units=1024, num_blocks=4, att_layer_num=6,
                     kernel_size=21, sanm_shfit=0)


def seaco_state_dict(cfg: dict, seed: int, no_bias: int) -> SD:
    """BiCif weights + decoder token table, hotword LSTM (2 layers), bias decoder (kernel 21, no output layer) and the hotword
    output layer, biased so that about half the positions answer NO_BIAS"""
    from funasr_amd import synth
    sd = synth.paraformer_state_dict(cfg, seed=seed, cif_bias=-0.3)
    sd.update(BO.predictor_v3_state_dict(cfg["predictor"], seed=seed + 5, prefix="predictor.", cif_bias=-0.3))
    g = torch.Generator().manual_seed(seed + 9)
    D, V = 512, cfg["decoder"]["vocab_size"]
    sd["decoder.embed.0.weight"] = torch.randn(V, D, generator=g) * 0.5
    for layer in range(2):
        sd[f"bias_encoder.weight_ih_l{layer}"] = torch.randn(4 * D, D, generator=g) / D ** 0.5
        sd[f"bias_encoder.weight_hh_l{layer}"] = torch.randn(4 * D, D, generator=g) * 0.7 / D ** 0.5
        sd[f"bias_encoder.bias_ih_l{layer}"] = torch.randn(4 * D, generator=g) * 0.1
        sd[f"bias_encoder.bias_hh_l{layer}"] = torch.randn(4 * D, generator=g) * 0.1
    dsd = synth.decoder_state_dict(dict(cfg["seaco_decoder"], vocab_size=4), seed=seed + 11, prefix="seaco_decoder.")
    sd.update({k: v for k, v in dsd.items() if "output_layer" not in k})
    sd["hotword_output_layer.weight"] = torch.randn(V, D, generator=g) * 0.1
    bias = torch.randn(V, generator=g) * 0.1
    bias[no_bias] = 6.0
    sd["hotword_output_layer.bias"] = bias
    return sd
