#!/usr/bin/env python3
"""GPU fuzz of the contextual (CLAS) decoder: random batch sizes, memory / token lengths, hotword counts, clas_scale, attention-block counts and
decoders2 counts through ContextualParaformerDecoder (pf_decoder_forward_contextual) against oracle/paraformer_oracle.py contextual_decoder
(itself pinned to the reference's own class, tests/test_contextual.py): logits within 1e-3 and arg-max ids (a flip only at a near-tie).
Not part of the test run. usage: fuzz_gpu_contextual_vs_oracle.py [seed] [cases]"""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from funasr_amd.contextual_paraformer import ContextualParaformerDecoder     # noqa: E402
from oracle import paraformer_oracle as O                                    # noqa: E402
from oracle.make_golden_contextual_decoder import decoder_weights            # noqa: E402

dev = torch.device("cuda:0")
seed = int(sys.argv[1]) if len(sys.argv) > 1 else 0
n_cases = int(sys.argv[2]) if len(sys.argv) > 2 else 20
g = torch.Generator().manual_seed(seed)


def ri(lo, hi):
    return int(torch.randint(lo, hi + 1, (1,), generator=g))


bad, worst, flips = 0, 0.0, 0
for ci in range(n_cases):
    n_att = ri(1, 3)
    dc = dict(vocab_size=ri(30, 300), encoder_output_size=512, attention_heads=4, linear_units=2048, num_blocks=n_att + ri(0, 2),
              att_layer_num=n_att, kernel_size=11, sanm_shfit=0)
    sd = decoder_weights(dc, 800 + ci)
    d = ContextualParaformerDecoder(**dc)
    d.load_state_dict(sd, strict=False)
    d = d.to(dev)
    B, T, N, n_hot = ri(1, 6), ri(2, 200), ri(1, 40), ri(1, 60)
    mlens = torch.randint(1, T + 1, (B,), generator=g, dtype=torch.int32); mlens[0] = T
    tlens = torch.randint(1, N + 1, (B,), generator=g, dtype=torch.int64); tlens[ri(0, B - 1)] = N
    memory = torch.randn(B, T, 512, generator=g) * 0.8
    embeds = torch.randn(B, N, 512, generator=g) * 0.8
    hot = torch.randn(1, n_hot, 512, generator=g)
    scale = (1.0, 0.6, 0.0, 1.7)[ri(0, 3)]
    want = O.contextual_decoder(memory, mlens, embeds, tlens, hot, sd, dc, clas_scale=scale)
    got, _ = d(memory.to(dev), mlens, embeds.to(dev), tlens, contextual_info=hot.to(dev), clas_scale=scale)
    ids, _ = d.greedy(memory.to(dev), mlens, embeds.to(dev), tlens, contextual_info=hot.to(dev), clas_scale=scale)
    ok = True
    for b in range(B):
        n = int(tlens[b])
        dd = (got[b, :n].cpu() - want[b, :n]).abs().max().item()
        worst = max(worst, dd)
        ok = ok and dd < 1e-3
        ref_ids = want[b, :n].argmax(-1).tolist()
        for pos, (a_, b_) in enumerate(zip(ids[b, :n].cpu().tolist(), ref_ids)):
            if a_ != b_:
                top2 = torch.topk(want[b, pos], 2).values
                flips += 1
                ok = ok and float(top2[0] - top2[1]) < 1e-4
    if not ok:
        bad += 1
        print(f"case {ci}: B={B} T={T} N={N} hot={n_hot} scale={scale} att={n_att} blocks={dc['num_blocks']} mlens={mlens.tolist()} tlens={tlens.tolist()}: MISMATCH")
print(json.dumps(dict(tool="fuzz_gpu_contextual_vs_oracle", seed=seed, cases=n_cases, bad=bad, worst_logit_abs_diff=worst, near_tie_flips=flips)))
sys.exit(1 if bad else 0)
