#!/usr/bin/env python
"""One-off check behind INTEGRATION's note on batch plans: which clips' ids depend on the batch they are decoded in. Full-size Paraformer
(random init, confident output layer), clips A (10 s), B (8 s), C (12 s), D (6 s): B next to A, next to C, next to both -- never the
longest -- must give the same ids; A as the longest of [A, B] against A as a shorter member of [C, A] may differ in its tail (the row
behind the last frame that CifPredictorV2's conv reads exists only for clips that are not the longest of their batch)."""
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from funasr_amd import synth                                      # noqa: E402
from funasr_amd.paraformer import Paraformer                      # noqa: E402
from funasr_amd.wav_frontend import WavFrontend                   # noqa: E402

dev = torch.device("cuda:0")
cfg = synth.PARAFORMER_LARGE
model = Paraformer.from_config(cfg)
model.load_state_dict(synth.paraformer_state_dict(cfg, seed=0, cif_bias=synth.BENCH_CIF_BIAS), strict=False)
model = model.to(dev)
sh, sc = synth.synthetic_cmvn(560)
fe = WavFrontend(cmvn=torch.stack([sh, sc]), lfr_m=7, lfr_n=6, dither=0.0, device=dev)
pool = [synth.speech_like(int(s * 16000), seed=300 + i) for i, s in enumerate((10, 8, 12, 6) * 8)]
cal = torch.nn.utils.rnn.pad_sequence(pool, batch_first=True).to(dev)
synth.make_paraformer_confident(model, *fe(cal, [p.numel() for p in pool]))


def ids(clips):
    wav = torch.nn.utils.rnn.pad_sequence(clips, batch_first=True).to(dev)
    return model.recognize_features(*fe(wav, [c.numel() for c in clips]))["ids"]


rows = []
for k in range(8):
    A, B, C, D = pool[4 * k: 4 * k + 4]
    b = [ids([A, B])[1], ids([C, B])[1], ids([C, A, B, D])[2]]
    a_long, a_short = ids([A, B])[0], ids([C, A])[1]
    n = min(len(a_long), len(a_short))
    rows.append({"B_same_in_three_batches": b[0] == b[1] == b[2], "A_longest_vs_not": a_long == a_short, "A_tokens": [len(a_long), len(a_short)],
                 "A_first_difference_at": next((i for i in range(n) if a_long[i] != a_short[i]), n if len(a_long) != len(a_short) else None)})
print(json.dumps({"clips": rows, "never_longest_always_equal": all(r["B_same_in_three_batches"] for r in rows),
                  "longest_clip_differs": sum(not r["A_longest_vs_not"] for r in rows)}))
