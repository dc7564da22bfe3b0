#!/usr/bin/env python3
"""GPU fuzz of the two encoder-only heads: (1) SenseVoiceSmall -- random batch sizes, frame counts, ragged lengths, languages / text-norm
queries, block counts, in every fp32-class mode, frame-level CTC arg-max ids and encoder output against the CPU oracle
(oracle/paraformer_oracle.py sensevoice_greedy); a frame id may differ only where the oracle's own top-2 log-probabilities are a near-tie.
(2) CT-Transformer punctuation -- random token batches through CTTransformer.punc_forward against oracle/punc_oracle.py.
Not part of the test run. usage: fuzz_gpu_sensevoice_punc_vs_oracle.py [seed] [cases]"""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from funasr_amd import synth                              # noqa: E402
from funasr_amd.ct_transformer import CTTransformer       # noqa: E402
from funasr_amd.sense_voice import SenseVoiceSmall        # noqa: E402
from oracle import paraformer_oracle as O                 # noqa: E402
from oracle import punc_oracle                            # noqa: E402

dev = torch.device("cuda:0")
seed = int(sys.argv[1]) if len(sys.argv) > 1 else 0
n_cases = int(sys.argv[2]) if len(sys.argv) > 2 else 20
g = torch.Generator().manual_seed(seed)


def ri(lo, hi):
    return int(torch.randint(lo, hi + 1, (1,), generator=g))


LANGS = ["auto", "zh", "en", "yue", "ja", "ko", "nospeech"]
sv_bad, sv_worst, sv_ties, sv_frames = 0, 0.0, 0, 0
for ci in range(n_cases):
    tp = ri(0, 2)
    cfg = synth.tiny(synth.SENSEVOICE_SMALL, enc_blocks=ri(1, 3), tp_blocks=tp, vocab=ri(40, 400))
    sd = synth.sensevoice_state_dict(cfg, seed=700 + ci)
    model = SenseVoiceSmall.from_config(cfg)
    model.load_state_dict(sd, strict=False)
    model = model.to(dev)
    B, T = ri(1, 8), ri(1, 260)
    lens = torch.randint(1, T + 1, (B,), generator=g, dtype=torch.int32)
    lens[ri(0, B - 1)] = T
    x = torch.randn(B, T, 560, generator=g) * 0.7
    for b in range(B):
        x[b, lens[b]:] = 0
    lang, tn = LANGS[ri(0, len(LANGS) - 1)], ("withitn", "woitn")[ri(0, 1)]
    ref = O.sensevoice_greedy(x, lens, sd, cfg, language_id=model.lid_dict[lang], textnorm_id=model.textnorm_dict[tn])
    top2 = torch.topk(ref["logp"], 2, dim=-1).values
    for mode in ("fp32", "bf16x3", "f16x2"):
        model.set_precision(mode)
        res = model.recognize_features(x.to(dev), lens, language=lang, textnorm=tn, return_intermediate=True)
        ok = True
        for b in range(B):
            n = int(ref["olens"][b])
            err = (res["enc"][b, :n].cpu() - ref["enc"][b, :n]).abs().max().item()
            sv_worst = max(sv_worst, err)
            got = res["frame_ids"][b][:n] if "frame_ids" in res else None
            want = ref["frame_ids"][b]
            sv_frames += n
            if got is not None and list(got) != list(want):
                for pos, (a_, b_) in enumerate(zip(got, want)):
                    if a_ != b_:
                        gap = float(top2[b, pos, 0] - top2[b, pos, 1])
                        sv_ties += 1
                        ok = ok and gap < 1e-4
            ok = ok and err < 1e-3
        if not ok:
            sv_bad += 1
            print(f"sensevoice case {ci} mode {mode} B={B} T={T} lens={lens.tolist()} lang={lang} tn={tn}: MISMATCH")

# ---------------------------------------------------------------------------------------------- punctuation
enc = dict(input_size=256, output_size=256, attention_heads=8, linear_units=1024, num_blocks=3, kernel_size=11, sanm_shfit=0)
pc_bad, pc_worst = 0, 0.0
for ci in range(n_cases):
    vocab = ri(50, 500)
    e = dict(enc, num_blocks=ri(1, 4))
    sd = punc_oracle.synthetic_state_dict(vocab, e, seed=900 + ci)
    model = CTTransformer(encoder="SANMEncoder", encoder_conf=dict(e, input_layer="pe"), vocab_size=vocab, punc_list=punc_oracle.PUNC_LIST,
                          embed_unit=256, att_unit=256, ignore_id=0, sentence_end_id=3)
    model.load_state_dict(sd, strict=True)
    model = model.to(dev)
    B, N = ri(1, 6), ri(1, 90)
    lens = torch.randint(1, N + 1, (B,), generator=g, dtype=torch.int32)
    lens[0] = N
    ids = torch.randint(1, vocab, (B, N), generator=g, dtype=torch.int64)
    for b in range(B):
        ids[b, lens[b]:] = 0
    want = punc_oracle.punc_forward(ids, lens, sd, e)
    got, _ = model.punc_forward(ids, lens)
    d = max((got[b, : lens[b]].cpu() - want[b, : lens[b]]).abs().max().item() for b in range(B))
    pc_worst = max(pc_worst, d)
    same = all(got[b, : lens[b]].cpu().argmax(-1).tolist() == want[b, : lens[b]].argmax(-1).tolist() for b in range(B))
    if d > 1e-3 or not same:
        pc_bad += 1
        print(f"punctuation case {ci} B={B} N={N} lens={lens.tolist()} blocks={e['num_blocks']}: |d| {d:.2e} argmax equal {same}")
# --------------------------------------------------------- realtime punctuation: causal mask + VAD corner in the last block
from funasr_amd.ct_transformer import CTTransformerStreaming      # noqa: E402
ps_bad, ps_worst = 0, 0.0
for ci in range(n_cases):
    vocab = ri(50, 500)
    e = dict(enc, num_blocks=ri(1, 4))
    sd = punc_oracle.synthetic_state_dict(vocab, e, seed=1100 + ci)
    model = CTTransformerStreaming(encoder="SANMVadEncoder", encoder_conf=dict(e, input_layer="pe"), vocab_size=vocab,
                                   punc_list=punc_oracle.PUNC_LIST, embed_unit=256, att_unit=256, ignore_id=0, sentence_end_id=3)
    model.load_state_dict(sd, strict=True)
    model = model.to(dev)
    B, N = ri(1, 6), ri(1, 90)
    lens = torch.randint(1, N + 1, (B,), generator=g, dtype=torch.int32)
    lens[0] = N
    ids = torch.randint(1, vocab, (B, N), generator=g, dtype=torch.int64)
    for b in range(B):
        ids[b, lens[b]:] = 0
    vad = torch.tensor([ri(0, N + 2) for _ in range(B)], dtype=torch.int32)      # 0 / >= T: no corner (mask.py:38-52)
    want = punc_oracle.punc_forward_vad(ids, lens, vad, sd, e)
    got, _ = model.punc_forward(ids, lens, vad)
    d = max((got[b, : lens[b]].cpu() - want[b, : lens[b]]).abs().max().item() for b in range(B))
    ps_worst = max(ps_worst, d)
    if d > 1e-3:
        ps_bad += 1
        print(f"realtime punctuation case {ci} B={B} N={N} lens={lens.tolist()} vad={vad.tolist()} blocks={e['num_blocks']}: |d| {d:.2e}")
pc_bad += ps_bad
print(json.dumps(dict(tool="fuzz_gpu_sensevoice_punc_vs_oracle", seed=seed, cases=n_cases, punctuation_realtime=dict(bad=ps_bad, worst_logit_abs_diff=ps_worst),
                      sensevoice=dict(bad=sv_bad, worst_encoder_abs_diff=sv_worst, frames=sv_frames, near_tie_frame_flips=sv_ties),
                      punctuation=dict(bad=pc_bad, worst_logit_abs_diff=pc_worst))))
sys.exit(1 if (sv_bad or pc_bad) else 0)
