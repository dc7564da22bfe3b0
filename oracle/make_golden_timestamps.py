#!/usr/bin/env python3
"""Golden vectors for the host-side timestamp path, produced by the REFERENCE's own functions
(funasr/utils/timestamp_tools.py:37-122 `ts_prediction_lfr6_standard`, funasr/utils/postprocess_utils.py:165-278
`sentence_postprocess` with and without timestamps) on seeded inputs. TEST INFRASTRUCTURE: run in the build container
(needs /root/reference); writes tests/golden/timestamps.json, which travels to the GPU box.

Cases: CIF alphas / peaks drawn like the predictor's output (fires every 2-30 frames, with leading / trailing silence,
long gaps that get cut into token + <sil>), both the matching case (#fires == #tokens + 1) and the mismatch case that
goes through the rescale-and-refire branch; token lists in pure CJK, pure alphabetic BPE ("@@" pieces, spelled
abbreviations) and mixed script.
"""
import json
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
from oracle import ref_import  # noqa: E402


def cif_like(rng, n_tokens, lead, tail, long_gap_at=None):
    """alphas whose running sum crosses an integer n_tokens + 1 times; peaks as cif_wo_hidden_v1 would report them."""
    alphas = [np.zeros(lead, np.float32)]
    for k in range(n_tokens + 1):
        gap = int(rng.integers(2, 9))
        if long_gap_at is not None and k == long_gap_at:
            gap = int(rng.integers(16, 30))
        w = rng.random(gap).astype(np.float32) + 0.05
        w = w / w.sum() * np.float32(1.0)
        alphas.append(w.astype(np.float32))
    alphas.append(np.zeros(tail, np.float32))
    a = torch.from_numpy(np.concatenate(alphas))
    cs = torch.cumsum(a.double(), 0).float()
    fl = torch.floor(cs + 1e-4)                          # the unit-mass segments sum to 1 up to fp32 rounding
    prev = torch.cat([torch.zeros(1), fl[:-1]])
    fire = (fl - prev) > 0
    peaks = fire.float() + (cs - torch.floor(cs)) * (~fire).float() * 0.5
    return a, peaks


def main():
    ref_import.install()
    from funasr.utils.timestamp_tools import ts_prediction_lfr6_standard
    from funasr.utils.postprocess_utils import sentence_postprocess
    rng = np.random.default_rng(20240923)
    cjk = list("今天天气真不错我们一起去公园散步吧欢迎大家来体验语音识别模型")
    alpha_words = ["hel@@", "lo", "wor@@", "ld", "the", "quick", "brown", "fox", "a", "i", "b", "m", "c", "d", "don't",
                   "spe@@", "ech", "re@@", "cog@@", "ni@@", "tion", "x", "y"]
    cases = []
    for ci in range(60):
        kind = ("cjk", "alpha", "mixed")[ci % 3]
        n = int(rng.integers(1, 24))
        if kind == "cjk":
            toks = [cjk[int(rng.integers(len(cjk)))] for _ in range(n)]
        elif kind == "alpha":
            toks = [alpha_words[int(rng.integers(len(alpha_words)))] for _ in range(n)]
            if toks[-1].endswith("@@"):
                toks[-1] = "fox"
        else:
            toks = [(cjk[int(rng.integers(len(cjk)))] if rng.random() < 0.5 else alpha_words[int(rng.integers(len(alpha_words)))])
                    for _ in range(n)]
            if toks[-1].endswith("@@"):
                toks[-1] = "好"
            if rng.random() < 0.3:
                toks.insert(int(rng.integers(len(toks))), "3d")          # neither CJK, a piece nor alphabetic
        n = len(toks)
        lead, tail = int(rng.integers(0, 14)), int(rng.integers(0, 14))
        a, p = cif_like(rng, n, lead, tail, long_gap_at=int(rng.integers(n)) if rng.random() < 0.4 else None)
        mismatch = ci % 5 == 4
        if mismatch:                                       # drop / add a fire so that the refire branch runs
            idx = torch.where(p >= 1.0 - 1e-4)[0]
            p = p.clone()
            p[idx[int(rng.integers(len(idx)))]] = 0.3
        extra = ["</s>"] if ci % 7 == 0 else []
        vad_offset = float(rng.integers(0, 3)) * 1234.0
        ups = 1 if ci % 2 == 0 else 3
        txt, ms = ts_prediction_lfr6_standard(a.clone(), p.clone(), list(toks) + extra, vad_offset=vad_offset, upsample_rate=ups)
        case = dict(kind=kind, tokens=toks, tail=extra, alphas=[float(x) for x in a.tolist()], peaks=[float(x) for x in p.tolist()],
                    vad_offset=vad_offset, upsample_rate=ups, text=txt, ms=ms)
        sent, words = sentence_postprocess(list(toks))
        case["sentence"], case["words"] = sent, words
        if len(ms) == len(toks):
            s2, ts2, w2 = sentence_postprocess(list(toks), [list(x) for x in ms])
            case["sentence_ts"], case["spans_ts"], case["words_ts"] = s2, ts2, w2
        cases.append(case)
    # merge_vad (funasr/utils/vad_utils.py:54-89) and the sentence records of _vad_segment_sentences (auto_model.py:71-105)
    from funasr.utils.vad_utils import merge_vad
    merge_cases = []
    for mi in range(24):
        n = int(rng.integers(0, 12))
        t, segs = int(rng.integers(0, 500)), []
        for _ in range(n):
            d = int(rng.integers(200, 9000))
            segs.append([t, t + d])
            t += d + int(rng.integers(50, 3000))
        mx = int(rng.choice([5000, 15000, 30000]))
        mn = int(rng.choice([0, 0, 1000]))
        merge_cases.append(dict(segments=segs, max_length=mx, min_length=mn,
                                merged=merge_vad([list(x) for x in segs], max_length=mx, min_length=mn)))
    out = os.path.join(os.path.dirname(HERE), "tests", "golden", "timestamps.json")
    with open(out, "w", encoding="utf-8") as f:
        json.dump(dict(generator="oracle/make_golden_timestamps.py", reference="funasr/utils/timestamp_tools.py:37-122, "
                       "funasr/utils/postprocess_utils.py:165-278, funasr/utils/vad_utils.py:54-89", cases=cases, merge_vad=merge_cases),
                  f, ensure_ascii=False)
    n_ts = sum(1 for c in cases if "spans_ts" in c)
    print(f"wrote {out}: {len(cases)} cases, {n_ts} with timestamped post-processing, "
          f"{sum(1 for i in range(len(cases)) if i % 5 == 4)} through the refire branch")


if __name__ == "__main__":
    main()
