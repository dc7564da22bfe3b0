"""Host-side timestamp path against golden vectors produced by the reference's own functions
(oracle/make_golden_timestamps.py: ts_prediction_lfr6_standard and sentence_postprocess with / without timestamps)."""
import json
import os

import torch

from funasr_amd.timestamps import cif_timestamps
from funasr_amd.tokenizer import sentence_postprocess

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "timestamps.json")


def _cases():
    with open(GOLD, encoding="utf-8") as f:
        return json.load(f)["cases"]


def test_cif_timestamps_equal_reference_strings_and_milliseconds():
    cases = _cases()
    assert len(cases) == 60 and sum(1 for i in range(60) if i % 5 == 4) == 12      # 12 go through the refire branch
    for c in cases:
        a = torch.tensor(c["alphas"], dtype=torch.float32)
        p = torch.tensor(c["peaks"], dtype=torch.float32)
        txt, ms = cif_timestamps(a, p, c["tokens"] + c["tail"], vad_offset=c["vad_offset"], upsample_rate=c["upsample_rate"])
        assert txt == c["text"], (c["tokens"], txt, c["text"])
        assert ms == c["ms"]
        # batched [1, T] form (paraformer/model.py:669 passes rows of the batch tensors)
        txt2, ms2 = cif_timestamps(a[None], p[None], c["tokens"] + c["tail"], vad_offset=c["vad_offset"],
                                   upsample_rate=c["upsample_rate"])
        assert (txt2, ms2) == (txt, ms)
    assert cif_timestamps(torch.zeros(4), torch.zeros(4), []) == ("", [])


def test_sentence_postprocess_with_and_without_timestamps_equals_reference():
    kinds = set()
    for c in _cases():
        kinds.add(c["kind"])
        sent, words = sentence_postprocess(c["tokens"])
        assert sent == c["sentence"] and words == c["words"], (c["tokens"], sent, c["sentence"])
        s2, spans, w2 = sentence_postprocess(c["tokens"], [list(x) for x in c["ms"]])
        assert s2 == c["sentence_ts"] and spans == c["spans_ts"] and w2 == c["words_ts"], (c["tokens"], spans, c["spans_ts"])
    assert kinds == {"cjk", "alpha", "mixed"}


def test_vectorised_token_spans_equal_the_reference_restatement():
    """cif_token_spans (what the model classes call) == the per-token restatement of ts_prediction_lfr6_standard, on the
    reference goldens and on 1500 random weight tracks incl. long gaps, short edges, vad offsets and count mismatches"""
    import random
    from funasr_amd.timestamps import cif_timestamps, cif_token_spans
    for c in _cases():                                           # the reference's own outputs
        a = torch.tensor(c["alphas"], dtype=torch.float32)
        p = torch.tensor(c["peaks"], dtype=torch.float32)
        assert cif_token_spans(a, p, c["tokens"] + c["tail"], vad_offset=c["vad_offset"], upsample_rate=c["upsample_rate"]) == c["ms"]
    rng = random.Random(3)
    g = torch.Generator().manual_seed(3)
    checked = 0
    for trial in range(1500):
        T = rng.randint(4, 400)
        up = rng.choice((1, 3))
        a = torch.rand(T, generator=g) * rng.choice((0.05, 0.2, 0.6))
        if rng.random() < 0.5:                                   # silent stretches -> gaps longer than 12 frames
            a[rng.randint(0, T - 1):] *= 0.0 if rng.random() < 0.3 else 0.02
        integrate, peaks = 0.0, []
        thr = float(torch.tensor(1.0 - 1e-4, dtype=torch.float32))
        acc = torch.zeros(())
        for t in range(T):
            acc = acc + a[t]
            peaks.append(float(acc))
            if float(acc) >= thr:
                acc = acc - torch.tensor(thr)
        peaks = torch.tensor(peaks, dtype=torch.float32)
        n_fire = int((peaks >= thr).sum())
        n_tok = max(n_fire - 1 + rng.choice((0, 0, 0, 1, -1)), 0)
        toks = ["t%d" % i for i in range(n_tok)] + (["</s>"] if rng.random() < 0.3 else [])
        off = rng.choice((0, 0, 1230, 60000))
        try:
            want = cif_timestamps(a, peaks, list(toks), vad_offset=off, upsample_rate=up)[1]
        except Exception as e:  # noqa: BLE001
            want = type(e).__name__
        try:
            got = cif_token_spans(a, peaks, list(toks), vad_offset=off, upsample_rate=up)
        except Exception as e:  # noqa: BLE001
            got = type(e).__name__
        assert got == want, (trial, T, n_fire, n_tok, off, got if isinstance(got, str) else got[-2:], want if isinstance(want, str) else want[-2:])
        checked += not isinstance(want, str)
    assert checked > 1000
