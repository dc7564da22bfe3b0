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
