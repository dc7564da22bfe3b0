"""Micro character/token error rate, the definition the reference uses to report CER
(runtime/llama.cpp/benchmarks/compute_cer.py:17-31: sum of edit distances / sum of reference lengths, Chinese text
normalised by dropping everything but word characters and CJK and upper-casing) and the recipe's
examples/aishell/paraformer/utils/compute_wer.py. Host-side only."""
from __future__ import annotations

import re
from typing import Iterable, Sequence, Tuple


def normalize_zh(text: str) -> str:
    text = re.sub(r"<\|[^|]*\|>", "", text)                 # SenseVoice rich-transcription tags
    return re.sub(r"[^\w一-鿿]", "", text).upper()


def edit_distance(ref: Sequence, hyp: Sequence) -> int:
    """Levenshtein distance (substitutions, insertions, deletions all cost 1), two-row DP."""
    ref, hyp = list(ref), list(hyp)
    if not ref:
        return len(hyp)
    prev = list(range(len(hyp) + 1))
    for i in range(1, len(ref) + 1):
        cur = [i] + [0] * len(hyp)
        ri = ref[i - 1]
        for j in range(1, len(hyp) + 1):
            cur[j] = min(prev[j] + 1, cur[j - 1] + 1, prev[j - 1] + (ri != hyp[j - 1]))
        prev = cur
    return prev[len(hyp)]


def micro_error_rate(refs: Iterable[Sequence], hyps: Iterable[Sequence]) -> Tuple[float, int, int]:
    """(error rate, total edits, total reference units) over paired sequences (token ids or characters)."""
    edits = total = 0
    for r, h in zip(refs, hyps):
        edits += edit_distance(r, h)
        total += len(r)
    return (edits / total if total else 0.0), edits, total


def cer(ref_texts: Iterable[str], hyp_texts: Iterable[str]) -> float:
    return micro_error_rate([normalize_zh(r) for r in ref_texts], [normalize_zh(h) for h in hyp_texts])[0]
