"""Keeping the ASR surface text when punctuation is added (host side).

ASR results that carry `words` next to `timestamp` (SenseVoice with word timestamps) are punctuated WITHOUT re-spelling
them: `AutoModel.inference_with_vad` (funasr/auto/auto_model.py:1046-1121) inserts the predicted marks into the original
string, re-times the ASR units onto the punctuation model's tokens when the two tokenisations differ, and cuts sentence
records on the original characters. Restated from the helpers at funasr/auto/auto_model.py:107-300
(`_get_punc_tokens`, `_surface_token_spans`, `_punc_symbol`, `_punctuate_surface_text`, `_merge_timestamp_units`,
`_timestamp_sentences_from_surface`); every function returns None when the two sides cannot be aligned, and the caller
falls back to the plain path. Pinned by the scenarios of the reference's tests/test_punc_model_none.py
(tests/test_punc_pipeline_spec.py) and fuzzed against the reference helpers (oracle/fuzz_text_vs_reference.py).
"""
from __future__ import annotations

from typing import List, Optional, Sequence, Tuple

from .ct_transformer import split_words

_FALLBACK_MARKS = {1: "", 2: "，", 3: "。", 4: "？", 5: "、"}
_ASCII_MARKS = {"，": ",", "。": ".", "？": "?", "、": ","}


def _size(x) -> int:
    try:
        return len(x)
    except TypeError:
        return -1


def punc_tokens(text: str, punc_array, punc_model) -> Optional[List[str]]:
    """the units one punctuation id stands for: the punctuation model's words, multi-character CJK words (jieba models)
    spelled out; None unless there is exactly one per id"""
    try:
        words = split_words(text, jieba_usr_dict=getattr(punc_model, "jieba_usr_dict", None))
    except Exception:  # noqa: BLE001 - any tokenisation problem means "cannot align", like the reference
        return None
    units: List[str] = []
    for w in words:
        if w and "฀" <= w[0] <= "龥" and len(w) > 1:
            units.extend(w)
        else:
            units.append(w)
    return units if _size(punc_array) == len(units) else None


def surface_token_spans(text: str, tokens: Sequence[str]) -> Optional[List[Tuple[int, int]]]:
    """[begin, end) of every token in `text` (blanks between tokens skipped, case ignored); None on any mismatch or left-over"""
    spans, pos = [], 0
    for tok in tokens:
        while pos < len(text) and text[pos].isspace():
            pos += 1
        if text[pos: pos + len(tok)].casefold() != tok.casefold():
            return None
        spans.append((pos, pos + len(tok)))
        pos += len(tok)
    return None if text[pos:].strip() else spans


def punc_symbol(punc_id, token: str, punc_model) -> str:
    """the mark of a punctuation id after `token`: the model's punc_list (or the CT-Transformer default), `_` = none, ASCII
    marks after ASCII tokens"""
    marks = getattr(punc_model, "punc_list", None)
    pid = int(punc_id)
    try:
        mark = marks[pid]
    except (IndexError, TypeError):
        mark = _FALLBACK_MARKS.get(pid, "")
    if mark == "_":
        mark = ""
    if mark and token[0].isascii():
        mark = _ASCII_MARKS.get(mark, mark)
    return mark


def punctuate_surface_text(text: str, punc_array, punc_model) -> Optional[str]:
    tokens = punc_tokens(text, punc_array, punc_model)
    spans = surface_token_spans(text, tokens) if tokens is not None else None
    if spans is None:
        return None
    out, pos = [], 0
    for tok, pid, (_, end) in zip(tokens, punc_array, spans):
        out.append(text[pos:end])
        out.append(punc_symbol(pid, tok, punc_model))
        pos = end
    out.append(text[pos:])
    return "".join(out)


def _squash(s: str) -> str:
    return "".join(s.split()).casefold()


def merge_timestamp_units(text: str, words: Sequence[str], timestamps, punc_array, punc_model):
    """ASR units (words / BPE pieces with [beg, end]) -> one [beg, end] per punctuation token: every unit's span is divided
    evenly over its characters (integer arithmetic), a token takes the first character's start and the last character's end.
    -> (" ".join(tokens), timestamps) or None"""
    tokens = punc_tokens(text, punc_array, punc_model)
    if tokens is None or len(words) != len(timestamps):
        return None
    chars, per_char = "", []
    for word, stamp in zip(words, timestamps):
        w = _squash(word)
        if not w or not isinstance(stamp, (list, tuple)) or len(stamp) < 2 or stamp[1] < stamp[0]:
            return None
        beg, end = stamp[:2]
        span = end - beg
        chars += w
        per_char += [[beg + span * i // len(w), beg + span * (i + 1) // len(w)] for i in range(len(w))]
    merged, pos = [], 0
    for tok in tokens:
        t = _squash(tok)
        stop = pos + len(t)
        if not t or chars[pos:stop] != t or stop > len(per_char):
            return None
        merged.append([per_char[pos][0], per_char[stop - 1][1]])
        pos = stop
    if pos != len(per_char):
        return None
    return " ".join(tokens), merged


def timestamp_sentences_from_surface(text: str, timestamps, punc_array, punc_model, return_raw_text: bool = False):
    """sentence records cut at every predicted mark, spelled with the characters of `text` itself"""
    tokens = punc_tokens(text, punc_array, punc_model)
    if tokens is None or len(tokens) != len(timestamps):
        return None
    spans = surface_token_spans(text, tokens)
    if spans is None:
        return None

    def record(first: int, last: int, mark: str) -> dict:
        raw = text[spans[first][0]: spans[last][1]].strip()
        rec = {"text": raw + mark, "start": timestamps[first][0], "end": timestamps[last][1],
               "timestamp": timestamps[first: last + 1]}
        if return_raw_text:
            rec["raw_text"] = raw
        return rec

    out, first = [], 0
    for i, (tok, pid) in enumerate(zip(tokens, punc_array)):
        mark = punc_symbol(pid, tok, punc_model)
        if mark:
            out.append(record(first, i, mark))
            first = i + 1
    if first < len(tokens):
        out.append(record(first, len(tokens) - 1, ""))
    return out
