"""Host-side helpers of the VAD-segmented pipeline (no GPU work).

`merge_vad` restates funasr/utils/vad_utils.py:54-89 (glue neighbouring VAD segments up to a maximum length),
`vad_segment_sentences` restates `_vad_segment_sentences` (funasr/auto/auto_model.py:71-105): one sentence record per
decoded VAD segment, spanning its token timestamps when there are any and the segment bounds otherwise.
"""
from __future__ import annotations

import re
from typing import List


def merge_vad(vad_result: List[List[int]], max_length: int = 15000, min_length: int = 0) -> List[List[int]]:
    if len(vad_result) <= 1:
        return vad_result
    cuts = sorted({t for seg in vad_result for t in seg[:2]})
    merged: List[List[int]] = []
    begin = 0
    for here, nxt in zip(cuts[:-1], cuts[1:]):
        if nxt - begin < max_length:
            continue                              # the next boundary still fits: keep growing
        if here - begin > min_length:
            merged.append([begin, here])
        begin = here
    merged.append([begin, cuts[-1]])
    return merged


def vad_segment_sentences(decoded: List[dict], segments: List[List[int]]) -> List[dict]:
    sentences = []
    for res, seg in zip(decoded, segments):
        text = re.sub(r"<\|[^|]*\|>", "", str(res.get("text", ""))).strip()
        if not text:
            continue
        raw = res.get("timestamp")
        if raw is None:
            raw = res.get("timestamps", [])
        stamps = []
        for item in raw or []:
            if isinstance(item, dict):
                if item.get("start_time") is None or item.get("end_time") is None:
                    continue
                stamps.append([int(float(item["start_time"]) * 1000), int(float(item["end_time"]) * 1000)])
            elif isinstance(item, (list, tuple)) and len(item) >= 2:
                stamps.append([int(item[0]), int(item[1])])
        sentences.append({"start": stamps[0][0] if stamps else seg[0], "end": stamps[-1][1] if stamps else seg[1],
                          "text": text, "sentence": text, "timestamp": stamps})
    return sentences


def join_vad_texts(texts) -> str:
    """`_join_vad_texts` (funasr/auto/auto_model.py:56-68): rich tags removed, chunks joined by a blank unless both sides
    of the seam are CJK characters."""
    cleaned = [re.sub(r"<\|[^|]*\|>", "", t).strip() for t in texts]
    cleaned = [t for t in cleaned if t]
    if not cleaned:
        return ""
    joined = cleaned[0]
    for t in cleaned[1:]:
        cjk_seam = "㐀" <= joined[-1] <= "鿿" and "㐀" <= t[0] <= "鿿"
        joined += ("" if cjk_seam else " ") + t
    return joined
