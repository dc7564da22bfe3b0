"""Text post-processing on the host, under the reference's module name.

`sentence_postprocess` (funasr/utils/postprocess_utils.py:165-278) lives in funasr_amd/tokenizer.py and is re-exported here;
this module adds the two remaining helpers users of this path call on its results:
  * `rich_transcription_postprocess` (:436-480) -- SenseVoice's tagged output (`<|zh|><|HAPPY|><|Speech|><|woitn|>text`)
    to plain text with emoji for emotions / events, the documented last step of the SenseVoice recipe;
  * `sentence_postprocess_sentencepiece` (:281-335) -- sentencepiece pieces (`▁` word starts) to a sentence and its words,
    used by the English Paraformer recipes.
Both are pinned to the reference functions by tests/golden/postprocess.json (oracle/make_golden_postprocess.py) and fuzzed
against them by oracle/fuzz_text_vs_reference.py.
"""
from __future__ import annotations

from typing import Iterable, List, Optional, Tuple, Union

from .tokenizer import sentence_postprocess  # noqa: F401  (same import path as the reference module offers)

_LANG_TAGS = ("<|zh|>", "<|en|>", "<|yue|>", "<|ja|>", "<|ko|>", "<|nospeech|>")
_EMOTION = {"<|HAPPY|>": "😊", "<|SAD|>": "😔", "<|ANGRY|>": "😡", "<|NEUTRAL|>": "", "<|FEARFUL|>": "😰",
            "<|DISGUSTED|>": "🤢", "<|SURPRISED|>": "😮"}
_EVENT = {"<|BGM|>": "🎼", "<|Speech|>": "", "<|Applause|>": "👏", "<|Laughter|>": "😀", "<|Cry|>": "😭", "<|Sneeze|>": "🤧",
          "<|Breath|>": "", "<|Cough|>": "🤧"}
# every tag that is counted and stripped from a segment, in the reference's order (:365-395)
_STRIPPED = ("<|nospeech|><|Event_UNK|>",) + _LANG_TAGS + (
    "<|HAPPY|>", "<|SAD|>", "<|ANGRY|>", "<|NEUTRAL|>", "<|BGM|>", "<|Speech|>", "<|Applause|>", "<|Laughter|>", "<|FEARFUL|>",
    "<|DISGUSTED|>", "<|SURPRISED|>", "<|Cry|>", "<|EMO_UNKNOWN|>", "<|Sneeze|>", "<|Breath|>", "<|Cough|>", "<|Sing|>",
    "<|Speech_Noise|>", "<|withitn|>", "<|woitn|>", "<|GBG|>", "<|Event_UNK|>")
_EMOTION_MARKS = ("😊", "😔", "😡", "😰", "🤢", "😮")
_EVENT_MARKS = ("🎼", "👏", "😀", "😭", "🤧", "😷")


def _format_segment(seg: str) -> str:
    """one language segment: tags out, event marks in front (one per kind that occurred), the dominant emotion behind
    (`format_str_v2`, :411-433)"""
    count = {}
    for tag in _STRIPPED:
        count[tag] = seg.count(tag)
        seg = seg.replace(tag, "")
    emotion = "<|NEUTRAL|>"
    for tag in _EMOTION:
        if count[tag] > count[emotion]:
            emotion = tag
    for tag, mark in _EVENT.items():
        if count[tag] > 0:
            seg = mark + seg
    seg += _EMOTION[emotion]
    for mark in _EMOTION_MARKS + _EVENT_MARKS:
        seg = seg.replace(" " + mark, mark).replace(mark + " ", mark)
    return seg.strip()


def _leading_event(s: str) -> Optional[str]:
    return s[0] if s[0] in _EVENT_MARKS else None


def _trailing_emotion(s: str) -> Optional[str]:
    return s[-1] if s[-1] in _EMOTION_MARKS else None


def rich_transcription_postprocess(s: str) -> str:
    s = s.replace("<|nospeech|><|Event_UNK|>", "❓")
    for tag in _LANG_TAGS:
        s = s.replace(tag, "<|lang|>")
    segments = [_format_segment(part).strip(" ") for part in s.split("<|lang|>")]
    text = " " + segments[0]
    event = _leading_event(text)
    for seg in segments[1:]:
        if not seg:
            continue
        if event is not None and _leading_event(seg) == event:      # the same event continues: keep one mark
            seg = seg[1:]
            if not seg:
                continue
        event = _leading_event(seg)
        emotion = _trailing_emotion(seg)
        if emotion is not None and emotion == _trailing_emotion(text):   # the same emotion continues: keep the last mark
            text = text[:-1]
        text += seg.strip()
    return text.replace("The.", " ").strip()


def sentence_postprocess_sentencepiece(words: Iterable[Union[str, bytes]]) -> Tuple[str, List[str]]:
    """pieces -> (sentence, words): `▁` starts a word, special symbols are dropped, a lone "i" and its contractions are
    capitalised in the word list (not in the sentence, like the reference)"""
    pieces = [w if isinstance(w, str) else w.decode("utf-8") for w in words]
    pieces = [p for p in pieces if p not in ("<s>", "</s>", "<unk>", "<OOV>")]
    parts: List[str] = []
    current = ""
    for i, piece in enumerate(pieces):
        if "▁" in piece:
            if i != 0:
                parts += [current, " "]
            current = piece.replace("▁", "")
        else:
            current += piece
    parts.append(current)
    fix = {"i": "I", "i'm": "I'm", "i've": "I've", "i'll": "I'll"}
    return "".join(parts), [fix.get(p, p) for p in parts if p != " "]
