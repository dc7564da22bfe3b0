"""Text-level hotword correction applied to finished results (host side).

Mirror of funasr/utils/postprocess_hotwords.py (the last step of `AutoModel.generate`, funasr/auto/auto_model.py:742-748):
not the model-level `hotword` biasing of SeACo, but a search-and-replace on the final text. Sources are a string (one entry
per line), a list, a dict {wrong: right} or a file; an entry is either an explicit mapping `wrong=>right` (also `->`, `→`) or a
bare target word that is matched FUZZILY by pinyin similarity (needs `pypinyin` and `rapidfuzz`; a clear ImportError without
them, like the reference). Only `text`, and `text` / `sentence` inside `sentence_info`, change; timestamps stay aligned to
the original recognition. Same public names as the reference module; pinned by the scenarios of the reference's
tests/test_postprocess_hotwords.py and, for the fuzzy search, against the reference module itself with stand-in
pinyin / ratio functions (tests/test_postprocess_hotwords.py).
"""
from __future__ import annotations

import os
import re
from dataclasses import asdict, dataclass
from typing import Any, Dict, Iterable, List, Mapping, Optional, Sequence, Tuple

_ARROWS = ("=>", "->", "→")
_HAS_WORD = re.compile(r"[一-鿿]|[a-zA-Z]+|[0-9]+")
_pinyin = None      # (lazy_pinyin, Style) once imported
_fuzz = None        # rapidfuzz.fuzz once imported


@dataclass(frozen=True)
class HotwordMatch:
    original: str
    replacement: str
    score: float
    start: int
    end: int

    def as_dict(self) -> Dict[str, Any]:
        return asdict(self)


def _require_pypinyin():
    global _pinyin
    if _pinyin is None:
        try:
            from pypinyin import Style, lazy_pinyin
        except ImportError as exc:
            raise ImportError("postprocess hotword fuzzy matching requires pypinyin. Install it with: pip install pypinyin") from exc
        _pinyin = (lazy_pinyin, Style)
    return _pinyin


def _require_rapidfuzz():
    global _fuzz
    if _fuzz is None:
        try:
            from rapidfuzz import fuzz
        except ImportError as exc:
            raise ImportError("postprocess hotword fuzzy matching requires rapidfuzz. Install it with: pip install rapidfuzz") from exc
        _fuzz = fuzz
    return _fuzz


def _to_pinyin_key(text: str) -> str:
    lazy_pinyin, style = _require_pypinyin()
    return "".join(lazy_pinyin(text, style=style.NORMAL, errors="ignore")).lower()


def _entry(line: str) -> Tuple[Optional[str], Optional[str]]:
    """one source line -> (wrong, right): (None, None) = nothing, (None, w) = fuzzy target, (a, b) = explicit mapping"""
    line = line.strip()
    if not line or line.startswith("#"):
        return None, None
    for arrow in _ARROWS:
        if arrow in line:
            wrong, right = (part.strip() for part in line.split(arrow, 1))
            return (wrong, right) if wrong and right else (None, None)
    return None, line


def _collect(lines: Iterable[str], explicit: Dict[str, str], fuzzy: List[str]) -> None:
    for line in lines:
        wrong, right = _entry(line)
        if not right:
            continue
        if wrong is not None:
            explicit[wrong] = right
        else:
            fuzzy.append(right)


def parse_hotword_file(path: str) -> Tuple[Dict[str, str], List[str]]:
    if not os.path.isfile(path):
        raise FileNotFoundError(f"postprocess_hotword_file not found: {path}")
    explicit: Dict[str, str] = {}
    fuzzy: List[str] = []
    with open(path, "r", encoding="utf-8") as f:
        _collect(f, explicit, fuzzy)
    return explicit, fuzzy


def parse_postprocess_hotwords(postprocess_hotwords) -> Tuple[Dict[str, str], List[str]]:
    explicit: Dict[str, str] = {}
    fuzzy: List[str] = []
    src = postprocess_hotwords
    if src is None:
        pass
    elif isinstance(src, str):
        _collect(src.splitlines(), explicit, fuzzy)
    elif isinstance(src, Mapping):
        for wrong, right in src.items():
            wrong, right = str(wrong).strip(), str(right).strip()
            if not right:
                continue
            if wrong and wrong != right:
                explicit[wrong] = right
            else:
                fuzzy.append(right)
    elif isinstance(src, Sequence) and not isinstance(src, (str, bytes)):
        _collect((str(item) for item in src if item is not None), explicit, fuzzy)
    else:
        raise TypeError(f"postprocess_hotwords must be None, str, list, or dict; got {type(src)!r}")
    return explicit, fuzzy


def _select_non_overlapping(candidates: List[HotwordMatch]) -> List[HotwordMatch]:
    """best score first (longer window on ties), a candidate is dropped when it touches an already chosen span"""
    taken: List[HotwordMatch] = []
    for cand in sorted(candidates, key=lambda m: (m.score, m.end - m.start), reverse=True):
        if all(cand.end <= t.start or cand.start >= t.end for t in taken):
            taken.append(cand)
    return sorted(taken, key=lambda m: m.start)


class PostprocessHotwordMatcher:
    def __init__(self, explicit_map: Optional[Dict[str, str]] = None, fuzzy_targets: Optional[Iterable[str]] = None,
                 threshold: float = 0.85, enable_fuzzy: bool = True):
        self.explicit_map = dict(explicit_map or {})
        self.threshold = float(threshold)
        if not 0.0 <= self.threshold <= 1.0:
            raise ValueError(f"postprocess_hotword_threshold must be between 0.0 and 1.0, got {threshold}")
        self.enable_fuzzy = bool(enable_fuzzy)
        self.fuzzy_targets: List[str] = list(dict.fromkeys(t for t in (str(x).strip() for x in (fuzzy_targets or [])) if t))
        self._length_buckets: Dict[int, List[Tuple[str, str]]] = {}
        self._fuzz = None
        if self.fuzzy_targets and self.enable_fuzzy:
            self._fuzz = _require_rapidfuzz()
            _require_pypinyin()
            for target in self.fuzzy_targets:
                self._length_buckets.setdefault(len(target), []).append((target, _to_pinyin_key(target)))

    # ---------------------------------------------------------------------------------------------------- text
    def apply_text(self, text: str) -> Tuple[str, List[HotwordMatch]]:
        if not text:
            return text, []
        matches: List[HotwordMatch] = []
        text = self._apply_explicit(text, matches)
        if self.fuzzy_targets and self.enable_fuzzy:
            text, fuzzy = self._apply_fuzzy(text)
            matches.extend(fuzzy)
        return text, matches

    def _apply_explicit(self, text: str, matches: List[HotwordMatch]) -> str:
        for wrong in sorted(self.explicit_map, key=len, reverse=True):          # longest first; positions refer to the
            right = self.explicit_map[wrong]                                     # text as edited so far
            pos = text.find(wrong)
            while pos >= 0:
                matches.append(HotwordMatch(wrong, right, 1.0, pos, pos + len(wrong)))
                text = text[:pos] + right + text[pos + len(wrong):]
                pos = text.find(wrong, pos + len(right))
        return text

    def _apply_fuzzy(self, text: str) -> Tuple[str, List[HotwordMatch]]:
        if not self._length_buckets:
            return text, []
        shortest, longest = min(self._length_buckets), max(self._length_buckets)
        found: List[HotwordMatch] = []
        for width in range(max(1, shortest - 1), longest + 2):                  # windows one character shorter .. longer
            lengths = [n for n in (width - 1, width, width + 1) if n in self._length_buckets]
            if not lengths:
                continue
            for start in range(0, len(text) - width + 1):
                window = text[start: start + width]
                if not window or not _HAS_WORD.search(window):
                    continue
                key = _to_pinyin_key(window)
                for n in lengths:
                    for target, target_key in self._length_buckets[n]:
                        if window == target:
                            continue
                        score = self._fuzz.ratio(key, target_key) / 100.0
                        if score >= self.threshold:
                            found.append(HotwordMatch(window, target, round(score, 4), start, start + width))
        if not found:
            return text, []
        chosen = _select_non_overlapping(found)
        for m in sorted(chosen, key=lambda m: m.start, reverse=True):           # right to left: earlier offsets stay valid
            text = text[: m.start] + m.replacement + text[m.end:]
        return text, chosen

    # -------------------------------------------------------------------------------------------------- result
    def apply_result(self, result: Dict[str, Any], return_matches: bool = False) -> Dict[str, Any]:
        text = result.get("text", "")
        if not isinstance(text, str) or not text:
            if return_matches:
                result["postprocess_hotword_matches"] = []
            return result
        stamps = result.get("timestamp")
        result["text"], matches = self.apply_text(text)
        info = result.get("sentence_info")
        if isinstance(info, list):
            for sentence in info:
                if isinstance(sentence, dict):
                    for field in ("text", "sentence"):
                        if isinstance(sentence.get(field), str):
                            sentence[field] = self.apply_text(sentence[field])[0]
        if return_matches:
            result["postprocess_hotword_matches"] = [m.as_dict() for m in matches]
        if stamps is not None:
            result["timestamp"] = stamps
        return result


def build_postprocess_hotword_matcher(postprocess_hotwords=None, postprocess_hotword_file: Optional[str] = None,
                                      postprocess_hotword_threshold: float = 0.85, enable_fuzzy: bool = True):
    explicit: Dict[str, str] = {}
    fuzzy: List[str] = []
    if postprocess_hotwords is not None:
        e, f = parse_postprocess_hotwords(postprocess_hotwords)
        explicit.update(e)
        fuzzy.extend(f)
    if postprocess_hotword_file:
        e, f = parse_hotword_file(postprocess_hotword_file)
        explicit.update(e)
        fuzzy.extend(f)
    if not explicit and not fuzzy:
        return None
    return PostprocessHotwordMatcher(explicit_map=explicit, fuzzy_targets=fuzzy, threshold=postprocess_hotword_threshold,
                                     enable_fuzzy=enable_fuzzy)


def apply_postprocess_hotwords_to_results(results: List[Dict[str, Any]], cfg: Mapping[str, Any]) -> List[Dict[str, Any]]:
    """one matcher per call, applied to every result dict in place; `results` itself is returned untouched when nothing is
    configured"""
    matcher = build_postprocess_hotword_matcher(
        postprocess_hotwords=cfg.get("postprocess_hotwords"), postprocess_hotword_file=cfg.get("postprocess_hotword_file"),
        postprocess_hotword_threshold=cfg.get("postprocess_hotword_threshold", 0.85),
        enable_fuzzy=cfg.get("postprocess_hotword_fuzzy", True))
    if matcher is None:
        return results
    want_matches = bool(cfg.get("return_postprocess_hotword_matches", False))
    for result in results:
        if isinstance(result, dict):
            matcher.apply_result(result, return_matches=want_matches)
    return results
