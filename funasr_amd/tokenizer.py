"""Host-side id -> text glue (no GPU work).

`CharTokenizer` mirrors the inference half of funasr/tokenizer/char_tokenizer.py:12-99 + BaseTokenizer.ids2tokens
(funasr/tokenizer/abs_tokenizer.py:108-116): a token list (tokens.json / list / one-token-per-line file), ids2tokens,
tokens2text. `sentence_postprocess` restates the no-timestamp path of funasr/utils/postprocess_utils.py:165-278:
drop <s>/</s>/<unk>/<OOV>, join CJK characters directly, re-assemble "@@"-continued BPE pieces, put blanks between
alphabetic words, merge spelled-out single letters into upper-case abbreviations (abbr_dispose, :57-163).
"""
from __future__ import annotations

import json
import os
from typing import Iterable, List, Optional, Sequence, Tuple, Union

from .register import tables

_DROP = ("<s>", "</s>", "<unk>", "<OOV>")


def _strip(tok: str) -> str:
    if " " not in tok and "<" not in tok:          # nothing to remove (every dropped symbol has a "<"): the common case
        return tok
    for d in (" ",) + _DROP:
        tok = tok.replace(d, "")
    return tok


def _is_cjk_token(tok: str) -> bool:
    return "一" <= tok <= "鿿" or "0" <= tok <= "9" or tok == "@"


def _all_cjk(tokens: Sequence[str]) -> bool:
    toks = [_strip(t) for t in tokens]
    return len(toks) > 0 and all(_is_cjk_token(t) for t in toks)


def _all_alpha(tokens: Sequence[str]) -> bool:
    toks = [_strip(t) for t in tokens]
    if not toks:
        return False
    for t in toks:
        if not t.isalpha() and t != "'":
            return False
        if t.isalpha() and _is_cjk_token(t):
            return False
    return True


def _is_letter(tok: str) -> bool:
    # one ASCII letter (the reference tests `len(word) == 1 and word.encode("utf-8").isalpha()`: bytes.isalpha is ASCII-only)
    return len(tok) == 1 and ("a" <= tok <= "z" or "A" <= tok <= "Z")


def _merge_abbreviations(words: List[str], spans: Optional[List[List[int]]] = None):
    """'a', ' ', 'b', ' ', 'c' -> 'ABC' (abbr_dispose, postprocess_utils.py:68-163). With `spans` (one [begin, end] per
    non-blank word) the merged word runs from its first letter's begin to its last letter's end and the merged span
    list is returned as well."""
    if spans is None and not any(len(w) == 1 and ("a" <= w <= "z" or "A" <= w <= "Z") for w in words):
        return list(words)                         # no single letters: nothing to merge
    out: List[str] = []
    out_spans: List[List[int]] = []
    # span index of every position: a blank shares the index of the word that follows it
    idx, k = [], 0
    for w in words:
        idx.append(k)
        if w != " ":
            k += 1
    i, n = 0, len(words)
    while i < n:
        if _is_letter(words[i]) and i + 2 < n and words[i + 1] == " " and _is_letter(words[i + 2]):
            j = i + 2
            while j + 2 < n and words[j + 1] == " " and _is_letter(words[j + 2]):
                j += 2
            out.append("".join(w.upper() for w in words[i:j + 1] if w != " "))
            if spans is not None:
                begin = spans[idx[i]][0]        # IndexError, like the reference (:130), when tokens without a span
                if idx[j] < len(spans):         # (neither CJK, alphabetic nor a piece) pushed the words past the spans
                    out_spans.append([begin, spans[idx[j]][1]])
            i = j + 1
        else:
            out.append(words[i])
            if spans is not None and words[i] != " " and idx[i] < len(spans):
                out_spans.append([spans[idx[i]][0], spans[idx[i]][1]])
            i += 1
    return (out, out_spans) if spans is not None else out


def sentence_postprocess(tokens: Iterable[Union[str, bytes]], time_stamp: Optional[List[List[int]]] = None):
    """-> (sentence, words) or, with one [begin_ms, end_ms] per token, (sentence, spans, words): the spans of "@@"
    continued pieces are merged into their word, words are joined by blanks (postprocess_utils.py:165-278)."""
    toks = [t if isinstance(t, str) else t.decode("utf-8") for t in tokens]
    toks = [t for t in toks if t not in _DROP]
    ts = time_stamp
    words: List[str] = []
    spans: List[List[int]] = []
    if _all_cjk(toks):
        words = [t.replace(" ", "") for t in toks]
        if ts is not None:
            spans = ts
    elif _all_alpha(toks):
        piece, begin = "", None
        for i, t in enumerate(toks):
            if ts is not None and begin is None:
                begin = ts[i][0]
            if "@@" in t:
                piece += t.replace("@@", "")
            else:
                words += [piece + t, " "]
                piece = ""
                if ts is not None:
                    spans.append([begin, ts[i][1]])
                    begin = None
    else:
        # mixed script. `fresh`: the next token starts a new span (False while "@@" pieces are pending); a token that
        # is neither CJK, a piece nor alphabetic joins the words but gets no span -- as in the reference
        piece, blank, fresh, begin, end = "", False, True, -1, -1
        for i, t in enumerate(toks):
            if ts is not None and fresh:
                begin, end = ts[i][0], ts[i][1]
            if _all_cjk(t):
                if blank:
                    words.pop()
                words.append(t)
                blank = False
                if ts is not None:
                    spans.append([begin, end])
                    fresh, begin = True, end
            elif "@@" in t:
                piece += t.replace("@@", "")
                blank = False
                if ts is not None:
                    fresh, end = False, ts[i][1]
            elif _all_alpha(t):
                words += [piece + t, " "]
                piece, blank = "", True
                if ts is not None:
                    end = ts[i][1]
                    spans.append([begin, end])
                    fresh, begin = True, end
            else:
                words.append(t)
    if ts is None:
        words = _merge_abbreviations(words)
        return "".join(words).strip(), [w for w in words if w != " "]
    words, spans = _merge_abbreviations(words, spans)
    real = [w for w in words if w != " "]
    return " ".join(real).strip(), spans, real


@tables.register("tokenizer_classes", "CharTokenizer")
class CharTokenizer:
    def __init__(self, token_list=None, unk_symbol: str = "<unk>", space_symbol: str = "<space>", **kwargs):
        if isinstance(token_list, (str, os.PathLike)):
            path = str(token_list)
            with open(path, "r", encoding="utf-8") as f:
                self.token_list = json.load(f) if path.endswith(".json") else [ln.rstrip("\n").split()[0] for ln in f if ln.strip()]
        elif token_list is not None:
            self.token_list = list(token_list)
        else:
            self.token_list = None
        self.unk_symbol, self.space_symbol = unk_symbol, space_symbol
        if self.token_list is not None:
            self.token2id = {t: i for i, t in enumerate(self.token_list)}

    def get_num_vocabulary_size(self) -> int:
        return len(self.token_list)

    def ids2tokens(self, integers) -> List[str]:
        return [self.token_list[int(i)] for i in integers]

    def tokens2ids(self, tokens) -> List[int]:
        unk = self.token2id.get(self.unk_symbol, 0)
        return [self.token2id.get(t, unk) for t in tokens]

    def tokens2text(self, tokens: Iterable[str]) -> str:
        return "".join(t if t != self.space_symbol else " " for t in tokens)

    def text2tokens(self, line) -> List[str]:
        """char_tokenizer.py:72-94 without seg_dict / non-linguistic symbols: a string is cut into characters (blanks
        dropped); a LIST of words (what the punctuation model passes) stays a list of words"""
        return [t for t in line if t != " "]

    def encode(self, text, **kwargs) -> List[int]:
        return self.tokens2ids(self.text2tokens(text))

    def decode(self, ids) -> str:
        return self.tokens2text(self.ids2tokens(ids))


@tables.register("tokenizer_classes", "SentencepiecesTokenizer")
class SentencepiecesTokenizer:
    """SenseVoiceSmall's tokenizer (funasr/tokenizer/sentencepiece_tokenizer.py:11-104): a thin wrapper over the
    `sentencepiece` package around the model directory's `*.bpe.model`; host-side only."""

    def __init__(self, bpemodel, **kwargs):
        import sentencepiece as spm

        self.bpemodel = str(bpemodel)
        self.sp = spm.SentencePieceProcessor()
        self.sp.load(self.bpemodel)

    def text2tokens(self, line: str) -> List[str]:
        return self.sp.EncodeAsPieces(line)

    def tokens2text(self, tokens: Iterable[str]) -> str:
        return self.sp.DecodePieces(list(tokens))

    def encode(self, line, **kwargs):
        """a string -> ids; a LIST of strings -> one id list per string (what `tokens2ids` relies on)"""
        return self.sp.EncodeAsIds(line)

    def decode(self, ids, **kwargs) -> str:
        return self.sp.DecodeIds([int(i) for i in ids])

    def get_vocab_size(self) -> int:
        return self.sp.GetPieceSize()

    # the reference aliases these two to decode / encode (sentencepiece_tokenizer.py:84-100): ids2tokens returns TEXT and
    # tokens2ids re-encodes every piece string, i.e. returns a list of id lists -- SenseVoice's timestamp branch depends on it
    def ids2tokens(self, *args, **kwargs):
        return self.decode(*args, **kwargs)

    def tokens2ids(self, *args, **kwargs):
        return self.encode(*args, **kwargs)
