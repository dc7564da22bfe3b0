"""rich_transcription_postprocess / sentence_postprocess_sentencepiece against cases produced by the reference functions
(oracle/make_golden_postprocess.py)."""
import json
import os

from funasr_amd.postprocess_utils import (rich_transcription_postprocess, sentence_postprocess,
                                          sentence_postprocess_sentencepiece)

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "postprocess.json")


def test_rich_transcription_postprocess_equals_reference():
    with open(GOLD, encoding="utf-8") as f:
        cases = json.load(f)["rich"]
    assert len(cases) >= 80
    for s, want in cases:
        assert rich_transcription_postprocess(s) == want, (s, rich_transcription_postprocess(s), want)
    assert rich_transcription_postprocess("<|en|><|HAPPY|><|Applause|><|woitn|>well done") == "👏well done😊"


def test_sentencepiece_postprocess_equals_reference():
    with open(GOLD, encoding="utf-8") as f:
        cases = json.load(f)["sentencepiece"]
    for words, (sentence, word_list) in cases:
        got = sentence_postprocess_sentencepiece(words)
        assert got[0] == sentence and got[1] == word_list, (words, got, sentence, word_list)
    assert sentence_postprocess(["你", "好"])[0] == "你好"
