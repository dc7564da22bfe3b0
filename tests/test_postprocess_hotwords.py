"""Text-level hotword correction: the scenarios of the reference's tests/test_postprocess_hotwords.py replayed against
funasr_amd.postprocess_hotwords, and 300 cases produced by the reference module (oracle/make_golden_hotwords.py; the fuzzy
search runs with the same stand-in pinyin / ratio functions on both sides because pypinyin / rapidfuzz are not installed)."""
import difflib
import json
import os

import pytest

from funasr_amd import postprocess_hotwords as PH

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "postprocess_hotwords.json")


def _fake_pinyin(text, style=None, errors="ignore"):          # identical to oracle/make_golden_hotwords.py:fake_pinyin
    out = []
    for ch in text:
        if "一" <= ch <= "鿿":
            out.append("bpmfdtnl"[ord(ch) % 8] + "aoeiu"[ord(ch) % 5] + ("ng" if ord(ch) % 3 == 0 else ""))
        elif ch.isascii() and ch.isalnum():
            out.append(ch)
    return out


class _Style:
    NORMAL = 0


class _Fuzz:
    @staticmethod
    def ratio(a, b):
        return 100.0 * difflib.SequenceMatcher(None, a, b).ratio()


@pytest.fixture
def stand_ins(monkeypatch):
    monkeypatch.setattr(PH, "_pinyin", (_fake_pinyin, _Style))
    monkeypatch.setattr(PH, "_fuzz", _Fuzz)


def test_cases_from_the_reference_module(stand_ins):
    with open(GOLD, encoding="utf-8") as f:
        gold = json.load(f)
    assert len(gold["cases"]) == 300
    for c in gold["cases"]:
        m = PH.PostprocessHotwordMatcher(explicit_map=c["explicit"], fuzzy_targets=c["targets"], threshold=c["threshold"])
        out, matches = m.apply_text(c["text"])
        assert out == c["out"] and [x.as_dict() for x in matches] == c["matches"], (c, out)
    for src, explicit, fuzzy in gold["parse"]:
        assert list(PH.parse_postprocess_hotwords(src)) == [explicit, fuzzy], src


def test_parsing(tmp_path):
    assert PH.parse_postprocess_hotwords(["科大讯飞", "东方财富"]) == ({}, ["科大讯飞", "东方财富"])
    assert PH.parse_postprocess_hotwords({"科大迅飞": "科大讯飞", "东方财富": "东方财富"}) == ({"科大迅飞": "科大讯飞"}, ["东方财富"])
    assert PH.parse_postprocess_hotwords(["撒贝你=>撒贝宁", "康辉"]) == ({"撒贝你": "撒贝宁"}, ["康辉"])
    p = tmp_path / "hot.txt"
    p.write_text("# comment\n科大讯飞\n科大迅飞=>科大讯飞\n", encoding="utf-8")
    assert PH.parse_hotword_file(str(p)) == ({"科大迅飞": "科大讯飞"}, ["科大讯飞"])
    with pytest.raises(FileNotFoundError):
        PH.parse_hotword_file(str(tmp_path / "nope.txt"))
    with pytest.raises(TypeError):
        PH.parse_postprocess_hotwords(3)


def test_explicit_replacement_and_result_fields():
    m = PH.PostprocessHotwordMatcher(explicit_map={"撒贝你": "撒贝宁"}, enable_fuzzy=False)
    text, matches = m.apply_text("我非常喜欢撒贝你说的新闻")
    assert text == "我非常喜欢撒贝宁说的新闻" and len(matches) == 1 and matches[0].replacement == "撒贝宁" and matches[0].score == 1.0
    result = {"text": "撒贝你主持节目", "timestamp": [[0, 100], [100, 200]],
              "sentence_info": [{"text": "撒贝你主持", "sentence": "撒贝你主持", "start": 0, "end": 1000},
                                {"text": "节目", "sentence": "节目", "start": 1000, "end": 1500}]}
    m.apply_result(result, return_matches=True)
    assert result["text"] == "撒贝宁主持节目" and result["sentence_info"][0]["text"] == "撒贝宁主持"
    assert result["sentence_info"][0]["sentence"] == "撒贝宁主持" and result["timestamp"] == [[0, 100], [100, 200]]
    assert result["postprocess_hotword_matches"][0]["replacement"] == "撒贝宁"
    assert PH.HotwordMatch("a", "b", 0.9, 1, 2).as_dict() == {"original": "a", "replacement": "b", "score": 0.9, "start": 1, "end": 2}
    with pytest.raises(ValueError):
        PH.PostprocessHotwordMatcher(explicit_map={"a": "b"}, threshold=1.5)


def test_missing_fuzzy_dependency_raises(monkeypatch):
    monkeypatch.setattr(PH, "_pinyin", None)
    monkeypatch.setattr(PH, "_fuzz", None)
    try:
        import pypinyin  # noqa: F401
        import rapidfuzz  # noqa: F401
        pytest.skip("the optional dependencies are installed")
    except ImportError:
        pass
    with pytest.raises(ImportError):
        PH.PostprocessHotwordMatcher(fuzzy_targets=["科大讯飞"], threshold=0.85)


def test_one_matcher_for_all_results_and_the_no_op(monkeypatch):
    built = []
    original = PH.build_postprocess_hotword_matcher
    monkeypatch.setattr(PH, "build_postprocess_hotword_matcher", lambda *a, **k: built.append(1) or original(*a, **k))
    results = [{"text": "撒贝你主持", "timestamp": [1]}, {"text": "康灰播报", "timestamp": [2]}]
    cfg = {"postprocess_hotwords": {"撒贝你": "撒贝宁", "康灰": "康辉"}, "return_postprocess_hotword_matches": True}
    out = PH.apply_postprocess_hotwords_to_results(results, cfg)
    assert built == [1] and out[0]["text"] == "撒贝宁主持" and out[1]["text"] == "康辉播报"
    assert len(out[0]["postprocess_hotword_matches"]) == 1
    same = [{"text": "不变", "timestamp": [1]}]
    assert PH.apply_postprocess_hotwords_to_results(same, {}) is same and same[0]["text"] == "不变"


def test_generate_applies_the_correction_to_final_results():
    """AutoModel.generate ends with the text-level correction on both routes (auto_model.py:742,748)"""
    from funasr_amd.auto_model import AutoModel
    am = AutoModel.__new__(AutoModel)
    am.vad_model, am.punc_model, am.punc_kwargs = None, None, {}
    am.kwargs, am._base_kwargs = {}, {}
    am.inference = lambda *a, **k: [{"key": "k", "text": "我喜欢撒贝你", "timestamp": [[0, 1]]}]
    out = am.generate("x", postprocess_hotwords=["撒贝你=>撒贝宁"], postprocess_hotword_fuzzy=False)
    assert out[0]["text"] == "我喜欢撒贝宁" and out[0]["timestamp"] == [[0, 1]]
    # a punctuation model without VAD punctuates every result on its own (:732-741)
    calls = []

    def scripted(data, *a, model=None, **k):
        calls.append(data)
        return [{"key": "k", "text": "你好 世界"}] if model is None else [{"text": "你好，世界。", "punc_array": [2, 3]}]

    am.punc_model = object()
    am.inference = scripted
    out = am.generate("x", return_raw_text=True)
    assert out[0]["text"] == "你好，世界。" and out[0]["raw_text"] == "你好 世界" and calls == ["x", "你好 世界"]
