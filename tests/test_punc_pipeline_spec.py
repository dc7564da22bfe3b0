"""The reference's own regression scenarios for `AutoModel.inference_with_vad` with / without a punctuation model
(/root/reference/tests/test_punc_model_none.py: scripted VAD / ASR / punctuation results, expected texts and sentence
records) replayed against funasr_amd.auto_model.AutoModel. The scripts and the expected values are the reference test's;
the harness is ours (the reference patches its own module internals). The speaker scenario is out of scope (spk_model)."""
import numpy as np
import pytest

from funasr_amd.auto_model import AutoModel

TAG = "<|zh|><|NEUTRAL|><|Speech|><|woitn|>"


class _Punc:
    jieba_usr_dict = None
    punc_list = ["<unk>", "_", "，", "。", "？", "、"]

    def parameters(self):
        return iter(())


def _run(script, n_samples, punc=None, expect_punc_input=None, **cfg):
    """script: the results `AutoModel.inference` returns call by call (VAD, one ASR call per segment on "cpu", punctuation)"""
    am = AutoModel.__new__(AutoModel)
    am.model, am.vad_model, am.vad_kwargs = object(), object(), {}
    am.punc_model, am.punc_kwargs = punc, {}
    am.kwargs = dict(batch_size_s=300, batch_size_threshold_s=60, device="cpu", disable_pbar=True, fs=16000)
    am._base_kwargs = dict(am.kwargs)
    script = list(script)
    seen = []

    def scripted(data, *args, **kwargs):
        seen.append(data)
        if len(script) == 1 and expect_punc_input is not None:
            assert data == expect_punc_input
        return script.pop(0)

    am.inference = scripted
    out = am.inference_with_vad(np.zeros(n_samples, dtype=np.float32), key="test_utt", **cfg)
    assert not script, "not every scripted result was consumed"
    return out


def _vad(*segments):
    return [{"key": "test_utt", "value": [list(s) for s in segments]}]


def test_without_punctuation_model():
    asr = [{"text": "hello world", "timestamp": [[0, 500], [500, 1000]]}]
    out = _run([_vad([0, 16000]), asr], 16000 * 16)
    assert len(out) == 1 and out[0]["text"] == "hello world" and out[0]["key"] == "test_utt"
    asr = [{"text": "hello world", "timestamp": [[0, 500], [500, 1000]]}]
    out = _run([_vad([0, 16000]), asr], 16000 * 16, sentence_timestamp=True)
    assert out[0].get("sentence_info") == []                       # token timestamps but nothing to cut sentences with


def test_vad_segments_become_sentences_when_asr_has_no_timestamps():
    out = _run([_vad([120, 800], [1050, 1900]), [{"text": "first phrase"}], [{"text": "second phrase"}]], 32000,
               sentence_timestamp=True)
    assert out[0]["sentence_info"] == [
        {"start": 120, "end": 800, "text": "first phrase", "sentence": "first phrase", "timestamp": []},
        {"start": 1050, "end": 1900, "text": "second phrase", "sentence": "second phrase", "timestamp": []}]
    # ... but only when there is no punctuation model at all
    out = _run([_vad([120, 800]), [{"text": "phrase", "timestamps": []}]], 16000, punc=_Punc(), sentence_timestamp=True)
    assert out[0]["sentence_info"] == []


def test_punctuated_text_replaces_the_joined_text():
    script = [_vad([0, 16000]), [{"text": "hello world", "timestamp": [[0, 500], [500, 1000]]}],
              [{"text": "Hello, world.", "punc_array": [1, 2]}]]
    out = _run(script, 16000 * 16, punc=_Punc())
    assert len(out) == 1 and out[0]["text"] == "Hello, world."


def test_asr_words_keep_their_spelling_and_drive_the_sentence_records():
    script = [_vad([0, 2000]),
              [{"text": TAG + "你好世界", "timestamp": [[0, 500], [500, 1000], [1000, 1500], [1500, 2000]], "words": ["你", "好", "世", "界"]}],
              [{"text": "你好，世界。", "punc_array": [1, 2, 1, 3]}]]
    out = _run(script, 32000, punc=_Punc(), expect_punc_input="你好世界", sentence_timestamp=True, return_raw_text=True)
    assert out[0]["raw_text"] == TAG + "你好世界"
    assert out[0]["sentence_info"] == [
        {"text": "你好，", "start": 0, "end": 1000, "timestamp": [[0, 500], [500, 1000]], "raw_text": "你好"},
        {"text": "世界。", "start": 1000, "end": 2000, "timestamp": [[1000, 1500], [1500, 2000]], "raw_text": "世界"}]


def test_words_align_across_vad_segments():
    script = [_vad([0, 1000], [1000, 2000]),
              [{"text": TAG + "你好", "timestamp": [[0, 500], [500, 1000]], "words": ["你", "好"]}],
              [{"text": TAG + "世界", "timestamp": [[0, 500], [500, 1000]], "words": ["世", "界"]}],
              [{"text": "你好，世界。", "punc_array": [1, 2, 1, 3]}]]
    out = _run(script, 32000, punc=_Punc(), expect_punc_input="你好世界", sentence_timestamp=True)
    info = out[0]["sentence_info"]
    assert [s["text"] for s in info] == ["你好，", "世界。"]
    assert [s["timestamp"] for s in info] == [[[0, 500], [500, 1000]], [[1000, 1500], [1500, 2000]]]


def test_malformed_punctuation_array_falls_back_to_vad_segments():
    script = [_vad([100, 1100], [1300, 2300]),
              [{"text": TAG + "第一句", "timestamp": [[0, 300], [300, 600], [600, 900]], "words": ["第", "一", "句"]}],
              [{"text": TAG + "第二句", "timestamp": [[0, 300], [300, 600], [600, 900]], "words": ["第", "二", "句"]}],
              [{"text": "第一句。第二句。", "punc_array": [3]}]]
    out = _run(script, 40000, punc=_Punc(), sentence_timestamp=True)
    assert out[0]["sentence_info"] == [
        {"start": 100, "end": 1000, "text": "第一句", "sentence": "第一句", "timestamp": [[100, 400], [400, 700], [700, 1000]]},
        {"start": 1300, "end": 2200, "text": "第二句", "sentence": "第二句", "timestamp": [[1300, 1600], [1600, 1900], [1900, 2200]]}]


def test_one_asr_word_split_across_punctuation_tokens(monkeypatch):
    from funasr_amd import punc_align
    monkeypatch.setattr(punc_align, "punc_tokens", lambda text, punc_array, punc_model: ["aon", "storyi", "ca", "说", "差", "距"])
    punc = _Punc()
    punc.punc_list = None
    script = [_vad([0, 1200]),
              [{"text": TAG + "aon storyica说差距", "timestamp": [[0, 100], [100, 900], [900, 1000], [1000, 1100], [1100, 1200]],
                "words": ["aon", "storyica", "说", "差", "距"]}],
              [{"text": "aon storyica.说差距。", "punc_array": [1, 1, 3, 1, 1, 3]}]]
    out = _run(script, 19200, punc=punc, sentence_timestamp=True)
    assert out[0]["sentence_info"] == [
        {"text": "aon storyica.", "start": 0, "end": 900, "timestamp": [[0, 100], [100, 700], [700, 900]]},
        {"text": "说差距。", "start": 900, "end": 1200, "timestamp": [[900, 1000], [1000, 1100], [1100, 1200]]}]


@pytest.mark.parametrize("en_post_proc", [False, True])
def test_english_surface_text_is_preserved(en_post_proc):
    surface = "don't stop https://nature.com email@example.com"
    words = ["don", "'", "t", "stop", "https", ":", "/", "/", "nature", ".", "com", "email", "@", "example", ".", "com"]
    tag = "<|en|><|NEUTRAL|><|Speech|><|woitn|>"
    script = [_vad([0, 1600]),
              [{"text": tag + surface, "timestamp": [[i * 100, (i + 1) * 100] for i in range(len(words))], "words": words}],
              [{"text": " Don ' t stop. Https : / / nature .com. Email @ example .com.", "punc_array": [1, 3, 1, 3]}]]
    out = _run(script, 25600, punc=_Punc(), expect_punc_input=surface, sentence_timestamp=True, en_post_proc=en_post_proc)
    assert out[0]["text"] == "don't stop. https://nature.com email@example.com."
    info = out[0]["sentence_info"]
    assert [(s["start"], s["end"]) for s in info] == [(0, 400), (400, 1600)]
    assert [s["text"] for s in info] == ["don't stop.", "https://nature.com email@example.com."]
    assert info[-1]["timestamp"][-1] == [1100, 1600]


def test_malformed_word_or_punctuation_metadata_uses_the_plain_path():
    script = [_vad([0, 1000]), [{"text": "你 好", "timestamp": [[0, 500], [500, 1000]], "words": ["你", ""]}],
              [{"text": "你好。", "punc_array": [1, 3]}]]
    out = _run(script, 16000, punc=_Punc(), sentence_timestamp=True)
    assert out[0]["sentence_info"][0]["text"] == "你好。"
    script = [_vad([0, 1000]), [{"text": "你 好", "timestamp": [[0, 500], [500, 1000]], "words": ["你", "好"]}],
              [{"text": "你好。", "punc_array": 3}]]
    out = _run(script, 16000, punc=_Punc(), sentence_timestamp=True)
    assert out[0]["sentence_info"][0]["text"] == ["你", "好"]


def test_forced_final_period_rewrites_the_last_punctuation_id():
    """TestCTTransformerPunctuation.test_forced_period_uses_sentence_end_id: whatever the network says about the last word,
    the text ends on a sentence end and punc_array[-1] is the sentence-end id"""
    from funasr_amd.ct_transformer import assemble, split_words
    marks = ["<unk>", "_", "，", "。", "？", "、"]
    for punc_id in (1, 2, 5):
        for text in ("hello world", "你好"):
            tokens = split_words(text)
            out, ids = assemble(tokens, np.arange(len(tokens)), lambda x: np.full(len(x), punc_id), marks, 3, split_size=20)
            assert out.endswith((".", "。")) and int(ids[-1]) == 3, (punc_id, text, out, ids)
