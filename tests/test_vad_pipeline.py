"""The ASR half of AutoModel.inference_with_vad (funasr/auto/auto_model.py:852-1254): VAD segments -> length-sorted
dynamic batches -> decode -> restore -> merge. CPU tests with stand-in VAD / ASR models that follow the FunASR model
contract; the batching policy is checked against a line-by-line transliteration of the reference loop (:946-966)."""
import json
import os

import pytest
import torch

from funasr_amd.auto_model import AutoModel
from funasr_amd.vad_utils import merge_vad, vad_segment_sentences

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "timestamps.json")


def test_merge_vad_equals_reference():
    with open(GOLD, encoding="utf-8") as f:
        cases = json.load(f)["merge_vad"]
    assert len(cases) == 24
    for c in cases:
        assert merge_vad([list(x) for x in c["segments"]], c["max_length"], c["min_length"]) == c["merged"]


def _reference_loop(durs, batch_size, threshold):
    """auto_model.py:946-966 with its own variable names; returns the [beg_idx, end_idx) of every self.inference call"""
    n = len(durs)
    calls = []
    beg_idx, end_idx, max_len_in_batch = 0, 1, 0
    for j in range(n):
        sample_length = durs[j]
        potential_batch_length = max(max_len_in_batch, sample_length) * (j + 1 - beg_idx)
        if j < n - 1 and sample_length < threshold and potential_batch_length < batch_size:
            max_len_in_batch = max(max_len_in_batch, sample_length)
            end_idx += 1
            continue
        calls.append((beg_idx, end_idx))
        beg_idx = end_idx
        end_idx += 1
        max_len_in_batch = sample_length
    return calls


def test_batch_plan_equals_reference_loop():
    g = torch.Generator().manual_seed(0)
    for trial in range(200):
        n = int(torch.randint(1, 40, (1,), generator=g))
        durs = sorted(int(x) for x in torch.randint(200, 70000, (n,), generator=g))
        bs = int(torch.randint(1, 400, (1,), generator=g)) * 1000
        thr = int(torch.randint(5, 80, (1,), generator=g)) * 1000
        plan = AutoModel.plan_vad_batches(durs, bs, thr)
        assert plan == _reference_loop(durs, bs, thr)
        covered = [i for b, e in plan for i in range(b, e)]
        assert covered == list(range(n))                       # every segment decoded exactly once, in sorted order


class _FakeVAD:
    """returns the i-th entry of `table` for the i-th recording it is asked about (tensor inputs get random keys,
    auto_model.py:398-413)"""
    def __init__(self, table):
        self.table, self.n = list(table), 0

    def parameters(self):
        return iter(())

    def inference(self, data_in, key=None, **kwargs):
        segs = self.table[self.n]
        self.n += 1
        return [{"key": key[0], "value": [list(s) for s in segs]}], {"batch_data_time": 1.0}


class _FakeASR:
    """'decodes' a clip into one token per 100 ms, named after the clip's first sample (the sample index is planted
    in the waveform), with timestamps relative to the clip start -- like Paraformer.inference(pred_timestamp=True)"""
    def __init__(self):
        self.calls = []

    def parameters(self):
        return iter(())

    def inference(self, data_in, key=None, **kwargs):
        self.calls.append([int(c.shape[0]) for c in data_in])
        res = []
        for k, c in zip(key, data_in):
            ms = c.shape[0] // 16
            n_tok = max(ms // 100, 1)
            start = int(round(float(c[0])))
            if start == 77_000 * 16:                                              # one segment decodes to nothing
                res.append({"key": k, "text": "", "timestamp": []})
                continue
            res.append({"key": k, "text": " ".join(f"t{start // 16}" for _ in range(n_tok)),
                        "timestamp": [[100 * i, 100 * i + 100] for i in range(n_tok)]})
        return res, {"batch_data_time": sum(c.shape[0] for c in data_in) / 16000.0}


def _auto(vad, asr, **kw):
    m = AutoModel.__new__(AutoModel)
    m.model, m.vad_model, m.vad_kwargs = asr, vad, {}
    m.punc_model, m.punc_kwargs = None, {}
    m.kwargs = dict(device="cuda", batch_size=1, **kw)
    m._base_kwargs = dict(m.kwargs)
    return m


def test_inference_with_vad_sorts_batches_restores_and_merges():
    n = 100 * 16000
    wav = torch.arange(n, dtype=torch.float32)                                   # sample value == sample index
    segs = {"rec0": [[1000, 4000], [5000, 5600], [7000, 20000], [21000, 21900], [30000, 33000], [77000, 77500]],
            "rec1": []}
    asr = _FakeASR()
    am = _auto(_FakeVAD([segs["rec0"], segs["rec1"]]), asr, batch_size_s=10, batch_size_threshold_s=8)
    out = am.generate([wav, wav[:16000]], sentence_timestamp=True)
    assert len(out) == 2 and out[0]["key"].startswith("rand_key_")
    assert out[1] == {"key": out[1]["key"], "text": "", "timestamp": []}
    # batches: durations sorted ascending 500, 600, 900, 3000, 3000, 13000 ms
    durs = sorted(e - b for b, e in segs["rec0"])
    plan = _reference_loop(durs, max(10_000, durs[0]), 8_000)
    assert [len(c) for c in asr.calls] == [e - b for b, e in plan]
    assert sorted(x for c in asr.calls for x in c) == sorted(d * 16 for d in durs)
    # text in recording order (the empty segment contributes an empty string), timestamps shifted by the segment start
    words = out[0]["text"].split()
    expect = []
    for b, e in segs["rec0"]:
        if b != 77000:
            expect += [f"t{b}"] * max((e - b) // 100, 1)
    assert words == expect
    ts = out[0]["timestamp"]
    assert ts[0] == [1000, 1100] and ts[-1][1] <= 33000 and all(ts[i][0] <= ts[i + 1][0] for i in range(len(ts) - 1))
    assert len(ts) == len(expect)
    # token timestamps but no punctuation model: the reference has nothing to cut sentences with (auto_model.py:1204-1208)
    assert out[0]["sentence_info"] == []


def test_inference_with_vad_row_budget_batches_give_the_same_result():
    """`batch_size_rows` (an encoder-row budget per batch, funasr_amd.dp.plan_batches_by_rows) only regroups the sorted
    segments: same text / timestamps as the reference's `batch_size_s` policy, batches within the budget"""
    n = 100 * 16000
    wav = torch.arange(n, dtype=torch.float32)
    segs = [[1000, 4000], [5000, 5600], [7000, 20000], [21000, 21900], [30000, 33000], [40000, 41000], [50000, 58000]]
    ref = _auto(_FakeVAD([segs]), _FakeASR(), batch_size_s=10, batch_size_threshold_s=8).generate([wav])
    asr = _FakeASR()
    out = _auto(_FakeVAD([segs]), asr, batch_size_s=10, batch_size_threshold_s=8, batch_size_rows=256).generate([wav])
    assert out[0]["text"] == ref[0]["text"] and out[0]["timestamp"] == ref[0]["timestamp"]
    frames = lambda samples: max(1, (samples // 16) // 60)        # the fake frontend-less estimate of auto_model: 60 ms per row
    for call in asr.calls:                                        # padded layout (the fake model has no f16x2 encoder)
        rows = len(call) * ((max(frames(c) for c in call) + 15) // 16 * 16)
        assert rows <= 256 or len(call) == 1, call
    assert sum(len(c) for c in asr.calls) == len(segs) and len(asr.calls) > 1


class _FakePunc:
    """marks a comma after every 3rd word and a period after every 7th; one punc id per word, like CTTransformer"""
    def parameters(self):
        return iter(())

    def inference(self, data_in, key=None, **kwargs):
        words = data_in[0].split()
        ids = [3 if (i + 1) % 7 == 0 else 2 if (i + 1) % 3 == 0 else 1 for i in range(len(words))]
        ids[-1] = 3
        text = "".join(w + {1: " ", 2: ", ", 3: ". "}[p] for w, p in zip(words, ids)).strip()
        return [{"key": key[0], "text": text, "punc_array": torch.tensor(ids)}], {}


def test_punctuation_branch_and_sentence_records():
    from funasr_amd.timestamps import timestamp_sentence
    wav = torch.arange(40 * 16000, dtype=torch.float32)
    segs = [[1000, 2000], [5000, 6500], [20000, 20400]]
    asr = _FakeASR()
    am = _auto(_FakeVAD([segs]), asr, batch_size_s=300)
    am.punc_model, am.punc_kwargs = _FakePunc(), {}
    out = am.generate(wav, sentence_timestamp=True, return_raw_text=True, en_post_proc=True)[0]
    words = [f"t{b}" for b, e in segs for _ in range(max((e - b) // 100, 1))]
    assert out["raw_text"] == " ".join(words)
    assert out["text"].endswith(".") and out["text"].count(",") + out["text"].count(".") == sum(1 for i in range(len(words)) if (i + 1) % 3 == 0 or (i + 1) % 7 == 0 or i == len(words) - 1)
    ids = [3 if (i + 1) % 7 == 0 else 2 if (i + 1) % 3 == 0 else 1 for i in range(len(words))]
    ids[-1] = 3
    assert out["sentence_info"] == timestamp_sentence(ids, out["timestamp"], " ".join(words), return_raw_text=True, english=True)
    assert out["sentence_info"][0]["start"] == 1000 and out["sentence_info"][-1]["end"] == out["timestamp"][-1][1]
    # an ASR model without timestamps: sentence records fall back to the VAD segments when there is no punc model either
    class NoStampASR(_FakeASR):
        def inference(self, data_in, key=None, **kwargs):
            res, meta = super().inference(data_in, key=key, **kwargs)
            return [{k: v for k, v in r.items() if k != "timestamp"} for r in res], meta
    am2 = _auto(_FakeVAD([segs]), NoStampASR(), batch_size_s=300)
    info = am2.generate(wav, sentence_timestamp=True)[0]["sentence_info"]
    assert [(s["start"], s["end"]) for s in info] == [tuple(s) for s in segs]


def test_merge_vad_option_and_missing_vad_model():
    n = 30 * 16000
    wav = torch.arange(n, dtype=torch.float32)
    segs = {"r": [[0, 2000], [2500, 4000], [9000, 12000], [12500, 29000]]}
    asr = _FakeASR()
    am = _auto(_FakeVAD([segs["r"]]), asr, batch_size_s=300, merge_length_s=5)
    out = am.generate(wav, key="r", merge_vad=True)
    merged = merge_vad([list(s) for s in segs["r"]], 5000)
    assert sorted(x for c in asr.calls for x in c) == sorted((e - b) * 16 for b, e in merged)
    assert out[0]["key"] == "r" and out[0]["text"]
    with pytest.raises(FileNotFoundError):
        AutoModel(model="Paraformer", vad_model="fsmn-vad")             # not a local directory, not a registered class
    assert vad_segment_sentences([{"text": "<|zh|>"}], [[0, 10]]) == []
