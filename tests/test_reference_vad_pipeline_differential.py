"""Differential test of the VAD-segmented pipeline: this package's AutoModel.generate with a VAD model against the
REFERENCE's own AutoModel.generate / inference_with_vad (funasr/auto/auto_model.py:852-1254, imported from /root/reference) --
the same stand-in VAD / ASR / punctuation models on both sides (deterministic, FunASR model contract), random recordings,
segment tables, batch budgets and option sets. Compared: the returned records (text, token timestamps shifted by the segment
starts, raw_text, sentence_info) and the composition of every ASR call (which segments were batched together, in which
order). Build container only (the GPU box has no /root/reference); tests/test_vad_pipeline.py checks the same logic against
a transliteration of the reference loop everywhere."""
import copy

import pytest
import torch

from oracle import ref_import
from tests.test_vad_pipeline import _FakeASR, _FakePunc, _FakeVAD

pytestmark = pytest.mark.skipif(not ref_import.available(), reason="reference checkout not present (GPU box)")


class _NoStampASR(_FakeASR):
    def inference(self, data_in, key=None, **kwargs):
        res, meta = super().inference(data_in, key=key, **kwargs)
        return [{k: v for k, v in r.items() if k != "timestamp"} for r in res], meta


class _NothingDecodedASR(_FakeASR):
    """like Paraformer.inference when no clip of a batch predicts a token: a BARE empty list (model.py:615-616), which
    AutoModel.inference turns into one {"text": ""} record for the whole batch (auto_model.py:815-817)"""
    def inference(self, data_in, key=None, **kwargs):
        if all(int(round(float(c[0]))) // 16 % 300 == 0 for c in data_in):
            self.calls.append([int(c.shape[0]) for c in data_in])
            return []
        return super().inference(data_in, key=key, **kwargs)


def _ours(vad, asr, punc, **kw):
    from funasr_amd.auto_model import AutoModel
    m = AutoModel.__new__(AutoModel)
    m.model, m.vad_model, m.vad_kwargs = asr, vad, {}
    m.punc_model, m.punc_kwargs = punc, {}
    m.kwargs = dict(dict(device="cuda", batch_size=1), **kw)
    m._base_kwargs = dict(m.kwargs)
    return m


def _theirs(RefAutoModel, vad, asr, punc, **kw):
    for fake in (vad, asr, punc):                   # the reference asks every model for its device (auto_model.py:846)
        if fake is not None:
            fake.parameters = lambda: iter([torch.zeros(1)])
    m = RefAutoModel.__new__(RefAutoModel)
    m.model, m.vad_model, m.vad_kwargs = asr, vad, {}
    m.punc_model, m.punc_kwargs = punc, {}
    m.spk_model, m.spk_mode, m.cb_model = None, "punc_segment", None
    m.kwargs = dict(dict(device="cuda", batch_size=1, frontend=None, disable_pbar=True), **kw)
    m._store_base_configs()
    return m


def _strip(results):
    out = []
    for r in results:
        r = {k: (v.tolist() if isinstance(v, torch.Tensor) else v) for k, v in r.items() if k != "key"}
        out.append(r)
    return out


@pytest.fixture(scope="module")
def RefAutoModel():
    AutoModel, _ = ref_import.reference_automodel()
    return AutoModel


def test_generate_with_vad_equals_the_reference_on_random_recordings(RefAutoModel, monkeypatch):
    monkeypatch.setattr(torch.cuda, "empty_cache", lambda: None)
    g = torch.Generator().manual_seed(2025)
    checked = with_punc = sentence = 0
    for trial in range(240):
        n_rec = int(torch.randint(1, 3, (1,), generator=g))
        wavs, tables = [], []
        for _ in range(n_rec):
            secs = int(torch.randint(5, 120, (1,), generator=g))
            wavs.append(torch.arange(secs * 16000, dtype=torch.float32))       # sample value == sample index (the fake ASR reads it)
            n_seg = int(torch.randint(0, 14, (1,), generator=g))
            cuts = sorted(set(int(v) * 100 for v in torch.randint(0, secs * 10, (2 * n_seg,), generator=g)))
            segs = [[cuts[i], cuts[i + 1]] for i in range(0, len(cuts) - 1, 2) if cuts[i + 1] > cuts[i]]
            tables.append(segs)
        opts = dict(batch_size_s=int(torch.randint(1, 40, (1,), generator=g)), batch_size_threshold_s=int(torch.randint(1, 30, (1,), generator=g)))
        call = {}
        if trial % 3 == 0:
            call["sentence_timestamp"] = True
        if trial % 4 == 1:
            call["return_raw_text"] = True
        if trial % 5 == 2:
            call.update(merge_vad=True)
            opts["merge_length_s"] = int(torch.randint(2, 20, (1,), generator=g))
        if trial % 7 == 3:
            call["en_post_proc"] = True
        if trial % 9 == 4:
            opts["device"] = "cpu"                       # the reference then decodes segment by segment (budget 0, :927-928)
        use_punc = trial % 2 == 0
        asr_cls = _NoStampASR if trial % 6 == 5 else _NothingDecodedASR if trial % 8 == 7 else _FakeASR
        a_asr, b_asr = asr_cls(), asr_cls()
        ours = _ours(_FakeVAD(copy.deepcopy(tables)), a_asr, _FakePunc() if use_punc else None, **opts)
        theirs = _theirs(RefAutoModel, _FakeVAD(copy.deepcopy(tables)), b_asr, _FakePunc() if use_punc else None, **opts)
        inp = wavs if n_rec > 1 else wavs[0]
        got = ours.generate([w.clone() for w in wavs] if n_rec > 1 else inp.clone(), **call)
        want = theirs.generate([w.clone() for w in wavs] if n_rec > 1 else inp.clone(), **call)
        assert a_asr.calls == b_asr.calls, (trial, opts, call, a_asr.calls, b_asr.calls)
        assert _strip(got) == _strip(want), (trial, opts, call, tables)
        checked += len(want)
        with_punc += int(use_punc)
        sentence += int("sentence_timestamp" in call)
    assert checked > 200 and with_punc > 100 and sentence > 60


class _EchoASR:
    """one record per clip: text names the clip by its first sample and length; remembers the batches and the runtime
    options it was called with"""
    def __init__(self):
        self.calls, self.seen = [], []

    def parameters(self):
        return iter([torch.zeros(1)])

    def inference(self, data_in, key=None, **kwargs):
        clips = [torch.as_tensor(c).reshape(-1) for c in data_in]
        self.calls.append([int(c.shape[0]) for c in clips])
        self.seen.append({k: kwargs.get(k) for k in ("hotword", "language", "batch_size", "use_itn")})
        res = [{"key": k, "text": " ".join(f"w{int(round(float(c[0])))}" for _ in range(max(c.shape[0] // 4000, 1)))} for k, c in zip(key, clips)]
        return res, {"batch_data_time": sum(c.shape[0] for c in clips) / 16000.0}


def test_generate_without_vad_equals_the_reference(RefAutoModel, monkeypatch):
    """AutoModel.generate / inference without a VAD model (auto_model.py:688-850): input forms (one tensor, lists of tensors /
    numpy arrays), keys, static batch_size, runtime options handed to the model, per-result punctuation with raw_text"""
    import numpy as np
    monkeypatch.setattr(torch.cuda, "empty_cache", lambda: None)
    g = torch.Generator().manual_seed(7)
    for trial in range(120):
        n = int(torch.randint(1, 9, (1,), generator=g))
        clips = [torch.arange(int(torch.randint(1, 20, (1,), generator=g)) * 4000, dtype=torch.float32) + 100 * i for i in range(n)]
        form = trial % 4
        if form == 0:
            make = lambda: clips[0].clone()
            n_in = 1
        elif form == 1:
            make = lambda: [c.clone() for c in clips]
            n_in = n
        elif form == 2:
            make = lambda: [c.numpy().copy() for c in clips]
            n_in = n
        else:
            make = lambda: clips[0].numpy().copy()
            n_in = 1
        opts = dict(batch_size=int(torch.randint(1, 5, (1,), generator=g)))
        call = {}
        use_punc = trial % 2 == 1
        if trial % 3 == 0 and not use_punc:
            # (with a punctuation model the reference merges `key` into punc_kwargs, :734, and then passes it twice to the punc
            # model's inference: TypeError -- this package's generate() accepts the combination)
            call["key"] = [f"utt{j}" for j in range(n_in)] if n_in > 1 else "solo"
        if trial % 5 == 1:
            call.update(hotword="魔搭", language="zh")
        if trial % 7 == 2:
            call["batch_size"] = int(torch.randint(1, 4, (1,), generator=g))      # a runtime override of the static batch size
        if trial % 4 == 1:
            call["return_raw_text"] = True
        a_asr, b_asr = _EchoASR(), _EchoASR()
        ours = _ours(None, a_asr, _FakePunc() if use_punc else None, **opts)
        theirs = _theirs(RefAutoModel, None, b_asr, _FakePunc() if use_punc else None, **opts)
        got, want = ours.generate(make(), **call), theirs.generate(make(), **call)
        assert a_asr.calls == b_asr.calls and a_asr.seen == b_asr.seen, (trial, opts, call, a_asr.calls, b_asr.calls, a_asr.seen, b_asr.seen)
        if "key" in call:
            assert [r["key"] for r in got] == [r["key"] for r in want], (trial, call)
        else:
            assert all(r["key"].startswith("rand_key_") for r in got) and all(r["key"].startswith("rand_key_") for r in want)
        assert _strip(got) == _strip(want), (trial, opts, call)
        # a second call on the same objects: runtime options of the first call must not stick (auto_model.py:1318-1359)
        got2, want2 = ours.generate(make()), theirs.generate(make())
        assert a_asr.seen[-1] == b_asr.seen[-1] and a_asr.calls == b_asr.calls, (trial, a_asr.seen[-1], b_asr.seen[-1])
        assert _strip(got2) == _strip(want2)


def test_prepare_data_iterator_equals_the_reference(RefAutoModel, tmp_path):
    """funasr/auto/auto_model.py:347-415 on the input forms of the hot path: key lists and data lists equal (random keys
    compared by their pattern of repetition)"""
    import sys
    import numpy as np
    ref_prepare = sys.modules[RefAutoModel.__module__].prepare_data_iterator
    from funasr_amd.auto_model import prepare_data_iterator
    wavs = []
    for i in range(3):
        p = tmp_path / f"clip{i}.wav"
        p.write_bytes(b"RIFF")
        wavs.append(str(p))
    scp = tmp_path / "wav.scp"
    scp.write_text("".join(f"id{i} {w}\n" for i, w in enumerate(wavs)) + f"{wavs[0]}\n", encoding="utf-8")
    jsonl = tmp_path / "list.jsonl"
    jsonl.write_text('{"source": "a.wav", "key": "k0"}\n{"source": "b.wav"}\n', encoding="utf-8")
    arr = np.zeros(100, dtype=np.float32)
    cases = [(wavs[0], {}), (wavs[0], dict(key="mine")), (str(scp), {}), (str(jsonl), {}), ([wavs[1], arr, wavs[2], arr], {}),
             ([arr, arr, wavs[0], arr], {}), ([arr, arr], dict(key="both")), ([arr, arr, arr], dict(key=["a", "b", "c"])),
             ((arr, arr), {}), (arr, {}), (arr, dict(key="one")), ("今天天气不错", {}), (torch.zeros(10), dict(key="t"))]

    def shape(keys):
        names = {}
        return [k if not (isinstance(k, str) and k.startswith("rand_key_")) else ("rand", names.setdefault(k, len(names))) for k in keys]
    for data_in, kw in cases:
        rk, rd = ref_prepare(data_in, **kw)
        ok, od = prepare_data_iterator(data_in, **kw)
        assert shape(ok) == shape(rk), (data_in if isinstance(data_in, str) else type(data_in), kw, ok, rk)
        assert len(od) == len(rd) and all((a is b) or (isinstance(a, str) and a == b) for a, b in zip(od, rd)), (kw, od, rd)
