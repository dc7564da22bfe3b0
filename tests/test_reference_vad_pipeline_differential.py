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
        asr_cls = _NoStampASR if trial % 6 == 5 else _FakeASR
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
