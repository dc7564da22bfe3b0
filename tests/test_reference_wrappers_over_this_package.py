"""Host-side wrappers of the reference that sit OUTSIDE the hot path are not restated in this package (round-3 review): the
reference's own modules serve them. What this package owes them is the surface they touch -- checked here by running the
REFERENCE's classes (build container only) over this package's objects:

  * `DynamicStreamingVAD` (funasr/models/fsmn_vad_streaming/dynamic_vad.py:37-230) writes `cache["stats"].speech_noise_thres`
    and `.max_end_sil_frame_cnt_thresh` before every chunk: `FsmnVADStreaming.init_cache` must hand out a state with those
    attributes, initialised from the same options (the scenario of the reference's tests/test_dynamic_streaming_vad.py);
  * the text-level hotword correction (funasr/utils/postprocess_hotwords.py) that ends `AutoModel.generate`
    (auto/auto_model.py:742-748): this package's `AutoModel.generate` hands the results to the reference module.
"""
import pytest
import torch

from oracle import ref_import

pytestmark = pytest.mark.skipif(not ref_import.available(), reason="reference checkout not present (GPU box)")


class _ThresholdAwareModel:
    """the stand-in AutoModel of the reference's own test: production cache initialiser, cuts on the thresholds in the cache"""
    sample_rate = 16000

    def __init__(self):
        from funasr_amd.fsmn_vad import FsmnVADStreaming
        from funasr_amd.vad_decision import VadOptions
        self.model = FsmnVADStreaming.__new__(FsmnVADStreaming)
        torch.nn.Module.__init__(self.model)
        self.model.vad_opts = VadOptions(window_size_ms=200, sil_to_speech_time_thres=150, speech_to_sil_time_thres=150,
                                         frame_in_ms=10, sil_pdf_ids=[0], max_end_silence_time=800, speech_noise_thres=0.5)

    def generate(self, input, cache, **kwargs):
        if not cache:
            self.model.init_cache(cache, **{k: v for k, v in kwargs.items() if k in ("max_end_silence_time", "speech_noise_thres")})
        audio = torch.cat((cache.get("_test_audio", torch.empty(0)), torch.as_tensor(input[0])))
        cache["_test_audio"] = audio
        speech = torch.nonzero(audio.abs() > 0.5)
        if not len(speech) or cache.get("_test_emitted"):
            return [{"value": []}]
        last = speech[-1].item()
        trailing_ms = int((len(audio) - last - 1) * 1000 / self.sample_rate)
        if trailing_ms < cache["stats"].max_end_sil_frame_cnt_thresh + self.model.vad_opts.speech_to_sil_time_thres:
            return [{"value": []}]
        cache["_test_emitted"] = True
        return [{"value": [[0, int((last + 1) * 1000 / self.sample_rate)]]}]


def _ref_vad(**kw):
    ref_import.install()
    from funasr.models.fsmn_vad_streaming.dynamic_vad import DynamicStreamingVAD
    return DynamicStreamingVAD(_ThresholdAwareModel(), silence_schedule=[(float("inf"), 10000)], speech_noise_thres=0.73, **kw)


def test_reference_dynamic_vad_wrapper_drives_this_packages_vad_state():
    vad = _ref_vad()
    vad.feed(torch.ones(960))
    assert vad.cache["stats"].max_end_sil_frame_cnt_thresh == 9850
    assert abs(vad.cache["stats"].speech_noise_thres - 0.73) < 1e-6
    audio = torch.cat((torch.ones(16000), torch.zeros(32000)))
    one = _ref_vad().feed(audio)
    split_vad = _ref_vad()
    split = split_vad.feed(audio[:960])
    split.extend(split_vad.feed(audio[960:]))
    assert one == split == []


def test_generate_hands_text_hotwords_to_the_reference_module():
    ref_import.install()
    from funasr_amd.auto_model import AutoModel
    am = AutoModel.__new__(AutoModel)
    am.vad_model, am.punc_model, am.kwargs = None, None, {}
    am.inference = lambda input, **cfg: [{"key": "k", "text": "今天 撒贝你 主持"}]
    out = am.generate("x", postprocess_hotwords=["撒贝你=>撒贝宁"], postprocess_hotword_fuzzy=False)
    assert out[0]["text"] == "今天 撒贝宁 主持"
    assert am.generate("x")[0]["text"] == "今天 撒贝你 主持"          # nothing requested: nothing imported, nothing changed


def test_text_hotwords_without_funasr_are_refused_not_ignored(monkeypatch):
    import sys
    from funasr_amd.auto_model import AutoModel
    monkeypatch.setitem(sys.modules, "funasr.utils.postprocess_hotwords", None)       # import -> ImportError
    am = AutoModel.__new__(AutoModel)
    am.vad_model, am.punc_model, am.kwargs = None, None, {}
    am.inference = lambda input, **cfg: [{"key": "k", "text": "abc"}]
    with pytest.raises(NotImplementedError):
        am.generate("x", postprocess_hotwords=["a=>b"])
