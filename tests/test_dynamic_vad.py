"""DynamicStreamingVAD: the two scenarios of the reference's tests/test_dynamic_streaming_vad.py (a stand-in AutoModel that
uses the production cache initialiser of FsmnVADStreaming and cuts on the thresholds found in cache["stats"]) plus the
schedule / event folding."""
import torch

from funasr_amd.dynamic_vad import DEFAULT_SILENCE_SCHEDULE, DynamicStreamingVAD
from funasr_amd.fsmn_vad import FsmnVADStreaming
from funasr_amd.vad_decision import VadOptions


class _ThresholdAwareModel:
    sample_rate = 16000

    def __init__(self):
        self.model = FsmnVADStreaming.__new__(FsmnVADStreaming)
        torch.nn.Module.__init__(self.model)
        self.model.vad_opts = VadOptions(window_size_ms=200, sil_to_speech_time_thres=150, speech_to_sil_time_thres=150,
                                         frame_in_ms=10, sil_pdf_ids=[0], max_end_silence_time=800, speech_noise_thres=0.5)

    def generate(self, input, cache, **kwargs):
        if not cache:
            self.model.init_cache(cache, **{k: v for k, v in kwargs.items() if k in ("max_end_silence_time", "speech_noise_thres")})
        audio = torch.cat((cache.get("_test_audio", torch.empty(0)), input[0]))
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


def _vad(**kw):
    return DynamicStreamingVAD(_ThresholdAwareModel(), silence_schedule=[(float("inf"), 10000)], speech_noise_thres=0.73, **kw)


def test_first_feed_initialises_the_thresholds_of_the_vad_state():
    vad = _vad()
    vad.feed(torch.ones(960))
    assert vad.cache["stats"].max_end_sil_frame_cnt_thresh == 9850
    assert abs(vad.cache["stats"].speech_noise_thres - 0.73) < 1e-6


def test_first_feed_is_chunking_invariant():
    audio = torch.cat((torch.ones(16000), torch.zeros(32000)))
    one = _vad().feed(audio)
    split_vad = _vad()
    split = split_vad.feed(audio[:960])
    split.extend(split_vad.feed(audio[960:]))
    assert one == split == []


def test_schedule_and_event_folding():
    class Scripted:
        def __init__(self, events):
            self.events, self.seen = list(events), []

        def generate(self, input, cache, **kwargs):
            cache.setdefault("stats", type("S", (), {})())
            self.seen.append((len(input[0]), kwargs.get("is_final"), kwargs.get("max_end_silence_time"),
                              getattr(cache["stats"], "max_end_sil_frame_cnt_thresh", None)))
            return [{"value": self.events.pop(0) if self.events else []}]

    m = Scripted([[[120, -1]], [], [[-1, 900]], [[1500, 2100]], []])
    vad = DynamicStreamingVAD(m)
    assert vad.silence_schedule == DEFAULT_SILENCE_SCHEDULE and vad.current_threshold_ms == 2000
    chunk = torch.zeros(16000 * 3)                                # 3 s per feed
    assert vad.feed(chunk) == [] and vad.is_speaking and m.seen[0][2] == 2000
    assert vad.feed(chunk) == [] and vad.current_duration_ms == 6000 and m.seen[1][3] == 1500 - 150
    assert vad.feed(chunk) == [[120, 900]] and not vad.is_speaking and vad.current_duration_ms == 0
    assert vad.feed(chunk) == [[1500, 2100]]
    assert vad.finalize() == [] and m.seen[-1][1] is True and m.seen[-1][0] == 160
    assert vad.confirmed_segments == [[120, 900], [1500, 2100]]
    m2 = Scripted([[[0, -1]]] + [[]] * 8 + [[[-1, 570]]])
    assert DynamicStreamingVAD(m2).process(torch.zeros(9600)) == [[0, 570]] and len(m2.seen) == 10 and m2.seen[-1][1] is True


def _chunk_call(**kwargs):
    """the reference's tests/test_fsmn_vad_dynamic_silence.py: one 1000 ms streaming block while the state machine is inside a
    speech segment; frontend and network stubbed out (no frames come back), only the threshold rule runs"""
    from types import SimpleNamespace
    from funasr_amd.vad_decision import IN_SPEECH

    class Frontend:
        fs, frame_shift, lfr_n = 16000, 10, 1

        def init_cache(self, *a, **k):
            return {}

        def __call__(self, wav, lens, cache=None, is_final=False):
            return torch.zeros(0), torch.zeros(1)

    vad = FsmnVADStreaming.__new__(FsmnVADStreaming)
    torch.nn.Module.__init__(vad)
    vad.vad_opts = VadOptions(speech_to_sil_time_thres=100)
    vad.encoder = torch.nn.Linear(1, 1)                                     # only its device is looked at
    cache = {"frontend": {}, "prev_samples": torch.empty(0), "encoder": {}, "frames_done": 0, "wave": None, "wave_start": 0,
             "decision": SimpleNamespace(state=IN_SPEECH, max_end_sil_ms=200, speech_noise_thres=0.6)}
    cache["stats"] = cache["decision"]
    vad.inference(torch.zeros(16000), frontend=Frontend(), cache=cache, key=["utt"], chunk_size=1000, is_final=False,
                  device="cpu", **kwargs)
    return cache


def test_explicit_max_end_silence_time_keeps_the_fixed_threshold_unless_the_schedule_is_asked_for():
    cache = _chunk_call(max_end_silence_time=300)
    assert cache["stats"].max_end_sil_ms == 200 and "_dynamic_accumulated_ms" not in cache
    cache = _chunk_call(max_end_silence_time=300, dynamic_silence=True)
    assert cache["stats"].max_end_sil_ms == 1900 and cache["_dynamic_accumulated_ms"] == 1000
    cache = _chunk_call()                                                    # no explicit time: the schedule is the default
    assert cache["stats"].max_end_sil_ms == 1900 and cache["stats"].speech_noise_thres == 0.5
