"""AutoModel plumbing (BASELINE configs[0]): a FunASR-format model directory is resolved through the registry by
name, the checkpoint loads strictly, `generate()` batches inputs and returns `[{key, text}]`. The executable spec in
the reference is tests/test_auto_model.py:43-71 (batch loop over `model.inference(...) -> (results, meta)`)."""
import os

import numpy as np
import pytest
import torch

from funasr_amd import synth
from funasr_amd.auto_model import AutoModel, prepare_data_iterator

from ._model_dir import VOCAB, make_model_dir, write_wav


@pytest.fixture(scope="module")
def model_dir(tmp_path_factory):
    d = str(tmp_path_factory.mktemp("paraformer_tiny"))
    info = make_model_dir(d)
    waves = {}
    for i, n in enumerate((40000, 24000, 56000)):
        p = os.path.join(d, f"utt{i}.wav")
        pcm = write_wav(p, synth.speech_like(n, seed=40 + i))
        waves[p] = pcm
    with open(os.path.join(d, "wav.scp"), "w") as f:
        for i, p in enumerate(waves):
            f.write(f"key_{i} {p}\n")
    return dict(dir=d, waves=waves, **info)


def test_prepare_data_iterator_forms(model_dir):
    d = model_dir["dir"]
    paths = list(model_dir["waves"])
    k, data = prepare_data_iterator(paths[0])
    assert k == ["utt0"] and data == [paths[0]]
    k, data = prepare_data_iterator(os.path.join(d, "wav.scp"))
    assert k == ["key_0", "key_1", "key_2"] and data == paths
    # lists: the reference keeps ONE key variable over the items (auto_model.py:397-405) -- a path sets it to its name and an
    # item without a name inherits whatever it holds; the first random key is reused (tests/test_reference_vad_pipeline_differential.py
    # compares this function with the reference's own)
    k, data = prepare_data_iterator([paths[1], np.zeros(16000, dtype=np.float32)])
    assert k == ["utt1", "utt1"] and len(data) == 2
    k, data = prepare_data_iterator([np.zeros(16000, dtype=np.float32), np.zeros(8000, dtype=np.float32), paths[0]])
    assert k[0].startswith("rand_key_") and k[1] == k[0] and k[2] == "utt0"
    # a string that is not a path is raw text for a punctuation model (auto_model.py:403-409); a missing wav path is
    # reported when the audio is loaded
    k, data = prepare_data_iterator("/nonexistent/a.wav")
    assert data == ["/nonexistent/a.wav"] and k[0].startswith("rand_key_")
    from funasr_amd.audio import load_audio
    with pytest.raises(FileNotFoundError):
        load_audio("/nonexistent/a.wav")


def test_build_from_model_dir_on_cpu_and_loud_failure(model_dir):
    am = AutoModel(model=model_dir["dir"], device="cpu", disable_update=True)
    assert type(am.model).__name__ == "Paraformer"
    assert am.kwargs.get("batch_size", 1) == 1 and am.kwargs["device"] == "cpu"     # an explicit device="cpu" sets no batch_size (auto_model.py:551-561)
    assert am.kwargs["ncpu"] == 4 and am.kwargs["vocab_size"] == len(VOCAB)
    assert am.kwargs["tokenizer"].get_num_vocabulary_size() == len(VOCAB)
    assert am.kwargs["frontend"].output_size() == 560 and tuple(am.kwargs["frontend"].cmvn.shape) == (2, 560)
    assert am.model.vocab_size == len(VOCAB)
    got = am.model.state_dict()
    for k, v in model_dir["sd"].items():
        assert torch.equal(got[k], v), k                       # strict load incl. the training-only embedding
    # no CPU implementation behind the boundary: a clear error, not a silent fallback
    with pytest.raises(RuntimeError, match="GPU"):
        am.generate(input=list(model_dir["waves"])[0])


def test_default_precision_and_the_reference_fp16_bf16_kwargs(model_dir):
    """the measured mode is the default; the reference's `bf16=True` / `fp16=True` construction kwargs
    (funasr/auto/auto_model.py:664-668) select the operand mode instead of raising"""
    am = AutoModel(model=model_dir["dir"], device="cpu")
    assert am.model.encoder._mode() == "f16x2" and am.model.decoder._mode() == "f16x2"
    assert AutoModel(model=model_dir["dir"], device="cpu", bf16=True).model.encoder._mode() == "bf16"
    assert AutoModel(model=model_dir["dir"], device="cpu", fp16=True).model.decoder._mode() == "f16x2"
    am.model.set_precision("fp32")
    assert am.model.encoder._mode() == "fp32" and am.model.decoder._mode() == "fp32"
    am.model.set_precision(None)
    assert am.model.encoder._mode() == "f16x2"
    from funasr_amd.sanm_encoder import SANMEncoder
    assert SANMEncoder(input_size=256, output_size=256, attention_heads=8, linear_units=1024, num_blocks=1, input_layer="pe")._mode() == "fp32"


def test_unknown_model_and_unbuilt_pipelines_raise(model_dir):
    with pytest.raises(FileNotFoundError):
        AutoModel(model="iic/not-a-local-dir", device="cpu")
    with pytest.raises(FileNotFoundError):                         # hub names need the network: local directories only
        AutoModel(model=model_dir["dir"], vad_model="fsmn-vad", device="cpu")
    with pytest.raises(FileNotFoundError):
        AutoModel(model=model_dir["dir"], punc_model="ct-punc", device="cpu")
    with pytest.raises(NotImplementedError):
        AutoModel(model=model_dir["dir"], spk_model="cam++", device="cpu")


@pytest.mark.gpu
def test_generate_matches_oracle_text(model_dir, cuda):
    from funasr_amd.tokenizer import CharTokenizer, sentence_postprocess
    from oracle import paraformer_oracle as O

    am = AutoModel(model=model_dir["dir"], device="cuda:0", batch_size=2)
    paths = list(model_dir["waves"])
    res = am.generate(input=os.path.join(model_dir["dir"], "wav.scp"))
    assert [r["key"] for r in res] == ["key_0", "key_1", "key_2"]
    assert am.speed_stats["rtf_avg"] is not None and float(am.speed_stats["rtf"]) >= 0
    # the same three clips through the CPU oracle (bs 1, like the reference on cpu) -> ids -> text
    cmvn = am.kwargs["frontend"].cmvn
    tok = CharTokenizer(token_list=VOCAB)
    for r, p in zip(res, paths):
        w = torch.from_numpy(model_dir["waves"][p].astype(np.float32) / 32768.0)
        feats, flens = O.wav_frontend([w], cmvn)
        ref = O.paraformer_greedy(feats, flens, model_dir["sd"], model_dir["cfg"])
        ids = [t for t in ref["raw_ids"][0] if t not in (0, 1, 2)]
        text, _ = sentence_postprocess(tok.ids2tokens(ids))
        assert r["text"] == text, (r, text)
    # mixed list input: path + ndarray + tensor, different batch size per call (runtime cfg override)
    x1 = model_dir["waves"][paths[1]].astype(np.float32) / 32768.0
    res2 = am.generate(input=[paths[0], x1, torch.from_numpy(x1)], batch_size=3)
    assert res2[0]["text"] == res[0]["text"] and res2[1]["text"] == res[1]["text"] == res2[2]["text"]
    assert am.kwargs["batch_size"] == 2 or am.kwargs["batch_size"] == 3


@pytest.mark.gpu
def test_generate_over_many_batches_overlapped_equals_the_plain_loop(model_dir, cuda, tmp_path):
    """`generate` over a list longer than the batch: the loop of auto_model.py:790-840 with the batches overlapped (pinned upload on
    its own stream, the decoder of batch i on a second stream beside the encoder of batch i + 1, text of batch i - 1 on the host
    meanwhile) returns the plain loop's records, in order, for ragged batches, a 50-ms clip and a last short batch."""
    am = AutoModel(model=model_dir["dir"], device="cuda:0", batch_size=3)
    paths = []
    for i, n in enumerate((40000, 9000, 56000, 31000, 16000, 47000, 52000, 23000, 800, 38000, 61000)):
        p = str(tmp_path / f"clip{i}.wav")
        write_wav(p, synth.speech_like(n, seed=70 + i))
        paths.append(p)
    plain = am.generate(input=paths, pipeline=False)
    over = am.generate(input=paths)
    assert over == plain and len(over) == len(paths)
    assert am.speed_stats["rtf_avg"] is not None
    assert am.model.__dict__.get("_dec_stream") is not None, "the overlapped loop did not run"
    # token timestamps: the two-part form (everything enqueued in `inference_begin`, text and spans in `inference_end`)
    stamped = am.generate(input=paths, pred_timestamp=True)
    assert stamped == am.generate(input=paths, pred_timestamp=True, pipeline=False) and all("timestamp" in r for r in stamped)
    # a rows budget instead of a count (this package's option): inputs taken longest first, batches cut by encoder rows, records back
    # in INPUT order with their own keys; a clip's text as in any other batch plan (up to the longest-clip property)
    by_rows = am.generate(input=paths, batch_size_rows=256)
    assert [r["key"] for r in by_rows] == [r["key"] for r in plain] and len(by_rows) == len(plain)
    assert sum(a["text"] == b["text"] for a, b in zip(by_rows, plain)) >= len(plain) - 4
    assert by_rows == am.generate(input=paths, batch_size_rows=256, pipeline=False)
    # keys, another batch size, again (the pinned buffers and both library slots are reused)
    keys = [f"k{i}" for i in range(len(paths))]
    assert am.generate(input=paths, key=keys, batch_size=4) == am.generate(input=paths, key=keys, batch_size=4, pipeline=False)
    # a file that cannot be read in the middle: the error surfaces, and the next call works (no ticket left open in the library)
    bad = paths[:5] + [str(tmp_path / "missing.wav")] + paths[5:]
    with pytest.raises(Exception):
        am.generate(input=bad)
    assert am.generate(input=paths) == plain


@pytest.mark.gpu
def test_generate_with_clips_of_a_few_milliseconds(model_dir, cuda, tmp_path):
    """Files shorter than one 25-ms analysis window in the list (5 ms, 19 ms): the reference's frontend gives each ONE frame from a window of
    its own length (wav_frontend.py:176) and decodes on; `generate` used to fail the whole call. Same records from the plain loop, the
    overlapped loop and the rows-budget plan; the tiny clips' text equals the CPU oracle's."""
    from funasr_amd.tokenizer import CharTokenizer, sentence_postprocess
    from oracle import paraformer_oracle as O
    am = AutoModel(model=model_dir["dir"], device="cuda:0", batch_size=2)
    paths, pcm = [], {}
    for i, n in enumerate((40000, 300, 9000, 80, 16000)):
        p = str(tmp_path / f"tiny{i}.wav")
        pcm[p] = write_wav(p, synth.speech_like(n, seed=40 + i))
        paths.append(p)
    plain = am.generate(input=paths, pipeline=False)
    assert len(plain) == len(paths) and am.generate(input=paths) == plain
    by_rows = am.generate(input=paths, batch_size_rows=128)
    assert [r["key"] for r in by_rows] == [r["key"] for r in plain]
    cmvn = am.kwargs["frontend"].cmvn
    tok = CharTokenizer(token_list=VOCAB)
    # the oracle on the plain loop's own batches (two files each, in input order): what a clip's last frame sees behind it depends on
    # its batch (the predictor's conv reads one encoder row past the clip, a reference property), so the comparison keeps the batches
    for b0 in (0, 2):
        ws = [torch.from_numpy(pcm[p].astype(np.float32) / 32768.0) for p in paths[b0:b0 + 2]]
        feats, flens = O.wav_frontend(ws, cmvn)
        assert flens.tolist()[1] == 1                              # the tiny clip: one LFR row from its one frame
        ref = O.paraformer_greedy(feats, flens, model_dir["sd"], model_dir["cfg"])
        for j in range(2):
            ids = [t for t in ref["raw_ids"][j] if t not in (0, 1, 2)]
            text, _ = sentence_postprocess(tok.ids2tokens(ids)) if ids else ("", None)
            assert plain[b0 + j]["text"] == text, (b0 + j, plain[b0 + j], text)


@pytest.mark.gpu
def test_sensevoice_batches_overlapped_equal_the_plain_loop(cuda):
    from funasr_amd.sense_voice import SenseVoiceSmall
    from funasr_amd.wav_frontend import WavFrontend
    cfg = synth.tiny(synth.SENSEVOICE_SMALL, enc_blocks=2, tp_blocks=1, vocab=997)
    model = SenseVoiceSmall.from_config(cfg)
    model.load_state_dict(synth.sensevoice_state_dict(cfg, seed=3), strict=False)
    model = model.to("cuda:0").eval()
    sh, sc = synth.synthetic_cmvn(560)
    fe = WavFrontend(cmvn=torch.stack([sh, sc]), lfr_m=7, lfr_n=6, dither=0.0, device="cuda:0")
    am = _bare_auto_model(model, batch_size=3, device="cuda:0", frontend=fe, tokenizer=None)
    am._base_kwargs = {k: v for k, v in am.kwargs.items() if k not in ("frontend", "tokenizer")}
    clips = [synth.speech_like(n, seed=90 + i) for i, n in enumerate((30000, 12000, 44000, 21000, 16000, 52000, 9000, 38000))]
    plain = AutoModel.inference(am, clips, pipeline=False)
    over = AutoModel.inference(am, clips)
    ids = lambda res: [r["token_int"] for r in res]              # (keys are random per call for tensor inputs, like the reference's)
    assert ids(over) == ids(plain) and len(over) == len(clips)
    assert len({tuple(t) for t in ids(over)}) > 1
    assert model.__dict__.get("_upload") is not None, "the overlapped loop did not run"


@pytest.mark.gpu
def test_generate_with_vad_segments_on_the_hip_path(model_dir, cuda):
    """inference_with_vad (auto_model.py:852-1254) with a stand-in VAD model: a long recording is cut at the VAD's
    segments, decoded in length-sorted batches by the HIP path and merged; every segment's text and (shifted)
    timestamps equal decoding that segment alone."""
    class FixedVAD:
        def __init__(self, segments):
            self.segments = segments

        def parameters(self):
            return iter(())

        def inference(self, data_in, key=None, **kwargs):
            return [{"key": key[0], "value": [list(s) for s in self.segments]}], {"batch_data_time": 1.0}

    paths = list(model_dir["waves"])
    clips = [torch.from_numpy(model_dir["waves"][p].astype(np.float32) / 32768.0) for p in paths]
    gap = torch.zeros(8000)
    long = torch.cat([clips[0], gap, clips[1], gap, gap, clips[2], gap])
    segs, t = [], 0
    for c, g in zip(clips, (1, 2, 1)):
        segs.append([t // 16, (t + c.numel()) // 16])
        t += c.numel() + g * 8000
    am = AutoModel(model=model_dir["dir"], device="cuda:0", vad_model=FixedVAD(segs))
    single = AutoModel(model=model_dir["dir"], device="cuda:0")
    texts, stamps = [], []
    for (b, e) in segs:
        r = single.generate(input=long[b * 16: e * 16], pred_timestamp=True)[0]
        texts.append(r["text"])
        stamps += [[x + b, y + b] for x, y in r["timestamp"]]
    # budget 0 s: every segment is its own ASR batch -> identical to decoding the segments one by one
    out = am.generate(input=long, pred_timestamp=True, batch_size_s=0)
    assert len(out) == 1 and out[0]["text"] == " ".join(texts)
    assert out[0]["timestamp"] == stamps and len(stamps) > 0
    # default budget (300 s): one batch of the three segments in ascending-duration order. A padded batch is not the
    # same computation as three single decodes in the reference either (the CIF conv peeks into the padded frames,
    # cif_predictor.py:275-277, and the timestamp function gets rows of the batch tensors, paraformer/model.py:669), so
    # the expectation is that very batch decoded directly, restored to recording order and shifted
    out2 = am.generate(input=long, pred_timestamp=True)
    order = sorted(range(len(segs)), key=lambda j: segs[j][1] - segs[j][0])
    direct = single.generate(input=[long[segs[j][0] * 16: segs[j][1] * 16] for j in order], pred_timestamp=True, batch_size=3)
    by_seg = {j: direct[pos] for pos, j in enumerate(order)}
    assert out2[0]["text"] == " ".join(by_seg[j]["text"] for j in range(len(segs)))
    assert out2[0]["timestamp"] == [[x + segs[j][0], y + segs[j][0]] for j in range(len(segs)) for x, y in by_seg[j]["timestamp"]]


@pytest.mark.gpu
def test_vad_model_directory_end_to_end(model_dir, cuda, tmp_path):
    """AutoModel(model=<Paraformer dir>, vad_model=<FSMN-VAD dir>): the VAD network + decision logic cut a recording
    with three bursts, the ASR decodes the cuts; the VAD alone (AutoModel(model=<VAD dir>)) reports the same segments."""
    from tests._model_dir import VAD_ENCODER_CONF, make_vad_model_dir
    from tests.test_vad_gpu import _energy_tracking_weights
    vad_dir = str(tmp_path / "vad")
    make_vad_model_dir(vad_dir, _energy_tracking_weights(VAD_ENCODER_CONF))
    fs = 16000
    long = 1e-4 * torch.randn(26 * fs, generator=torch.Generator().manual_seed(8))
    for i, (a, b) in enumerate([(1.0, 5.5), (8.0, 11.0), (15.0, 24.0)]):
        seg = synth.speech_like(int((b - a) * fs), seed=70 + i)
        long[int(a * fs): int(a * fs) + seg.numel()] += seg
    vad = AutoModel(model=vad_dir, device="cuda:0")
    segs = vad.generate(input=long)[0]["value"]
    assert len(segs) == 3 and all(abs(s[0] / 1000 - a) < 0.5 and abs(s[1] / 1000 - b) < 1.2
                                   for s, (a, b) in zip(segs, [(1.0, 5.5), (8.0, 11.0), (15.0, 24.0)])), segs
    am = AutoModel(model=model_dir["dir"], device="cuda:0", vad_model=vad_dir)
    out = am.generate(input=long, batch_size_s=0)                              # one ASR batch per segment
    single = AutoModel(model=model_dir["dir"], device="cuda:0")
    texts = [single.generate(input=long[b * 16: min(e * 16, long.numel())])[0]["text"] for b, e in segs]
    assert len(out) == 1 and out[0]["text"] == " ".join(texts)
    # several segments per batch plan, overlapped (the default) against the plain loop: the same record
    long2 = torch.cat([long, long.roll(4000), long.roll(9000)])
    many = am.generate(input=long2, batch_size_s=6)
    assert many[0]["text"] == am.generate(input=long2, batch_size_s=6, pipeline=False)[0]["text"] and len(many[0]["text"]) > len(out[0]["text"])
    assert am.model.__dict__.get("_dec_stream") is not None, "the segment batches did not overlap"
    # several recordings with a rows budget: their segments share batches (this package's own mode) -- the texts per recording are
    # those of the per-recording batches and of the plain loop
    recs = [long, long.roll(4000)[: 20 * fs], long2, 1e-4 * torch.randn(3 * fs, generator=torch.Generator().manual_seed(9)), long.roll(9000)]
    shared = am.generate(input=recs, batch_size_rows=512)
    txt = lambda res: [r["text"] for r in res]
    assert txt(shared) == txt(am.generate(input=recs, batch_size_rows=512, pipeline=False))          # the same batches, one after the other
    # against per-recording batches: the same texts up to the last token of a clip that is the longest of its batch in one plan and
    # not in the other (CifPredictorV2 reads the row behind the last frame: a property of the reference's padded batches)
    apart = am.generate(input=recs, batch_size_rows=512, batch_across_recordings=False)
    assert len(shared) == len(apart) == 5 and shared[3]["text"] == apart[3]["text"] == ""
    # the VAD alone over several recordings: the recordings' score passes overlap (FsmnVADStreaming.inference_begin / _end), same segments
    vals = lambda res: [r["value"] for r in res]
    assert vals(vad.generate(input=recs)) == vals(vad.generate(input=recs, pipeline=False)) and vals(vad.generate(input=recs))[0] == segs
    assert vad.model.__dict__.get("_host_ring") is not None, "the VAD's overlapped form did not run"
    import difflib
    for a, b in zip(txt(shared), txt(apart)):
        assert a == b or difflib.SequenceMatcher(None, a, b, autojunk=False).ratio() > 0.95, (a, b)
    # the whole long-form chain from three model directories: VAD -> ASR (with token timestamps) -> CT-Transformer.
    # The text is what the punctuation model makes of the joined segment texts, the sentence records are cut at its marks
    from funasr_amd.timestamps import timestamp_sentence
    from funasr_amd.vad_utils import join_vad_texts
    from tests.test_punctuation import _punc_dir
    punc_dir, _ = _punc_dir(tmp_path)
    full = AutoModel(model=model_dir["dir"], device="cuda:0", vad_model=vad_dir, punc_model=punc_dir)
    res = full.generate(input=long, batch_size_s=0, pred_timestamp=True, sentence_timestamp=True, return_raw_text=True)[0]
    stamped = [single.generate(input=long[b * 16: min(e * 16, long.numel())], pred_timestamp=True)[0] for b, e in segs]
    punc_in = join_vad_texts(r["text"] for r in stamped)
    punc = AutoModel(model=punc_dir, device="cuda:0").generate(input=punc_in)[0]
    # with token timestamps the ASR text is one blank-separated word per stamp (sentence_postprocess with time_stamp)
    assert res["raw_text"] == " ".join(r["text"] for r in stamped) and res["raw_text"].replace(" ", "") == " ".join(texts).replace(" ", "")
    assert res["text"] == punc["text"] and res["text"] != res["raw_text"]
    stamps = [[x + b, y + b] for r, (b, e) in zip(stamped, segs) for x, y in r["timestamp"]]
    assert res["timestamp"] == stamps
    if len(punc["punc_array"]) == len(stamps):
        assert res["sentence_info"] == timestamp_sentence(punc["punc_array"], stamps, punc_in, return_raw_text=True)
        assert res["sentence_info"] and res["sentence_info"][-1]["end"] == stamps[-1][1]
    else:                                                                       # misaligned: VAD-segment records
        assert [s["start"] for s in res["sentence_info"]] == [b for b, e in segs]


def test_audio_inputs_resample_and_bytes(tmp_path):
    """load_audio handles the input kinds of load_audio_text_image_video (load_utils.py:48-179) that the ASR path sees."""
    import wave as _wave
    from funasr_amd.audio import load_audio, load_audio_list
    x = synth.speech_like(8000, seed=1)
    p8 = str(tmp_path / "a8k.wav")
    pcm = write_wav(p8, x, fs=8000)
    y = load_audio(p8, fs=16000)                                  # 8 kHz file -> 16 kHz
    assert abs(y.numel() - 16000) <= 1 and y.dtype == torch.float32
    raw = pcm.tobytes()                                            # headerless 16-bit PCM bytes
    z = load_audio(raw, fs=16000, audio_fs=16000)
    assert z.numel() == 8000 and torch.allclose(z, torch.from_numpy(pcm.astype(np.float32) / 32768.0))
    stereo = np.stack([pcm, pcm // 2]).astype(np.int16)
    with _wave.open(str(tmp_path / "st.wav"), "wb") as f:
        f.setnchannels(2); f.setsampwidth(2); f.setframerate(16000)
        f.writeframes(stereo.T.copy().tobytes())
    m = load_audio(str(tmp_path / "st.wav"))
    assert m.numel() == 8000                                       # channel mean (load_utils.py:126-127)
    assert len(load_audio_list([p8, x.numpy()], fs=8000, audio_fs=8000)) == 2
    with pytest.raises(FileNotFoundError):
        load_audio("/nope.wav")


def test_sentencepiece_tokenizer_registered(tmp_path):
    import sentencepiece as spm
    from funasr_amd.register import tables
    corpus = tmp_path / "c.txt"
    corpus.write_text("\n".join(["hello world this is a test", "speech recognition on amd gpus", "paraformer and sensevoice"] * 20))
    spm.SentencePieceTrainer.train(input=str(corpus), model_prefix=str(tmp_path / "m"), vocab_size=40, model_type="bpe",
                                   minloglevel=2)
    cls = tables.tokenizer_classes.get("SentencepiecesTokenizer")
    tok = cls(bpemodel=str(tmp_path / "m.model"))
    ids = tok.encode("hello world")
    assert tok.decode(ids) == "hello world" and tok.get_vocab_size() == 40
    assert tok.tokens2text(tok.ids2tokens(ids)) == "hello world"


def test_metrics_and_datadir_writer(tmp_path):
    from funasr_amd.datadir_writer import DatadirWriter
    from funasr_amd.metrics import cer, edit_distance, micro_error_rate
    assert edit_distance("kitten", "sitting") == 3 and edit_distance([], [1, 2]) == 2 and edit_distance([1, 2, 3], [1, 2, 3]) == 0
    rate, edits, total = micro_error_rate([[1, 2, 3, 4], [5, 6]], [[1, 2, 4], [5, 6, 7]])
    assert (edits, total) == (2, 6) and abs(rate - 2 / 6) < 1e-12
    assert cer(["欢迎 大家，来体验!"], ["欢迎大家来体验"]) == 0.0 and cer(["abc"], ["ABD"]) == 1 / 3
    with DatadirWriter(str(tmp_path / "out")) as w:
        w["1best_recog"]["text"]["utt1"] = "你好"
        w["1best_recog"]["token"]["utt1"] = "你 好"
    assert (tmp_path / "out" / "1best_recog" / "text").read_text(encoding="utf-8") == "utt1 你好\n"
    assert (tmp_path / "out" / "1best_recog" / "token").read_text(encoding="utf-8") == "utt1 你 好\n"


def test_progress_callback_called_like_the_reference_spec():
    """The reference's executable spec of the AutoModel -> model boundary (tests/test_auto_model.py:43-71 there): a model
    object exposing only parameters()/eval()/inference(data_in=..., **kw) -> (results, {"batch_data_time": 1}) is driven in
    batches of `batch_size`, and the progress callback sees (items done, total) after every batch."""
    class DummyModel:
        def __init__(self):
            self.param = torch.nn.Parameter(torch.zeros(1))

        def parameters(self):
            return iter([self.param])

        def eval(self):
            pass

        def inference(self, data_in=None, **kwargs):
            return [{"text": str(d)} for d in data_in], {"batch_data_time": 1}

    am = AutoModel.__new__(AutoModel)
    am.model = DummyModel()
    am.kwargs = {"batch_size": 2, "disable_pbar": True}
    am._base_kwargs = dict(am.kwargs)
    progress = []
    res = AutoModel.inference(am, ["a", "b", "c"], progress_callback=lambda idx, total: progress.append((idx, total)))
    assert progress == [(2, 3), (3, 3)]
    assert [r["text"] for r in res] == ["a", "b", "c"]


class _SplitDummy:
    """a model that offers `inference` in three parts (funasr_amd/paraformer.py inference_begin / _launch / _end) and logs the calls"""

    def __init__(self, split=True, fail_at=None):
        self.param = torch.nn.Parameter(torch.zeros(1))
        self.log, self.split, self.fail_at, self.open = [], split, fail_at, set()

    def parameters(self):
        return iter([self.param])

    def eval(self):
        pass

    def inference(self, data_in=None, **kwargs):
        self.log.append(("whole", tuple(data_in)))
        return [{"text": str(d)} for d in data_in], {"batch_data_time": 1}

    def inference_begin(self, data_in=None, **kwargs):
        if not self.split:
            return None
        if self.fail_at is not None and data_in[0] == self.fail_at:
            raise RuntimeError("cannot load " + str(data_in[0]))
        self.log.append(("begin", tuple(data_in)))
        self.open.add(tuple(data_in))
        return {"data": tuple(data_in)}

    def inference_launch(self, pending):
        self.log.append(("launch", pending["data"]))
        pending["launched"] = True

    def inference_end(self, pending):
        if not pending.get("launched"):
            self.inference_launch(pending)
        self.log.append(("end", pending["data"]))
        self.open.discard(pending["data"])
        return [{"text": str(d)} for d in pending["data"]], {"batch_data_time": 1}


def _bare_auto_model(model, **kw):
    am = AutoModel.__new__(AutoModel)
    am.model = model
    am.kwargs = dict({"batch_size": 2, "disable_pbar": True}, **kw)
    am._base_kwargs = dict(am.kwargs)
    return am


def test_batches_overlap_in_three_stages_when_the_model_offers_them():
    """AutoModel.inference over several batches: begin(i + 1) | launch(i) | end(i - 1), records and progress as the plain loop's"""
    m = _SplitDummy()
    progress = []
    res = AutoModel.inference(_bare_auto_model(m), list("abcdefg"), progress_callback=lambda i, n: progress.append((i, n)))
    assert [r["text"] for r in res] == list("abcdefg")
    assert progress == [(2, 7), (4, 7), (6, 7), (7, 7)]
    b = [("a", "b"), ("c", "d"), ("e", "f"), ("g",)]
    assert m.log == [("begin", b[0]), ("begin", b[1]), ("launch", b[0]), ("begin", b[2]), ("launch", b[1]), ("end", b[0]),
                     ("begin", b[3]), ("launch", b[2]), ("end", b[1]), ("launch", b[3]), ("end", b[2]), ("end", b[3])]
    # pipeline=False, a single batch, and a model whose configuration has no split form: the plain loop
    for am, items in ((_bare_auto_model(_SplitDummy(), pipeline=False), "abc"), (_bare_auto_model(_SplitDummy()), "ab"),
                      (_bare_auto_model(_SplitDummy(split=False)), "abc")):
        res = AutoModel.inference(am, list(items))
        assert [r["text"] for r in res] == list(items)
        assert all(kind == "whole" for kind, _ in am.model.log) and len(am.model.log) == (len(items) + 1) // 2


def test_a_failing_batch_leaves_no_batch_open():
    m = _SplitDummy(fail_at="e")
    with pytest.raises(RuntimeError, match="cannot load e"):
        AutoModel.inference(_bare_auto_model(m), list("abcdefg"))
    assert not m.open and ("end", ("c", "d")) in m.log
