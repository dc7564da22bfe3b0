"""Fuzz the host-side text path against the REFERENCE functions (build container only; TEST INFRASTRUCTURE):
sentence_postprocess with/without timestamps (40k calls), merge_vad (3k), ts_prediction_lfr6_standard (3k), incl. the
error behaviour. Prints mismatch counts; 0/0/0 at the time of the round-1 commit."""
import sys, random, copy
sys.path.insert(0, "/root/repo")
import numpy as np, torch
from oracle import ref_import
ref_import.install()
from funasr.utils.timestamp_tools import ts_prediction_lfr6_standard
from funasr.utils.postprocess_utils import sentence_postprocess as ref_pp
from funasr.utils.vad_utils import merge_vad as ref_merge
from funasr_amd.timestamps import cif_timestamps
from funasr_amd.tokenizer import sentence_postprocess
from funasr_amd.vad_utils import merge_vad
rng = random.Random(1)
cjk = list("今天天气真不错我们一起去公园散步吧欢迎大家来体验语音识别模型")
alpha = ["hel@@", "lo", "wor@@", "ld", "the", "quick", "a", "i", "b", "m", "c", "don't", "re@@", "cog@@", "tion", "x", "y", "z", "ok", "U", "S", "A"]
other = ["3d", "<unk>", "</s>", "<s>", "1", "2", "@", "'", "a1", ",", "。", "mp3", "你好", "ab"]
bad = 0
for trial in range(20000):
    n = rng.randint(0, 14)
    kind = rng.random()
    toks = []
    for _ in range(n):
        r = rng.random()
        if kind < 0.25: toks.append(rng.choice(cjk))
        elif kind < 0.5: toks.append(rng.choice(alpha))
        else: toks.append(rng.choice(cjk) if r < 0.4 else rng.choice(alpha) if r < 0.8 else rng.choice(other))
    try: want = ref_pp(list(toks))
    except Exception as e: want = ("EXC", type(e).__name__)
    try: got = sentence_postprocess(list(toks))
    except Exception as e: got = ("EXC", type(e).__name__)
    if tuple(want) != tuple(got):
        bad += 1
        if bad < 6: print("PP", toks, want, got)
    ts = [[100 * i, 100 * i + rng.randint(10, 99)] for i in range(len(toks))]
    try: want = ref_pp(list(toks), [list(t) for t in ts])
    except Exception as e: want = ("EXC", type(e).__name__)
    try: got = sentence_postprocess(list(toks), [list(t) for t in ts])
    except Exception as e: got = ("EXC", type(e).__name__)
    if tuple(want) != tuple(got):
        bad += 1
        if bad < 12: print("PPTS", toks, want, got)
print("postprocess mismatches:", bad)
bad = 0
for trial in range(3000):
    n = rng.randint(0, 10); t = rng.randint(0, 400); segs = []
    for _ in range(n):
        d = rng.randint(100, 9000); segs.append([t, t + d]); t += d + rng.randint(0, 3000)
    mx, mn = rng.choice([3000, 15000, 30000]), rng.choice([0, 500, 2000])
    if ref_merge([list(s) for s in segs], mx, mn) != merge_vad([list(s) for s in segs], mx, mn): bad += 1
print("merge_vad mismatches:", bad)
bad = 0
nprng = np.random.default_rng(3)
for trial in range(3000):
    n = rng.randint(1, 12)
    T = rng.randint(n + 2, 120)
    a = torch.from_numpy(nprng.random(T).astype(np.float32)) * 0.4
    p = torch.from_numpy(nprng.random(T).astype(np.float32)) * 0.9
    k = rng.choice([n + 1, n + 1, n, n + 2, 0])
    idx = nprng.choice(T, size=min(k, T), replace=False)
    p[idx] = 1.0 + torch.from_numpy(nprng.random(len(idx)).astype(np.float32)) * 0.5
    toks = [rng.choice(cjk + alpha) for _ in range(n)] + (["</s>"] if rng.random() < 0.2 else [])
    kw = dict(vad_offset=rng.choice([0.0, 500.0]), upsample_rate=rng.choice([1, 3]), sil_in_str=rng.random() < 0.8)
    try: want = ts_prediction_lfr6_standard(a.clone(), p.clone(), list(toks), **kw)
    except Exception as e: want = ("EXC", type(e).__name__)
    try: got = cif_timestamps(a.clone(), p.clone(), list(toks), **kw)
    except Exception as e: got = ("EXC", type(e).__name__)
    if want != got:
        bad += 1
        if bad < 5: print("TS", toks, kw, want, got)
    # the vectorised milliseconds-only variant the model classes call
    from funasr_amd.timestamps import cif_token_spans
    try: got2 = cif_token_spans(a.clone(), p.clone(), list(toks), vad_offset=kw["vad_offset"], upsample_rate=kw["upsample_rate"])
    except Exception as e: got2 = ("EXC", type(e).__name__)
    if (want if isinstance(want, tuple) and want[0] == "EXC" else want[1]) != got2:
        bad += 1
        if bad < 5: print("TS-fast", toks, kw, want, got2)
print("timestamp mismatches:", bad)
from funasr.utils.timestamp_tools import timestamp_sentence as ref_ts, timestamp_sentence_en as ref_ts_en
from funasr_amd.timestamps import timestamp_sentence
bad = 0
for trial in range(6000):
    n = rng.randint(0, 16)
    words = [rng.choice(cjk + ["hello", "world", "a", "GPU", "ok"]) for _ in range(n)]
    nts = max(0, n + rng.choice([0, 0, 0, -1, 1]))
    ts = [[100 * i, 100 * i + rng.randint(10, 99)] for i in range(nts)]
    npc = max(0, n + rng.choice([0, 0, 0, -1, 1, -n]))
    pids = [rng.choice([1, 1, 1, 2, 3, 4, 5]) for _ in range(npc)]
    pid_arg = rng.choice([pids, torch.tensor(pids) if pids else pids, None if not pids else pids])
    text = " ".join(words) if rng.random() < 0.95 else None
    raw = rng.random() < 0.5
    for en, ref_fn in ((False, ref_ts), (True, ref_ts_en)):
        try: want = ref_fn(copy.deepcopy(pid_arg), copy.deepcopy(ts), text, return_raw_text=raw)
        except Exception as e: want = ("EXC", type(e).__name__)
        try: got = timestamp_sentence(copy.deepcopy(pid_arg), copy.deepcopy(ts), text, return_raw_text=raw, english=en)
        except Exception as e: got = ("EXC", type(e).__name__)
        if want != got:
            bad += 1
            if bad < 6: print("TSENT", en, words, ts, pids, raw, want, got)
print("timestamp_sentence mismatches:", bad)

# _join_vad_texts / _vad_segment_sentences live in funasr/auto/auto_model.py, whose import pulls the whole package: lift
# the two function definitions out of the reference source with ast and run them as they stand
import ast, re
_src = open("/root/reference/funasr/auto/auto_model.py", encoding="utf-8").read()
_ns = {"re": re}
for node in ast.parse(_src).body:
    if isinstance(node, ast.FunctionDef) and node.name in ("_join_vad_texts", "_vad_segment_sentences"):
        exec(compile(ast.Module([node], []), "auto_model.py", "exec"), _ns)
from funasr_amd.vad_utils import join_vad_texts, vad_segment_sentences
bad = 0
pieces = ["今天", "天气 ", " hello", "world", "<|zh|>", "<|NEUTRAL|><|Speech|>", "ok。", "", "  ", "3d", "、好", "㐀", "鿿x", "<|en|>the cat"]
for trial in range(5000):
    n = rng.randint(0, 6)
    texts = ["".join(rng.choice(pieces) for _ in range(rng.randint(0, 3))) for _ in range(n)]
    if _ns["_join_vad_texts"](list(texts)) != join_vad_texts(list(texts)):
        bad += 1
        if bad < 6: print("JOIN", texts)
    segs, t = [], 0
    for _ in range(n):
        t += rng.randint(10, 900); b = t; t += rng.randint(50, 5000); segs.append([b, t])
    rd = [{"text": x} for x in texts]
    try: want = _ns["_vad_segment_sentences"](copy.deepcopy(rd), copy.deepcopy(segs))
    except Exception as e: want = ("EXC", type(e).__name__)
    try: got = vad_segment_sentences(copy.deepcopy(rd), copy.deepcopy(segs))
    except Exception as e: got = ("EXC", type(e).__name__)
    if want != got:
        bad += 1
        if bad < 6: print("SEGS", texts, want, got)
print("join_vad_texts / vad_segment_sentences mismatches:", bad)

# rich_transcription_postprocess / sentence_postprocess_sentencepiece (funasr/utils/postprocess_utils.py:281-480)
from funasr.utils.postprocess_utils import rich_transcription_postprocess as ref_rich, sentence_postprocess_sentencepiece as ref_sp
from funasr_amd.postprocess_utils import rich_transcription_postprocess, sentence_postprocess_sentencepiece
from oracle.make_golden_postprocess import PIECES, rich_case
bad = 0
for trial in range(20000):
    s = rich_case(rng)
    try: want = ref_rich(s)
    except Exception as e: want = ("EXC", type(e).__name__)
    try: got = rich_transcription_postprocess(s)
    except Exception as e: got = ("EXC", type(e).__name__)
    if want != got:
        bad += 1
        if bad < 6: print("RICH", repr(s), repr(want), repr(got))
    w = [rng.choice(PIECES) for _ in range(rng.randint(0, 12))]
    if tuple(ref_sp(list(w))) != tuple(sentence_postprocess_sentencepiece(list(w))):
        bad += 1
        if bad < 6: print("SP", w)
print("rich_transcription_postprocess / sentencepiece mismatches:", bad)

# the surface-alignment helpers of inference_with_vad (funasr/auto/auto_model.py:107-300) vs funasr_amd/punc_align.py
_names = ("_get_punc_tokens", "_surface_token_spans", "_punc_symbol", "_punctuate_surface_text", "_merge_timestamp_units",
          "_timestamp_sentences_from_surface")
for node in ast.parse(_src).body:
    if isinstance(node, ast.FunctionDef) and node.name in _names:
        exec(compile(ast.Module([node], []), "auto_model.py", "exec"), _ns)
from funasr_amd import punc_align as PA


class _PM:
    jieba_usr_dict = None
    def __init__(self, marks): self.punc_list = marks


bad = 0
units = ["你", "好", "世", "界", "hello", "World", "don't", "a", "3d", "https", ":", "/", "nature", ".", "com", "@", "ok"]
for trial in range(6000):
    n = rng.randint(0, 8)
    toks = [rng.choice(units) for _ in range(n)]
    # surface text: tokens glued with or without blanks (ASCII runs need a blank to stay separate words)
    text = ""
    for t in toks:
        text += (" " if (text and (rng.random() < 0.5 or (text[-1].isascii() and t[0].isascii()))) else "") + t
    if rng.random() < 0.1: text += rng.choice([" ", "x", "。"])
    pm = _PM(rng.choice([["<unk>", "_", "，", "。", "？", "、"], None, ["<unk>", "_", ","]]))
    k = rng.choice([n, n, n, n + 1, max(n - 1, 0)])
    pa = [rng.choice([1, 1, 2, 3, 4, 5]) for _ in range(k)]
    pa_arg = rng.choice([pa, np.array(pa, dtype=np.int64), torch.tensor(pa)]) if rng.random() < 0.95 else 3
    ts = [[100 * i, 100 * i + rng.randint(0, 99)] for i in range(n)]
    # ASR units: sometimes the same tokens, sometimes split differently
    words, wts = [], []
    for t, (b, e) in zip(toks, ts):
        if len(t) > 2 and rng.random() < 0.4:
            c = rng.randint(1, len(t) - 1); m = (b + e) // 2
            words += [t[:c], t[c:]]; wts += [[b, m], [m, e]]
        else:
            words.append(t); wts.append([b, e])
    if rng.random() < 0.05 and wts: wts[-1] = [wts[-1][1] + 5, wts[-1][0]]
    raw = rng.random() < 0.5
    pairs = [
        (lambda: _ns["_get_punc_tokens"](text, pa_arg, pm), lambda: PA.punc_tokens(text, pa_arg, pm)),
        (lambda: _ns["_surface_token_spans"](text, toks), lambda: PA.surface_token_spans(text, toks)),
        (lambda: _ns["_punctuate_surface_text"](text, pa_arg, pm), lambda: PA.punctuate_surface_text(text, pa_arg, pm)),
        (lambda: _ns["_merge_timestamp_units"](text, words, wts, pa_arg, pm), lambda: PA.merge_timestamp_units(text, words, wts, pa_arg, pm)),
        (lambda: _ns["_timestamp_sentences_from_surface"](text, ts, pa_arg, pm, return_raw_text=raw),
         lambda: PA.timestamp_sentences_from_surface(text, ts, pa_arg, pm, return_raw_text=raw)),
    ]
    if toks and pa:
        pairs.append((lambda: _ns["_punc_symbol"](pa[0], toks[0], pm), lambda: PA.punc_symbol(pa[0], toks[0], pm)))
    for fi, (ref_f, my_f) in enumerate(pairs):
        try: want = ref_f()
        except Exception as e: want = ("EXC", type(e).__name__)
        try: got = my_f()
        except Exception as e: got = ("EXC", type(e).__name__)
        norm = lambda v: [tuple(x) if isinstance(x, (list, tuple)) else x for x in v] if isinstance(v, list) else v
        if norm(want) != norm(got) and str(want) != str(got):
            bad += 1
            if bad < 8: print("ALIGN", fi, repr(text), toks, pa_arg, want, got)
print("surface-alignment helper mismatches:", bad)

# byte-container sniffing (funasr/utils/load_utils.py:182-303) vs funasr_amd.audio.is_audio_container
_lu = open("/root/reference/funasr/utils/load_utils.py", encoding="utf-8").read()
_ns2 = {}
for node in ast.parse(_lu).body:
    if isinstance(node, ast.FunctionDef) and node.name in ("_mp3_header_fields", "_mp3_frame_length", "_has_consecutive_mp3_frames", "_is_audio_container"):
        exec(compile(ast.Module([node], []), "load_utils.py", "exec"), _ns2)
from funasr_amd.audio import is_audio_container
bad = 0
magic = [b"RIFF", b"RIFX", b"RF64", b"BW64", b"ID3", b"OggS", b"fLaC", b"\x1a\x45\xdf\xa3", b"\xff\xfb", b"\xff\xf3", b"\xff\xe2", b"\xff\xfa"]
for trial in range(60000):
    n = rng.choice([0, 2, 3, 4, 8, 11, 12, 40, 300, 1200, 3000])
    data = bytearray(nprng.integers(0, 256, size=n, dtype=np.uint8).tobytes())
    r = rng.random()
    if r < 0.5 and n >= 4:
        m = rng.choice(magic); data[: len(m)] = m[: len(data)]
        if rng.random() < 0.5 and n >= 12: data[8:12] = rng.choice([b"WAVE", b"WAVX", b"AVI "])
    if r < 0.35 and n >= 300:                                 # MPEG-like header chains
        hdr = bytes([0xFF, rng.choice([0xFB, 0xF3, 0xE3, 0xFD, 0xFA, 0xF2]), rng.choice([0x00, 0x10, 0x50, 0x90, 0xE0, 0x92, 0x02]), 0])
        data[:4] = hdr
        step = rng.choice([24, 104, 144, 208, 313, 417, 418, 52, 96])
        for k in range(1, 4):
            if k * step + 4 <= n and rng.random() < 0.85: data[k * step: k * step + 4] = hdr
    if 0.5 <= r < 0.6 and n >= 8: data[4:8] = b"ftyp"
    data = bytes(data)
    if bool(_ns2["_is_audio_container"](data)) != bool(is_audio_container(data)):
        bad += 1
        if bad < 6: print("SNIFF", data[:16], len(data))
print("is_audio_container mismatches:", bad)
