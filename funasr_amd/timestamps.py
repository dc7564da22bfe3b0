"""Token timestamps from the CIF weights (host side, no GPU work).

Restates `ts_prediction_lfr6_standard` (funasr/utils/timestamp_tools.py:37-122), the function Paraformer.inference
calls when `pred_timestamp=True` (funasr/models/paraformer/model.py:668-676): a token lasts from its CIF fire to the
next one, fires are shifted by `force_time_shift` frames, one (upsampled) LFR frame is 60 ms / upsample_rate, long gaps
and the two ends of the utterance become `<sil>` spans. When the number of fires does not match the number of tokens
the alphas are rescaled to sum to len(tokens) + 1 and integrated again sequentially in fp32
(`cif_wo_hidden`, timestamp_tools.py:12-34) -- the arithmetic below keeps the reference's dtypes and order so that the
millisecond values come out identical.
"""
from __future__ import annotations

from typing import List, Sequence, Tuple

import numpy as np
import torch

_EDGE_FRAMES = 5          # shorter leading / trailing gaps are not reported as silence
_MAX_TOKEN_FRAMES = 12    # a token is cut after this many frames, the rest of the gap is silence
_FIRE = 1.0 - 1e-4


def _refire(alphas: torch.Tensor, n_fires: int) -> torch.Tensor:
    """alphas rescaled to sum to n_fires, then integrate-and-fire at 1 - 1e-4; returns the running integral per frame
    (the value BEFORE the threshold is taken off), fp32 like the reference loop."""
    a = (alphas / (alphas.sum() / n_fires)).to(torch.float32).cpu().numpy()
    thr = np.float32(_FIRE)
    acc = np.float32(0.0)
    out = np.empty(a.shape[0], dtype=np.float32)
    for t in range(a.shape[0]):
        acc = np.float32(acc + a[t])
        out[t] = acc
        if acc >= thr:
            acc = np.float32(acc - thr)
    return torch.from_numpy(out)


def cif_timestamps(us_alphas: torch.Tensor, us_peaks: torch.Tensor, char_list: Sequence[str], vad_offset: float = 0.0,
                   force_time_shift: float = -1.5, sil_in_str: bool = True, upsample_rate: int = 3
                   ) -> Tuple[str, List[List[int]]]:
    """-> ("tok beg end;..." with seconds, [[beg_ms, end_ms] per non-silence token])."""
    if not len(char_list):
        return "", []
    frame_s = 10.0 * 6 / 1000 / upsample_rate
    alphas, peaks = (us_alphas[0], us_peaks[0]) if us_alphas.dim() == 2 else (us_alphas, us_peaks)   # batch of one
    tokens = list(char_list[:-1]) if char_list[-1] == "</s>" else list(char_list)
    fires = torch.where(peaks >= _FIRE)[0].cpu().numpy() + force_time_shift
    if len(fires) != len(tokens) + 1:
        peaks = _refire(alphas, len(tokens) + 1)
        fires = torch.where(peaks >= _FIRE)[0].cpu().numpy() + force_time_shift
    n_frames = peaks.shape[0]

    names: List[str] = []
    spans: List[List[float]] = []
    if fires[0] > _EDGE_FRAMES:
        names.append("<sil>")
        spans.append([0.0, fires[0] * frame_s])
    for i in range(len(fires) - 1):
        names.append(tokens[i])
        if fires[i + 1] - fires[i] <= _MAX_TOKEN_FRAMES:
            spans.append([fires[i] * frame_s, fires[i + 1] * frame_s])
        else:
            cut = fires[i] + _MAX_TOKEN_FRAMES
            spans.append([fires[i] * frame_s, cut * frame_s])
            names.append("<sil>")
            spans.append([cut * frame_s, fires[i + 1] * frame_s])
    if n_frames - fires[-1] > _EDGE_FRAMES:
        end = (n_frames + fires[-1]) * 0.5
        spans[-1][1] = end * frame_s
        names.append("<sil>")
        spans.append([end * frame_s, n_frames * frame_s])
    elif spans:
        spans[-1][1] = n_frames * frame_s
    if vad_offset:                                        # segment start inside the recording, milliseconds
        for sp in spans:
            sp[0] += vad_offset / 1000.0
            sp[1] += vad_offset / 1000.0
    text = "".join("{} {} {};".format(nm, str(sp[0] + 0.0005)[:5], str(sp[1] + 0.0005)[:5])
                   for nm, sp in zip(names, spans) if sil_in_str or nm != "<sil>")
    ms = [[int(sp[0] * 1000), int(sp[1] * 1000)] for nm, sp in zip(names, spans) if nm != "<sil>"]
    return text, ms


ts_prediction_lfr6_standard = cif_timestamps       # the reference's name, for code that imports it


def cif_token_spans(us_alphas, us_peaks, char_list: Sequence[str], vad_offset: float = 0.0, force_time_shift: float = -1.5,
                    upsample_rate: int = 3) -> List[List[int]]:
    """The second result of `cif_timestamps` ([[beg_ms, end_ms] per token]) without building the name / text lists: the
    same float64 arithmetic, vectorised over the fires with numpy (a 30 s utterance: ~40 us instead of ~600 us of per-token
    Python; the model classes call this once per utterance on the host while the GPU runs the next batch). Falls back to
    `cif_timestamps` for the rare shapes whose bookkeeping is not worth vectorising (no tokens, fire count mismatch)."""
    if not len(char_list):
        return []
    n_tok = len(char_list) - 1 if char_list[-1] == "</s>" else len(char_list)
    peaks = us_peaks[0] if getattr(us_peaks, "ndim", 1) == 2 else us_peaks
    p = peaks.detach().cpu().numpy() if isinstance(peaks, torch.Tensor) else np.asarray(peaks)
    fires = np.nonzero(p >= np.float32(_FIRE))[0] + force_time_shift
    if n_tok > 0 and len(fires) != n_tok + 1:
        # the reference's repair (timestamp_tools.py:60-63): alphas rescaled to n_tok + 1 fires and integrated again, once
        alphas = us_alphas[0] if getattr(us_alphas, "ndim", 1) == 2 else us_alphas
        p = _refire(alphas if isinstance(alphas, torch.Tensor) else torch.as_tensor(np.asarray(alphas)), n_tok + 1).numpy()
        fires = np.nonzero(p >= np.float32(_FIRE))[0] + force_time_shift
    if n_tok == 0 or len(fires) != n_tok + 1:
        return cif_timestamps(us_alphas, us_peaks, char_list, vad_offset, force_time_shift, True, upsample_rate)[1]
    frame_s = 10.0 * 6 / 1000 / upsample_rate
    n_frames = p.shape[0]
    beg = fires[:-1] * frame_s
    long_gap = (fires[1:] - fires[:-1]) > _MAX_TOKEN_FRAMES
    end = np.where(long_gap, (fires[:-1] + _MAX_TOKEN_FRAMES) * frame_s, fires[1:] * frame_s)
    if not long_gap[-1]:                       # otherwise the utterance ends on a <sil> span and that one is stretched
        if n_frames - fires[-1] > _EDGE_FRAMES:
            end[-1] = ((n_frames + fires[-1]) * 0.5) * frame_s
        else:
            end[-1] = n_frames * frame_s
    if vad_offset:
        beg = beg + vad_offset / 1000.0
        end = end + vad_offset / 1000.0
    return np.stack([(beg * 1000).astype(np.int64), (end * 1000).astype(np.int64)], axis=1).tolist()


def _ascii_letter(ch: str) -> bool:
    return "a" <= ch <= "z" or "A" <= ch <= "Z"


def timestamp_sentence(punc_id_list, timestamp_postprocessed, text_postprocessed, return_raw_text: bool = False,
                       english: bool = False) -> List[dict]:
    """Cut a punctuated recording into sentence records {text, start, end, timestamp[, raw_text]} at every punctuation
    mark (ids > 1 of the CT-Transformer's punc_array). Restates `timestamp_sentence` / `timestamp_sentence_en`
    (funasr/utils/timestamp_tools.py:125-222 / :223-316): words are glued without blanks except around ASCII words, marks
    are `，。？、` (`,.?,` for English); the English variant strips the leading blank and restarts the sentence clock on
    the first word after a mark, the Chinese one on the first word that has a timestamp."""
    from itertools import zip_longest
    marks = [",", ".", "?", ","] if english else ["，", "。", "？", "、"]
    res: List[dict] = []
    if text_postprocessed is None or timestamp_postprocessed is None:
        return res
    if len(timestamp_postprocessed) == 0 or len(text_postprocessed) == 0:
        return res
    if punc_id_list is None or len(punc_id_list) == 0:
        return [{"text": text_postprocessed.split(), "start": timestamp_postprocessed[0][0],
                 "end": timestamp_postprocessed[-1][1], "timestamp": timestamp_postprocessed}]
    if hasattr(punc_id_list, "tolist"):       # a tensor / array: iterate plain ints (tensor iteration makes one 0-d tensor per id)
        punc_id_list = punc_id_list.tolist()
    sent, raw, stamps = "", "", []
    start, end = timestamp_postprocessed[0][0], timestamp_postprocessed[0][1]
    fresh = True
    for punc_id, stamp, word in zip_longest(punc_id_list, timestamp_postprocessed, text_postprocessed.split(), fillvalue=None):
        if not english and start is None and stamp is not None:
            start = stamp[0]
        if word is not None:
            if _ascii_letter(word[0]) or (len(sent) and _ascii_letter(sent[-1])):
                sent += " " + word
            else:
                sent += word
            raw += word + " "
        stamps.append(stamp)
        punc_id = int(punc_id) if punc_id is not None else 1
        end = stamp[1] if stamp is not None else end
        if english:
            sent = sent[1:] if sent[0] == " " else sent
            if fresh:
                start = stamp[0] if stamp is not None else start
                fresh = False
        else:
            raw = raw[:-1] if raw and raw[-1] == " " else raw
        if punc_id > 1:
            sent += marks[punc_id - 2]
            if english:
                fresh = True
                raw = raw[:-1] if raw[-1] == " " else raw
            rec = {"text": sent, "start": start, "end": end, "timestamp": stamps}
            if return_raw_text:
                rec["raw_text"] = raw
            res.append(rec)
            sent, raw, stamps = "", "", []
            if not english:
                start = None
    return res
