"""SenseVoiceSmall on gfx950 (encoder-only + CTC greedy).

Host-side mirror of `SenseVoiceSmall` (funasr/models/sense_voice/model.py:658-1080, `model_classes["SenseVoiceSmall"]`)
for the CTC-greedy inference path: same constructor keywords, state_dict layout (encoder.*, embed.weight,
ctc.ctc_lo.*), the four query frames [language, event, emotion, textnorm] placed in front of the speech features
(:971-995) and `inference(...) -> (results, meta_data)`. The [B, T, 25055] log-softmax is never materialised: the
arg-max is fused into the CTC projection GEMM (`ban_emo_unk`: a -inf bias for <|EMO_UNKNOWN|> on a sibling head, same
single launch); `output_timestamp` adds the reference's CTC forced alignment of the decoded pieces on the host (:1036-1078).
"""
from __future__ import annotations

import time
from typing import List

import numpy as np
import torch
import torch.nn as nn

from .audio import batch_to_features
from .ctc import CTC
from .hip_module import HostCopyRing, StagedUpload
from .register import tables
from . import sanm_encoder as _sanm_encoder  # noqa: F401
from . import wav_frontend as _wav_frontend  # noqa: F401


def ctc_forced_align(log_probs, targets, blank: int = 0):
    """`ctc_forced_align` (funasr/models/sense_voice/utils/ctc_alignment.py:2-77, batch of one) on the host: the Viterbi path of
    the target labels (with optional blanks between and around them) through the emissions. log_probs float [T, C], targets
    int [L] -> int64 [T]: the label (or `blank`) aligned to every frame. O(T L) scalar work on a few dozen frames -- the
    reference runs it as ~T small tensor ops; numpy float32 here, the same comparisons in the same order."""
    import numpy as np
    lp = np.asarray(log_probs, dtype=np.float32)
    tg = np.asarray(targets, dtype=np.int64)
    T, L = lp.shape[0], tg.shape[0]
    ext = np.full(2 * L + 1, blank, dtype=np.int64)              # blank, y1, blank, y2, ..., blank
    ext[1::2] = tg
    diff = np.concatenate(([False, False], ext[2:] != ext[:-2]))
    neg_inf = np.float32(-np.inf)
    PAD = 2
    best = np.full(PAD + ext.shape[0], neg_inf, dtype=np.float32)
    best[PAD + 0] = lp[0, blank]
    best[PAD + 1] = lp[0, ext[1]]
    back = np.zeros((T, PAD + ext.shape[0]), dtype=np.int64)
    for t in range(1, T):
        prev = np.stack((best[2:], best[1:-1], np.where(diff, best[:-2], neg_inf)))
        idx = prev.argmax(axis=0)                                # first maximum, like torch.max over dim 0
        best[PAD:] = lp[t, ext] + prev[idx, np.arange(prev.shape[1])]
        back[t, PAD:] = idx
    last = np.array([best[PAD + 2 * L - 1], best[PAD + 2 * L]])
    path = np.zeros(T, dtype=np.int64)
    path[T - 1] = PAD + 2 * L - 1 + int(last.argmax())
    for t in range(T - 1, 0, -1):
        path[t - 1] += path[t] - back[t, path[t]]
    return ext[np.clip(path - PAD, 0, None)]


@tables.register("model_classes", "SenseVoiceSmall")
class SenseVoiceSmall(nn.Module):
    def __init__(self, specaug: str = None, specaug_conf: dict = None, normalize: str = None,
                 normalize_conf: dict = None, encoder: str = None, encoder_conf: dict = None, ctc_conf: dict = None,
                 input_size: int = 80, vocab_size: int = -1, ignore_id: int = -1, blank_id: int = 0, sos: int = 1,
                 eos: int = 2, length_normalized_loss: bool = False, **kwargs):
        super().__init__()
        # built like the reference (sense_voice/model.py:705-707,722) -- and like there used by the training `encode` only: `inference`
        # goes from the frontend straight to the encoder (model.py:998), so no feature normalisation is applied on this path
        self.normalize = None
        if normalize is not None:
            from . import normalize as _normalize  # noqa: F401  (registers normalize_classes)
            self.normalize = tables.normalize_classes.get(normalize)(**(normalize_conf or {}))
        enc_conf = dict(encoder_conf or {})
        enc_conf.pop("input_size", None)
        self.encoder = tables.encoder_classes.get(encoder)(input_size=input_size, **enc_conf)
        self.encoder_output_size = self.encoder.output_size()
        self.ctc = CTC(odim=vocab_size, encoder_output_size=self.encoder_output_size, **(ctc_conf or {}))
        self.blank_id, self.vocab_size, self.ignore_id = blank_id, vocab_size, ignore_id
        self.sos = sos if sos is not None else vocab_size - 1
        self.eos = eos if eos is not None else vocab_size - 1
        self.lid_dict = {"auto": 0, "zh": 3, "en": 4, "yue": 7, "ja": 11, "ko": 12, "nospeech": 13}
        self.textnorm_dict = {"withitn": 14, "woitn": 15}
        self.emo_dict = {"unk": 25009, "happy": 25001, "sad": 25002, "angry": 25003, "neutral": 25004}   # model.py:738-744
        self.embed = nn.Embedding(7 + len(self.lid_dict) + len(self.textnorm_dict), input_size)
        self.embed.weight.requires_grad_(False)
        if kwargs.get("precision"):                      # model_conf: {precision: f16x2 | fp32 | bf16x3 | bf16}
            self.set_precision(kwargs["precision"])

    def set_precision(self, mode=None):
        """Arithmetic mode of the encoder (see SANMEncoder.set_precision); in "f16x2" the CTC projection with its fused
        arg-max runs from two-plane operands too, otherwise it stays on the fp32 MFMA."""
        self.encoder.set_precision(mode)
        self.ctc.set_precision(mode)
        return self

    @classmethod
    def from_config(cls, cfg: dict) -> "SenseVoiceSmall":
        ec = dict(cfg["encoder"])
        input_size = ec.pop("input_size")
        return cls(encoder="SenseVoiceEncoderSmall", encoder_conf=dict(ec, input_layer="pe"), input_size=input_size,
                   vocab_size=cfg["vocab_size"])

    def prepend_queries(self, speech: torch.Tensor, speech_lengths, language: str = "auto", textnorm: str = "woitn"):
        """[language, event, emotion, textnorm, speech...] (sense_voice/model.py:971-995); a device-side concat."""
        B = speech.shape[0]
        w = self.embed.weight.to(speech.device)
        lid = self.lid_dict[language] if language in self.lid_dict else 0
        q = torch.stack([w[lid], w[1], w[2], w[self.textnorm_dict[textnorm]]], 0)[None].expand(B, -1, -1)
        lens = torch.as_tensor(speech_lengths).to(torch.int32).cpu() + 4
        return torch.cat((q.to(speech.dtype), speech), dim=1).contiguous(), lens

    def enqueue_features(self, speech: torch.Tensor, speech_lengths, language: str = "auto", textnorm: str = "woitn",
                         return_intermediate: bool = False, ban_ids=None):
        """[B, T, 560] features -> query frames, encoder, CTC projection with the fused arg-max and the batch's one D2H copy
        (frame ids into pinned memory) ENQUEUED on the current HIP stream, no host synchronisation. `collect()` waits for that
        copy only; a serving loop enqueues batch i+1 before collecting batch i, so the GPU never waits for the host-side
        collapse of repeated frames (model.py:1013-1016)."""
        x, lens = self.prepend_queries(speech, speech_lengths, language, textnorm)
        if hasattr(self.encoder, "set_row_packing"):
            # the CTC head reads rows < len only (model.py:1014): in the f16x2 mode the padding rows are not computed at all
            self.encoder.set_row_packing(self.encoder.ALL_ROWS if return_intermediate else 0)
        enc, olens = self.encoder(x, lens)
        frame_ids = self.ctc.argmax(enc, ban_ids=ban_ids)
        pending = dict(frame_ids=self.__dict__.setdefault("_host_ring", HostCopyRing()).start(frame_ids),
                       olens=[int(v) for v in lens.tolist()])       # == olens (no subsampling), already on the host
        if return_intermediate:
            pending["extra"] = dict(enc=enc, olens=olens)
        return pending

    def collect(self, pending: dict) -> dict:
        frame_ids = HostCopyRing.wait(pending["frame_ids"]).numpy()
        olens = pending["olens"]
        ids: List[List[int]] = []
        frames: List[List[int]] = []
        blank = self.blank_id
        for b, n in enumerate(olens):
            row = frame_ids[b, :n]
            keep = np.ones(n, dtype=bool)
            keep[1:] = row[1:] != row[:-1]                           # unique_consecutive (model.py:1013-1016)
            y = row[keep]
            ids.append(y[y != blank].tolist())
            frames.append(row.tolist())
        out = dict(ids=ids, frame_ids=frames)
        out.update(pending.get("extra", {}))
        return out

    def recognize_features(self, speech: torch.Tensor, speech_lengths, language: str = "auto",
                           textnorm: str = "woitn", return_intermediate: bool = False, ban_ids=None):
        return self.collect(self.enqueue_features(speech, speech_lengths, language, textnorm, return_intermediate, ban_ids))

    @staticmethod
    def post(timestamp):
        """model.py:1080-1112: [piece, start_s, end_s] per aligned piece -> ([[start_ms, end_ms]], [word]): "▁" alone is
        dropped, a piece starting with "▁" opens a word, consecutive ASCII-alphabetic pieces are glued to the word before"""
        stamps, words, prev = [], [], None
        for i, (word, start, end) in enumerate(timestamp):
            start, end = int(start * 1000), int(end * 1000)
            if word == "▁":
                continue
            if i == 0:
                stamps.append([start, end])
                words.append(word)
            elif word.startswith("▁"):
                word = word[1:]
                stamps.append([start, end])
                words.append(word)
            elif prev is not None and prev.isalpha() and prev.isascii() and word.isalpha() and word.isascii():
                word = prev + word
                stamps[-1][1] = end
                words[-1] = word
            else:
                stamps.append([start, end])
                words.append(word)
            prev = word
        return stamps, words

    def ctc_timestamps(self, text: str, logp_speech, tokenizer):
        """The `output_timestamp` branch of the reference for one utterance (model.py:1036-1078): the decoded text is cut
        into pieces again (the four rich-tag pieces in front dropped), the pieces' ids are force-aligned to the CTC
        log-probabilities of the speech frames (frames whose arg-max is blank get blank log-probability 0 first), every run
        of a non-blank label becomes [piece, start, end] in seconds at 60 ms per frame, shifted by half a frame.
        logp_speech: float [T - 4, V] (host). -> (timestamp, words) or None when the text has no pieces."""
        from itertools import groupby
        import numpy as np
        tokens = tokenizer.text2tokens(text)[4:]
        token_ids = []
        for ids in tokenizer.tokens2ids(tokens):
            if ids:
                token_ids.extend(ids)
            else:
                token_ids.append(124)                            # the reference's stand-in for a piece without ids
        if len(token_ids) == 0:
            return None
        lp = np.array(logp_speech, dtype=np.float32, copy=True)
        pred = lp.argmax(-1)
        lp[pred == self.blank_id, self.blank_id] = 0
        tg = np.asarray(token_ids, dtype=np.int64)
        tg[tg == self.ignore_id] = self.blank_id
        align = ctc_forced_align(lp, tg, blank=self.blank_id)
        ts_max = lp.shape[0]
        timestamp, start, k = [], 0, 0
        for label, run in groupby(align.tolist()):
            end = start + len(list(run))
            if label != 0:
                timestamp.append([tokens[k], max((start * 60 - 30) / 1000, 0), min((end * 60 - 30) / 1000, (ts_max * 60 - 30) / 1000)])
                k += 1
            start = end
        return self.post(timestamp)

    # per-process staging objects (HIP streams, pinned buffers) are never copied or pickled with the module
    def __getstate__(self):
        st = dict(self.__dict__)
        for k in ("_upload", "_host_ring"):
            st.pop(k, None)
        return st

    def inference(self, data_in, data_lengths=None, key: list = ["wav_file_tmp_name"], tokenizer=None, frontend=None,
                  **kwargs):
        args, ban, meta_data, output_timestamp = self._inference_inputs(data_in, data_lengths, tokenizer, frontend, kwargs, staged=True)
        res = self.recognize_features(*args, ban_ids=ban, return_intermediate=output_timestamp)
        return self._records(res, key, tokenizer, meta_data, output_timestamp)

    # ---- the same call in parts for AutoModel.inference's loop over batches (see paraformer.py inference_begin): the encoder +
    #      CTC arg-max of batch i + 1 are enqueued before batch i's frame ids are read and turned into text
    def inference_begin(self, data_in, data_lengths=None, key: list = ["wav_file_tmp_name"], tokenizer=None, frontend=None, **kwargs):
        if (kwargs.get("output_timestamp", False) or kwargs.get("data_type", "sound") == "fbank"
                or not str(kwargs.get("device", "")).startswith("cuda")):
            return None
        args, ban, meta_data, _ = self._inference_inputs(data_in, data_lengths, tokenizer, frontend, kwargs, staged=True)
        return dict(enq=self.enqueue_features(*args, ban_ids=ban), key=key, tokenizer=tokenizer, meta_data=meta_data)

    def inference_launch(self, pending: dict) -> None:
        return None                                              # nothing waits for the host between the encoder and the ids

    def inference_end(self, pending: dict):
        return self._records(self.collect(pending.pop("enq")), pending["key"], pending["tokenizer"], pending["meta_data"], False)

    def _inference_inputs(self, data_in, data_lengths, tokenizer, frontend, kwargs, staged):
        """-> ((speech, speech_lengths, language, textnorm) and ban_ids of recognize_features / enqueue_features, meta_data, output_timestamp)"""
        output_timestamp = kwargs.get("output_timestamp", False)
        if output_timestamp and tokenizer is None:
            raise ValueError("output_timestamp needs the tokenizer (text2tokens / tokens2ids)")
        speech, speech_lengths, meta_data = batch_to_features(
            data_in, data_lengths, frontend, kwargs, uploader=self.__dict__.setdefault("_upload", StagedUpload()) if staged else None)
        use_itn = kwargs.get("use_itn", False)
        textnorm = kwargs.get("text_norm", None) or ("withitn" if use_itn else "woitn")
        ban = [self.emo_dict["unk"]] if kwargs.get("ban_emo_unk", False) else None          # model.py:1004-1005
        return (speech, speech_lengths, kwargs.get("language", "auto"), textnorm), ban, meta_data, output_timestamp

    def _records(self, res, key, tokenizer, meta_data, output_timestamp):
        B = len(res["ids"])
        logp = None
        if output_timestamp:                                     # one D2H copy of the batch's log-probabilities (:1045)
            logp = self.ctc.log_softmax(res["enc"]).cpu()
        if isinstance(key[0], (list, tuple)):
            key = key[0]
        if len(key) < B:
            key = list(key) * B
        results = []
        for i in range(B):
            if tokenizer is None:
                results.append({"key": key[i], "token_int": res["ids"][i]})
                continue
            item = {"key": key[i], "text": tokenizer.decode(res["ids"][i])}
            if output_timestamp:
                n = int(res["olens"][i])
                ts = self.ctc_timestamps(item["text"], logp[i, 4:n].numpy(), tokenizer)
                if ts is not None:
                    item["timestamp"], item["words"] = ts
            results.append(item)
        return results, meta_data
