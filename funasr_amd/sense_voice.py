"""SenseVoiceSmall on gfx950 (encoder-only + CTC greedy).

Host-side mirror of `SenseVoiceSmall` (funasr/models/sense_voice/model.py:658-1080, `model_classes["SenseVoiceSmall"]`)
for the CTC-greedy inference path: same constructor keywords, state_dict layout (encoder.*, embed.weight,
ctc.ctc_lo.*), the four query frames [language, event, emotion, textnorm] placed in front of the speech features
(:971-995) and `inference(...) -> (results, meta_data)`. The [B, T, 25055] log-softmax is never materialised: the
arg-max is fused into the CTC projection GEMM (`ban_emo_unk`: a -inf bias for <|EMO_UNKNOWN|> on a sibling head, same
single launch); `output_timestamp` (CTC forced alignment, out of scope per SURVEY 2) raises.
"""
from __future__ import annotations

import time
from typing import List

import torch
import torch.nn as nn

from .audio import load_audio_list
from .ctc import CTC
from .register import tables
from . import sanm_encoder as _sanm_encoder  # noqa: F401
from . import wav_frontend as _wav_frontend  # noqa: F401


@tables.register("model_classes", "SenseVoiceSmall")
class SenseVoiceSmall(nn.Module):
    def __init__(self, specaug: str = None, specaug_conf: dict = None, normalize: str = None,
                 normalize_conf: dict = None, encoder: str = None, encoder_conf: dict = None, ctc_conf: dict = None,
                 input_size: int = 80, vocab_size: int = -1, ignore_id: int = -1, blank_id: int = 0, sos: int = 1,
                 eos: int = 2, length_normalized_loss: bool = False, **kwargs):
        super().__init__()
        if normalize is not None:
            raise NotImplementedError("SenseVoiceSmall(HIP): `normalize` is not part of the published recipe")
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
        if kwargs.get("precision"):                      # model_conf: {precision: fp32 | bf16x3 | bf16}
            self.set_precision(kwargs["precision"])

    def set_precision(self, mode: str = "fp32"):
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

    def recognize_features(self, speech: torch.Tensor, speech_lengths, language: str = "auto",
                           textnorm: str = "woitn", return_intermediate: bool = False, ban_ids=None):
        x, lens = self.prepend_queries(speech, speech_lengths, language, textnorm)
        if hasattr(self.encoder, "set_row_packing"):
            # the CTC head reads rows < len only (model.py:1014): in the f16x2 mode the padding rows are not computed at all
            self.encoder.set_row_packing(self.encoder.ALL_ROWS if return_intermediate else 0)
        enc, olens = self.encoder(x, lens)
        frame_ids = self.ctc.argmax(enc, ban_ids=ban_ids).cpu()     # one D2H copy for the batch
        ids: List[List[int]] = []
        for b in range(enc.shape[0]):
            y = torch.unique_consecutive(frame_ids[b, : int(olens[b])], dim=-1)     # model.py:1013-1016
            ids.append([int(t) for t in y.tolist() if t != self.blank_id])
        out = dict(ids=ids, frame_ids=[frame_ids[b, : int(olens[b])].tolist() for b in range(enc.shape[0])])
        if return_intermediate:
            out.update(enc=enc, olens=olens)
        return out

    def inference(self, data_in, data_lengths=None, key: list = ["wav_file_tmp_name"], tokenizer=None, frontend=None,
                  **kwargs):
        if kwargs.get("output_timestamp", False):
            raise NotImplementedError("CTC forced-alignment timestamps are outside the hot path (SURVEY 2)")
        meta_data = {}
        device = kwargs.get("device", None)
        if isinstance(data_in, torch.Tensor) and kwargs.get("data_type", "sound") == "fbank":
            speech, speech_lengths = data_in, data_lengths
            if speech.dim() < 3:
                speech = speech[None]
            if speech_lengths is None:
                speech_lengths = [speech.shape[1]] * speech.shape[0]
        else:
            t1 = time.perf_counter()
            audio = load_audio_list(data_in, fs=frontend.fs, audio_fs=kwargs.get("fs", 16000))
            t2 = time.perf_counter()
            meta_data["load_data"] = f"{t2 - t1:0.3f}"
            lens = [int(a.shape[0]) for a in audio]
            wav = torch.nn.utils.rnn.pad_sequence(audio, batch_first=True)
            if device is not None:
                wav = wav.to(device)
            speech, speech_lengths = frontend(wav, lens)
            t3 = time.perf_counter()
            meta_data["extract_feat"] = f"{t3 - t2:0.3f}"
            meta_data["batch_data_time"] = int(speech_lengths.sum().item()) * frontend.frame_shift * frontend.lfr_n / 1000
        use_itn = kwargs.get("use_itn", False)
        textnorm = kwargs.get("text_norm", None) or ("withitn" if use_itn else "woitn")
        ban = [self.emo_dict["unk"]] if kwargs.get("ban_emo_unk", False) else None          # model.py:1004-1005
        res = self.recognize_features(speech, speech_lengths, kwargs.get("language", "auto"), textnorm, ban_ids=ban)
        B = len(res["ids"])
        if isinstance(key[0], (list, tuple)):
            key = key[0]
        if len(key) < B:
            key = list(key) * B
        results = []
        for i in range(B):
            if tokenizer is not None:
                results.append({"key": key[i], "text": tokenizer.decode(res["ids"][i])})
            else:
                results.append({"key": key[i], "token_int": res["ids"][i]})
        return results, meta_data
