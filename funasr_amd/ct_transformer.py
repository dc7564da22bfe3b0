"""CT-Transformer punctuation on gfx950: the network on the GPU, the sentence assembly on the host.

Host-side mirror of `CTTransformer` (`model_classes["CTTransformer"]`, funasr/models/ct_transformer/model.py:35-477), the
`punc_model` of the long-form pipeline (`AutoModel.inference_with_vad`, auto_model.py:1057-1075):
  * `punc_forward(text, text_lengths)` (:105-124): embedding lookup -> SAN-M encoder (the same encoder handle as the ASR
    path; d_model 256, 8 heads of d_k = 32 go through the small-head attention kernel) -> Linear(att_unit, n_punc);
  * `inference` (:289-477): text -> words (`split_words`, utils.py:23-83: blanks separate, every non-ASCII character is its
    own word, ASCII runs stay together) -> token ids -> mini-sentences of `split_size` words, each decoded together with
    the unfinished tail of the previous one (cut at the last predicted period / question mark, or at the last comma once
    the carried text exceeds 200 words) -> words + punctuation marks glued with the reference's rules (ASCII words are
    blank-separated and capitalised after a sentence end and take ASCII marks, the text always ends on a sentence end).
Output `[{"key", "text", "punc_array"}]` like the reference. `assemble()` holds the host logic with the network as a
callable, which is how tests/test_punctuation.py pins it to the reference (same injected predictions on both sides).
"""
from __future__ import annotations

import copy
import re
from typing import Callable, List, Sequence

import numpy as np
import torch

from . import sanm_encoder as _sanm_encoder  # noqa: F401  (registers SANMEncoder)
from .hip_module import linear
from .register import tables

_SENT_END = ("。", "？")
_ASCII_MARK = {"，": ",", "。": ".", "？": "?"}


def _ascii_first(word: str) -> bool:
    return len(word[0].encode()) == 1


def split_to_mini_sentence(words: Sequence, word_limit: int = 20) -> List[Sequence]:
    """utils.py:8-21: consecutive blocks of `word_limit` words, the remainder as a last shorter block"""
    assert word_limit > 1
    if len(words) <= word_limit:
        return [words]
    return [words[i: i + word_limit] for i in range(0, len(words), word_limit)]


def _is_english(token: str) -> bool:
    return re.search("^[a-zA-Z']+$", token) is not None


def split_words(text: str, jieba_usr_dict=None, **kwargs) -> List[str]:
    if jieba_usr_dict:
        # word-level models: runs of one language are grouped, the Chinese runs are cut by jieba (utils.py:29-62)
        groups, cur, lang = [], [], None
        for tok in text.split():
            now = "English" if _is_english(tok) else "Chinese"
            if lang is not None and now != lang:
                groups.append((lang, cur))
                cur = []
            cur.append(tok)
            lang = now
        if cur:
            groups.append((lang, cur))
        out: List[str] = []
        for lang, toks in groups:
            if lang == "English":
                out.extend(toks)
            else:
                line = ""
                for t in toks:
                    line = line + (" " + t if _is_english(t) else t)
                out.extend(jieba_usr_dict.cut(line.strip(), HMM=False))
        return out
    words: List[str] = []
    for seg in text.split():
        run = ""
        for ch in seg:
            if len(ch.encode()) == 1:
                run += ch
            else:
                if run:
                    words.append(run)
                    run = ""
                words.append(ch)
        if run:
            words.append(run)
    return words


def assemble(tokens: List[str], token_ids: Sequence[int], predict: Callable[[np.ndarray], np.ndarray], punc_list: List[str],
             sentence_end_id: int, split_size: int = 20, carry_limit: int = 200):
    """-> (punctuated text, punc ids of the whole text as int64 array). `predict(ids[int32, L]) -> punc ids[L]` is the
    network (arg-max of punc_forward). Mirrors the loop of CTTransformer.inference (model.py:325-457)."""
    minis = split_to_mini_sentence(tokens, split_size)
    minis_id = split_to_mini_sentence(np.asarray(token_ids), split_size)
    assert len(minis) == len(minis_id)
    carry: List[str] = []
    carry_id = np.array([], dtype="int32")
    text, marks = "", []
    out_text, all_marks = "", None
    last = len(minis) - 1
    for mi, (sent, ids) in enumerate(zip(minis, minis_id)):
        sent = carry + list(sent)
        ids = np.concatenate((carry_id, ids), axis=0)
        puncs = np.array(predict(ids), dtype=np.int64)
        assert puncs.shape[0] == len(sent)
        if mi < last:
            end, comma = -1, -1
            for i in range(len(puncs) - 2, 1, -1):
                if punc_list[puncs[i]] in _SENT_END:
                    end = i
                    break
                if comma < 0 and punc_list[puncs[i]] == "，":
                    comma = i
            if end < 0 and len(sent) > carry_limit and comma >= 0:
                end = comma                                      # too long without a sentence end: cut at a comma
                puncs[end] = sentence_end_id
            carry, carry_id = sent[end + 1:], ids[end + 1:]
            sent, puncs = sent[: end + 1], puncs[: end + 1]
        marks += [int(x) for x in puncs]
        pieces = []
        for i in range(len(sent)):
            if (i == 0 or punc_list[puncs[i - 1]] in _SENT_END) and _ascii_first(sent[i]):
                sent[i] = sent[i].capitalize()
            if i == 0:
                if _ascii_first(sent[i]):
                    sent[i] = " " + sent[i]
            elif _ascii_first(sent[i]) and _ascii_first(sent[i - 1]):
                sent[i] = " " + sent[i]
            pieces.append(sent[i])
            mark = punc_list[puncs[i]]
            if mark != "_":
                pieces.append(_ASCII_MARK.get(mark, mark) if _ascii_first(sent[i]) else mark)
        text += "".join(pieces)
        out_text = text
        if mi == last:                                            # the text always ends on a sentence end
            tail = text[-1]
            close = None
            if tail in ("，", "、"):
                close = text[:-1] + "。"
            elif tail == ",":
                close = text[:-1] + "."
            elif tail not in _SENT_END and len(tail.encode()) != 1:
                close = text + "。"
            elif tail not in (".", "?") and len(tail.encode()) == 1:
                close = text + "."
            if close is not None:
                out_text = close
                if len(puncs):
                    puncs[-1] = sentence_end_id
        all_marks = puncs if all_marks is None else np.concatenate([all_marks, puncs], axis=0)
    return out_text, all_marks


@tables.register("model_classes", "CTTransformer")
class CTTransformer(torch.nn.Module):
    def __init__(self, encoder: str = None, encoder_conf: dict = None, vocab_size: int = -1, punc_list: list = None,
                 punc_weight: list = None, embed_unit: int = 128, att_unit: int = 256, dropout_rate: float = 0.5,
                 ignore_id: int = -1, sos: int = 1, eos: int = 2, sentence_end_id: int = 3, **kwargs):
        super().__init__()
        enc_cls = tables.encoder_classes.get(encoder)
        if enc_cls is None:
            raise KeyError(f"encoder {encoder!r} is not registered: {sorted(tables.encoder_classes)}")
        self.embed = torch.nn.Embedding(vocab_size, embed_unit)
        self.embed.weight.requires_grad_(False)
        self.encoder = enc_cls(**(encoder_conf or {}))
        self.decoder = linear(len(punc_list), att_unit)
        self.punc_list, self.punc_weight = list(punc_list), punc_weight or [1] * len(punc_list)
        self.ignore_id, self.sos, self.eos, self.sentence_end_id = ignore_id, sos, eos, sentence_end_id
        self.jieba_usr_dict = None
        if kwargs.get("jieba_usr_dict") is not None:
            import jieba
            jieba.load_userdict(kwargs["jieba_usr_dict"])
            self.jieba_usr_dict = jieba

    def with_vad(self) -> bool:
        return False

    def punc_forward(self, text: torch.Tensor, text_lengths: torch.Tensor, **kwargs):
        """text int [B, L] -> (logits [B, L, n_punc], None)"""
        from . import ops
        dev = self.embed.weight.device
        if dev.type != "cuda":
            raise RuntimeError("CTTransformer runs only on an AMD GPU through libparaformer_hip.so (no CPU fallback)")
        ids = text.to(device=dev, dtype=torch.int32).contiguous()
        x = ops.gather_rows(self.embed.weight.detach().to(torch.float32), ids.view(-1)).view(ids.shape[0], ids.shape[1], -1)
        h, _, _ = self.encoder(x, text_lengths)
        y = ops.gemm(h.reshape(-1, h.shape[-1]).contiguous(), self.decoder.weight.detach().float().contiguous(),
                     self.decoder.bias.detach().float().contiguous())
        return y.view(h.shape[0], h.shape[1], -1), None

    def inference(self, data_in, data_lengths=None, key: list = None, tokenizer=None, frontend=None, **kwargs):
        assert len(data_in) == 1
        if not data_in[0] or (isinstance(data_in[0], str) and not data_in[0].strip()):
            return [{"key": key[0] if key else "", "text": "", "punc_array": None}], {"batch_data_time": -1}
        text = data_in[0]
        tokens = split_words(text, jieba_usr_dict=self.jieba_usr_dict)
        token_ids = tokenizer.encode(tokens)

        def predict(ids: np.ndarray) -> np.ndarray:
            t = torch.from_numpy(np.ascontiguousarray(ids, dtype=np.int64))[None]
            y, _ = self.punc_forward(t, torch.tensor([t.shape[1]], dtype=torch.int32))
            return y.view(-1, y.shape[-1]).argmax(dim=1).cpu().numpy()

        out_text, marks = assemble(tokens, token_ids, predict, self.punc_list, self.sentence_end_id,
                                   split_size=kwargs.get("split_size", 20))
        punc_array = torch.from_numpy(np.asarray(marks, dtype=np.int64))
        if self.jieba_usr_dict is not None:                      # word-level model: one mark per character (:459-471)
            flat = copy.copy(punc_array.reshape(-1)).tolist()
            n = len(tokens)
            for i, tok in enumerate(tokens[::-1]):
                if "฀" <= tok[0] <= "龥" and len(tok) > 1:
                    for _ in range(len(tok) - 1):
                        flat.insert(n - i - 1, 1)
            punc_array = torch.tensor(flat)
        return [{"key": key[0], "text": out_text, "punc_array": punc_array}], {}

    def forward(self, *a, **k):  # pragma: no cover
        raise NotImplementedError("training forward() is out of scope; use inference() / punc_forward()")


# ------------------------------------------------------------------------------------------- streaming (realtime) model
def assemble_streaming(text: str, cache: dict, encode: Callable[[List[str]], Sequence[int]],
                       predict: Callable[[np.ndarray, int], np.ndarray], punc_list: List[str], sentence_end_id: int,
                       split_size: int = 20, carry_limit: int = 200):
    """One call of CTTransformerStreaming.inference (ct_transformer_streaming/model.py:78-219) with the network as a
    callable `predict(ids[int32, L], vad_pos) -> punc ids[L]`: the words carried over from the previous calls (`cache
    ["pre_text"]`, everything after the last sentence end) are put in front of the new text, the whole is decoded in
    mini-sentences like the offline model, and only what belongs to the new text is returned -- without a trailing mark,
    which the next call may still revise. -> (text, punc ids of the LAST mini-sentence as the reference returns them)."""
    if len(cache) == 0:
        cache["pre_text"] = []
    pre = cache["pre_text"]
    text = "".join(pre) + " " + text
    tokens = split_words(text)
    token_ids = np.asarray(encode(tokens))
    minis = split_to_mini_sentence(tokens, split_size)
    minis_id = split_to_mini_sentence(token_ids, split_size)
    assert len(minis) == len(minis_id)
    carry: List[str] = []
    carry_id = np.array([], dtype="int32")
    marks_all: List[str] = []
    words_all: List[str] = []
    puncs = np.array([], dtype=np.int64)
    last = len(minis) - 1
    for mi in range(len(minis)):
        sent = carry + list(minis[mi])
        ids = np.concatenate((carry_id, minis_id[mi]), axis=0)
        puncs = np.array(predict(ids, len(pre)), dtype=np.int64).reshape(-1)
        assert puncs.shape[0] == len(sent)
        if mi < last:
            end, comma = -1, -1
            for i in range(len(puncs) - 2, 1, -1):
                if punc_list[puncs[i]] in _SENT_END:
                    end = i
                    break
                if comma < 0 and punc_list[puncs[i]] == "，":
                    comma = i
            if end < 0 and len(sent) > carry_limit and comma >= 0:
                end = comma                                      # too long without a sentence end: cut at a comma
                puncs[end] = sentence_end_id
            carry, carry_id = sent[end + 1:], ids[end + 1:]
            sent, puncs = sent[: end + 1], puncs[: end + 1]
        marks_all += [punc_list[int(x)] for x in puncs]
        words_all += sent
    assert len(marks_all) == len(words_all)
    pieces, marks_out, skipped = [], [], 0
    for i in range(len(words_all)):
        if i > 0 and len(words_all[i][0].encode()) == 1 and len(words_all[i - 1][-1].encode()) == 1:
            words_all[i] = " " + words_all[i]
        if skipped < len(pre):                                   # the carried words were returned by an earlier call
            skipped += 1
        else:
            pieces.append(words_all[i])
        if skipped >= len(pre):
            marks_out.append(marks_all[i])
            if marks_all[i] != "_":
                pieces.append(marks_all[i])
    out = "".join(pieces)
    end = -1
    for i in range(len(marks_all) - 2, 1, -1):
        if marks_all[i] in _SENT_END:
            end = i
            break
    cache["pre_text"] = words_all[end + 1:]
    if out and out[-1] in punc_list:                             # (the reference indexes out[-1] unguarded: IndexError on "")
        out = out[:-1]
        marks_out[-1] = "_"
    return out, puncs


@tables.register("model_classes", "CTTransformerStreaming")
class CTTransformerStreaming(CTTransformer):
    """`CTTransformerStreaming` (funasr/models/ct_transformer_streaming/model.py:33-219): the realtime punctuation model of
    the streaming pipelines. Same parameters as CTTransformer over a `SANMVadEncoder` (causal self-attention, VAD corner in
    the last block: sanm_encoder.py); `inference(..., cache=<dict kept by the caller>)` carries the unfinished sentence
    from call to call."""

    def with_vad(self) -> bool:
        return True

    def punc_forward(self, text: torch.Tensor, text_lengths: torch.Tensor, vad_indexes: torch.Tensor = None, **kwargs):
        """text int [B, L], vad_indexes int [B] -> (logits [B, L, n_punc], None)   (model.py:59-72)"""
        from . import ops
        dev = self.embed.weight.device
        if dev.type != "cuda":
            raise RuntimeError("CTTransformerStreaming runs only on an AMD GPU through libparaformer_hip.so (no CPU fallback)")
        ids = text.to(device=dev, dtype=torch.int32).contiguous()
        x = ops.gather_rows(self.embed.weight.detach().to(torch.float32), ids.view(-1)).view(ids.shape[0], ids.shape[1], -1)
        h, _, _ = self.encoder(x, text_lengths, vad_indexes=vad_indexes)
        y = ops.gemm(h.reshape(-1, h.shape[-1]).contiguous(), self.decoder.weight.detach().float().contiguous(),
                     self.decoder.bias.detach().float().contiguous())
        return y.view(h.shape[0], h.shape[1], -1), None

    def inference(self, data_in, data_lengths=None, key: list = None, tokenizer=None, frontend=None, cache: dict = None,
                  **kwargs):
        if cache is None:
            cache = {}
        assert len(data_in) == 1

        def predict(ids: np.ndarray, vad_pos: int) -> np.ndarray:
            t = torch.from_numpy(np.ascontiguousarray(ids, dtype=np.int64))[None]
            y, _ = self.punc_forward(t, torch.tensor([t.shape[1]], dtype=torch.int32), torch.tensor([vad_pos], dtype=torch.int32))
            return y.view(-1, y.shape[-1]).argmax(dim=1).cpu().numpy()

        out, marks = assemble_streaming(data_in[0], cache, tokenizer.encode, predict, self.punc_list, self.sentence_end_id,
                                        split_size=kwargs.get("split_size", 20))
        punc_array = torch.from_numpy(np.asarray(marks, dtype=np.int64))
        if punc_array.numel() == 1:
            punc_array = punc_array.view(1, 1)                   # the reference does not squeeze a single prediction (:140-141)
        return [{"key": key[0], "text": out, "punc_array": punc_array}], {}

