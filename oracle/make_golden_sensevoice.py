"""Golden vectors for `SenseVoiceSmall.inference`, made by the REFERENCE's own class (build container only; TEST
INFRASTRUCTURE). Writes tests/golden/sensevoice_inference.npz.

The reference `SenseVoiceSmall` (funasr/models/sense_voice/model.py:658-1080) is built from a small configuration of the
same architecture (3 + 1 SAN-M blocks, vocabulary 211) with seeded weights and `inference()` is called with LFR features
(`data_type="fbank"`, :939-945) for every language / text-norm query combination the glue code distinguishes
(:971-995: language id lookup incl. the unknown-language fallback, use_itn / text_norm, the fixed event + emotion
queries), on a ragged batch. Stored: features, lengths, per case the decoded strings of a stand-in tokenizer (ids joined
by spaces -- i.e. the token ids after arg-max, unique_consecutive and blank removal, :1006-1020).

    python oracle/make_golden_sensevoice.py
"""
import json
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
from funasr_amd import synth  # noqa: E402
from oracle import paraformer_oracle as O  # noqa: E402
from oracle import ref_import  # noqa: E402


class IdTokenizer:
    """decode(ids) -> "id id id": keeps the integer result visible through the reference's text-only return value"""

    def decode(self, ids):
        return " ".join(str(int(i)) for i in ids)


CASES = [dict(language="auto"), dict(language="zh", use_itn=True), dict(language="en", text_norm="woitn"),
         dict(language="ko", text_norm="withitn"), dict(language="klingon"), dict(language="nospeech", use_itn=False)]


def main():
    ref_import.install()
    from funasr.models.sense_voice.model import SenseVoiceSmall
    cfg = synth.tiny(synth.SENSEVOICE_SMALL, enc_blocks=3, tp_blocks=1, vocab=211)
    sd = synth.sensevoice_state_dict(cfg, seed=21)
    ec = dict(cfg["encoder"])
    input_size = ec.pop("input_size")
    model = SenseVoiceSmall(encoder="SenseVoiceEncoderSmall", encoder_conf=ec, input_size=input_size,
                            vocab_size=cfg["vocab_size"]).eval()
    missing, unexpected = model.load_state_dict(sd, strict=False)
    assert not unexpected and all(k.startswith("criterion") for k in missing), (missing, unexpected)
    g = torch.Generator().manual_seed(77)
    B, T = 4, 57
    lens = torch.tensor([57, 40, 23, 9], dtype=torch.int32)
    # slowly varying features (every value held for 3 frames, a little noise on top) and a CTC bias towards blank: the
    # frame-wise arg-max then really contains repeats and blanks for unique_consecutive / the blank mask to remove
    feats = (torch.randn(B, (T + 2) // 3, 560, generator=g) * 0.8).repeat_interleave(3, dim=1)[:, :T]
    feats = feats + 0.03 * torch.randn(B, T, 560, generator=g)
    sd["ctc.ctc_lo.bias"][0] += 1.5
    model.load_state_dict(sd, strict=False)
    for b in range(B):
        feats[b, lens[b]:] = 0
    out = dict(config=json.dumps(cfg), seed=21, ctc_blank_bias_add=1.5, feats=feats.numpy(), lens=lens.numpy(), cases=json.dumps(CASES))
    tok = IdTokenizer()
    for ci, kw in enumerate(CASES):
        with torch.no_grad():
            res, _ = model.inference(feats.clone(), data_lengths=lens.clone().long(), key=[f"utt{i}" for i in range(B)],
                                     tokenizer=tok, frontend=None, device="cpu", data_type="fbank", **kw)
        texts = [r["text"] for r in res]
        keys = [r["key"] for r in res]
        # the oracle restates the same glue: pin it here
        lid = {"auto": 0, "zh": 3, "en": 4, "yue": 7, "ja": 11, "ko": 12, "nospeech": 13}.get(kw.get("language", "auto"), 0)
        tn = kw.get("text_norm") or ("withitn" if kw.get("use_itn", False) else "woitn")
        with torch.no_grad():
            oref = O.sensevoice_greedy(feats, lens, sd, cfg, language_id=lid, textnorm_id={"withitn": 14, "woitn": 15}[tn])
        assert [" ".join(str(i) for i in ids) for ids in oref["ids"]] == texts, (ci, texts, oref["ids"])
        out[f"texts_{ci}"] = json.dumps(texts)
        out[f"keys_{ci}"] = json.dumps(keys)
        print(f"case {ci} {kw}: tokens per clip {[len(t.split()) for t in texts]}")
    path = os.path.join(os.path.dirname(HERE), "tests", "golden", "sensevoice_inference.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()
