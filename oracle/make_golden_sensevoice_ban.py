"""Golden vectors for `SenseVoiceSmall.inference(..., ban_emo_unk=True)` made by the REFERENCE's own class (build container
only; TEST INFRASTRUCTURE). Writes tests/golden/sensevoice_ban.npz.

`ban_emo_unk` sets the CTC log-probability of the emotion token <|EMO_UNKNOWN|> -- id 25009, hard-coded in `emo_dict`
(funasr/models/sense_voice/model.py:738-744) -- to -inf before the frame-wise arg-max (:1004-1005), so the case needs the
real vocabulary size (25055) on a small encoder (3 + 1 blocks). The CTC bias of that id is raised so that it really wins a
good share of the frames without the ban.

    python oracle/make_golden_sensevoice_ban.py
"""
import json
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
from funasr_amd import synth  # noqa: E402
from oracle import ref_import  # noqa: E402
from oracle.make_golden_sensevoice import IdTokenizer  # noqa: E402

EMO_UNK = 25009
BIAS_ADD = {0: 1.0, EMO_UNK: 3.0}


def main():
    ref_import.install()
    from funasr.models.sense_voice.model import SenseVoiceSmall
    cfg = synth.tiny(synth.SENSEVOICE_SMALL, enc_blocks=3, tp_blocks=1, vocab=25055)
    sd = synth.sensevoice_state_dict(cfg, seed=23)
    for k, v in BIAS_ADD.items():
        sd["ctc.ctc_lo.bias"][k] += v
    ec = dict(cfg["encoder"])
    input_size = ec.pop("input_size")
    model = SenseVoiceSmall(encoder="SenseVoiceEncoderSmall", encoder_conf=ec, input_size=input_size,
                            vocab_size=cfg["vocab_size"]).eval()
    assert model.emo_dict["unk"] == EMO_UNK
    missing, unexpected = model.load_state_dict(sd, strict=False)
    assert not unexpected and all(k.startswith("criterion") for k in missing), (missing, unexpected)
    g = torch.Generator().manual_seed(5)
    B, T = 3, 41
    lens = torch.tensor([41, 30, 12], dtype=torch.int32)
    feats = (torch.randn(B, (T + 2) // 3, 560, generator=g) * 0.8).repeat_interleave(3, dim=1)[:, :T]
    feats = feats + 0.03 * torch.randn(B, T, 560, generator=g)
    for b in range(B):
        feats[b, lens[b]:] = 0
    out = dict(config=json.dumps(cfg), seed=23, bias_add=json.dumps({str(k): v for k, v in BIAS_ADD.items()}),
               feats=feats.numpy(), lens=lens.numpy())
    tok = IdTokenizer()
    for name, kw in (("plain", {}), ("ban", dict(ban_emo_unk=True))):
        with torch.no_grad():
            res, _ = model.inference(feats.clone(), data_lengths=lens.clone().long(), key=[f"utt{i}" for i in range(B)],
                                     tokenizer=tok, frontend=None, device="cpu", data_type="fbank", language="auto", **kw)
        out[f"texts_{name}"] = json.dumps([r["text"] for r in res])
    plain, ban = json.loads(out["texts_plain"]), json.loads(out["texts_ban"])
    n_unk = sum(t.split().count(str(EMO_UNK)) for t in plain)
    assert n_unk >= 3 and not any(str(EMO_UNK) in t.split() for t in ban) and plain != ban, (n_unk, plain, ban)
    path = os.path.join(os.path.dirname(HERE), "tests", "golden", "sensevoice_ban.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path), "bytes;", n_unk, "<|EMO_UNKNOWN|> tokens without the ban")


if __name__ == "__main__":
    main()
