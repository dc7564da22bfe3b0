"""Differential test of `SenseVoiceSmall.inference` (BASELINE configs[2]): this package's class against the
REFERENCE's own (funasr/models/sense_voice/model.py:918-1078, run for real on the CPU) on random ragged batches -- every
language / text-norm query the glue distinguishes incl. unknown languages, `ban_emo_unk`, feature and sound inputs, key forms.
The product's device half (`recognize_features`) is stood in by the CPU oracle, so what is compared is the host glue: query
selection, the key handling, the decoded strings, the records. tests/golden/sensevoice_inference.npz pins six fixed cases of
the same class; this sweeps. Build container only."""
import copy

import pytest
import torch

from funasr_amd import synth
from oracle import paraformer_oracle as O
from oracle import ref_import

pytestmark = pytest.mark.skipif(not ref_import.available(), reason="reference checkout not present (GPU box)")


class IdTokenizer:
    def decode(self, ids):
        return " ".join(str(int(i)) for i in ids)


class _Frontend:
    fs, frame_shift, lfr_n = 16000, 10, 6

    def __init__(self, feats, lens):
        self.feats, self.lens = feats, lens

    def __call__(self, data, data_len, **kwargs):
        return self.feats, self.lens


LID = {"auto": 0, "zh": 3, "en": 4, "yue": 7, "ja": 11, "ko": 12, "nospeech": 13}


def test_sensevoice_inference_equals_the_reference(monkeypatch):
    ref_import.install()
    from funasr.models.sense_voice.model import SenseVoiceSmall as RefSV
    from funasr_amd.sense_voice import SenseVoiceSmall
    cfg = synth.tiny(synth.SENSEVOICE_SMALL, enc_blocks=2, tp_blocks=1, vocab=25055)
    ec = dict(cfg["encoder"])
    input_size = ec.pop("input_size")
    g = torch.Generator().manual_seed(13)
    compared = 0
    for trial in range(24):
        sd = synth.sensevoice_state_dict(cfg, seed=300 + trial)
        sd["ctc.ctc_lo.bias"][0] += 1.0
        if trial % 4 == 1:
            sd["ctc.ctc_lo.bias"][25009] += 6.0                              # <|EMO_UNKNOWN|> wins frames unless banned
        ref = RefSV(encoder="SenseVoiceEncoderSmall", encoder_conf=dict(ec), input_size=input_size, vocab_size=cfg["vocab_size"]).eval()
        missing, unexpected = ref.load_state_dict(sd, strict=False)
        assert not unexpected and all(k.startswith("criterion") for k in missing), (missing, unexpected)
        ours = SenseVoiceSmall.from_config(cfg)
        ours.load_state_dict(sd, strict=False)
        seen = {}

        def recognize_features(speech, speech_lengths, language="auto", textnorm="woitn", return_intermediate=False, ban_ids=None, _sd=sd, _seen=seen):
            _seen.update(language=language, textnorm=textnorm, ban_ids=ban_ids)
            lens = torch.as_tensor(speech_lengths, dtype=torch.int32).reshape(-1)
            r = O.sensevoice_greedy(speech.float(), lens, _sd, cfg, language_id=LID.get(language, 0), textnorm_id={"withitn": 14, "woitn": 15}[textnorm])
            if ban_ids:
                logp = r["logp"].clone()
                logp[:, :, list(ban_ids)] = -float("inf")
                ids = []
                for b in range(speech.shape[0]):
                    y = torch.unique_consecutive(logp[b, : int(r["olens"][b])].argmax(-1))
                    ids.append(y[y != 0].tolist())
                return dict(ids=ids, olens=r["olens"])
            return dict(ids=r["ids"], olens=r["olens"])

        monkeypatch.setattr(ours, "recognize_features", recognize_features)
        B = int(torch.randint(1, 4, (1,), generator=g))
        T = int(torch.randint(6, 40, (1,), generator=g))
        lens = torch.randint(3, T + 1, (B,), generator=g, dtype=torch.int32)
        lens[int(torch.randint(0, B, (1,), generator=g))] = T
        feats = (torch.randn(B, (T + 2) // 3, 560, generator=g) * 0.8).repeat_interleave(3, dim=1)[:, :T].contiguous()
        for b in range(B):
            feats[b, lens[b]:] = 0
        kw = dict(device="cpu")
        kw["language"] = ["auto", "zh", "en", "yue", "ja", "ko", "nospeech", "klingon"][trial % 8]
        if trial % 3 == 0:
            kw["use_itn"] = bool(trial % 2)
        if trial % 5 == 1:
            kw["text_norm"] = ["withitn", "woitn"][trial % 2]
        if trial % 4 == 1:
            kw["ban_emo_unk"] = True
        keys = [f"u{b}" for b in range(B)] if trial % 4 else [[f"u{b}" for b in range(B)]]
        if trial % 2:
            r_in = dict(data_in=feats.clone(), data_lengths=lens.clone().long(), data_type="fbank")
            o_in = dict(data_in=feats.clone(), data_lengths=lens.clone(), data_type="fbank")
            r_fe = o_fe = None
        else:
            waves = [torch.zeros(int(l) * 960) for l in lens]
            r_in, o_in = dict(data_in=[w.clone() for w in waves]), dict(data_in=[w.clone() for w in waves])
            r_fe, o_fe = _Frontend(feats.clone(), lens.clone().long()), _Frontend(feats.clone(), lens.clone())
        with torch.no_grad():
            want, _ = ref.inference(key=copy.deepcopy(keys), tokenizer=IdTokenizer(), frontend=r_fe, **r_in, **copy.deepcopy(kw))
        got, _ = ours.inference(key=copy.deepcopy(keys), tokenizer=IdTokenizer(), frontend=o_fe, **o_in, **copy.deepcopy(kw))
        assert got == want, (trial, kw, got, want)
        assert seen["ban_ids"] == ([25009] if kw.get("ban_emo_unk") else None) and seen["language"] == kw["language"], (trial, seen)
        compared += len(want)
    assert compared > 30
