"""Golden vectors for `SenseVoiceSmall.inference(..., output_timestamp=True)` made by the REFERENCE's own class and its own
`SentencepiecesTokenizer` (build container only; TEST INFRASTRUCTURE). Writes tests/golden/sensevoice_ts.npz and the small
sentencepiece model it trains, tests/golden/sv_bpe.model.

The branch under test (funasr/models/sense_voice/model.py:1036-1078 + utils/ctc_alignment.py + `post` :1080-1112) is host
logic over the CTC log-probabilities, so the reference is driven two ways:
  * INJECTED log-probabilities (the reference's `ctc.log_softmax` is replaced by a table): utterances whose CTC path spells
    real piece sequences -- four rich-tag pieces, Chinese and English words cut into several pieces, repeated pieces
    separated by blanks, runs of several frames per piece, blanks -- so that decoding, re-tokenising and aligning meet the
    cases `post` distinguishes; the product's `ctc_timestamps` is checked against them on the CPU;
  * the real seeded network on LFR features (random ids: whatever the decode / re-tokenise round trip makes of them).

    python oracle/make_golden_sensevoice_ts.py
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

GOLD = os.path.join(os.path.dirname(HERE), "tests", "golden")
WORDS = ["hello", "world", "speech", "model", "voice", "sense", "small", "time", "stamp", "today", "weather", "nice", "park", "walk",
         "语音", "识别", "今天", "天气", "不错", "我们", "公园", "散步"]
UNSEEN = ["timestamps", "walking", "smallest", "senseless", "天天", "modeling"]       # not in the bpe corpus: cut into several pieces
TAGS = ["<|zh|>", "<|en|>", "<|NEUTRAL|>", "<|HAPPY|>", "<|Speech|>", "<|woitn|>", "<|withitn|>"]


def train_bpe(path_prefix):
    import sentencepiece as spm
    rng = np.random.default_rng(0)
    corpus = path_prefix + ".corpus.txt"
    with open(corpus, "w", encoding="utf-8") as f:
        for _ in range(400):
            f.write(" ".join(WORDS[int(i)] for i in rng.integers(len(WORDS), size=int(rng.integers(3, 9)))) + "\n")
    spm.SentencePieceTrainer.train(input=corpus, model_prefix=path_prefix, vocab_size=128, model_type="bpe",
                                   user_defined_symbols=TAGS, character_coverage=1.0, add_dummy_prefix=False, minloglevel=2)
    os.remove(corpus)
    os.remove(path_prefix + ".vocab")


def main():
    ref_import.install()
    from funasr.models.sense_voice.model import SenseVoiceSmall
    from funasr.tokenizer.sentencepiece_tokenizer import SentencepiecesTokenizer
    bpe = os.path.join(GOLD, "sv_bpe")
    train_bpe(bpe)
    tok = SentencepiecesTokenizer(bpemodel=bpe + ".model")
    V = tok.get_vocab_size()
    cfg = synth.tiny(synth.SENSEVOICE_SMALL, enc_blocks=2, tp_blocks=1, vocab=V)
    sd = synth.sensevoice_state_dict(cfg, seed=29)
    sd["ctc.ctc_lo.bias"][0] += 1.0
    ec = dict(cfg["encoder"])
    input_size = ec.pop("input_size")
    model = SenseVoiceSmall(encoder="SenseVoiceEncoderSmall", encoder_conf=ec, input_size=input_size, vocab_size=V).eval()
    missing, unexpected = model.load_state_dict(sd, strict=False)
    assert not unexpected and all(k.startswith("criterion") for k in missing), (missing, unexpected)
    rng = np.random.default_rng(4)
    g = torch.Generator().manual_seed(3)
    # ---- injected CTC paths
    injected, tables = [], {}
    real_ls = model.ctc.log_softmax
    for ci in range(16):
        n_words = int(rng.integers(1, 7))
        pool = WORDS + UNSEEN * 2
        text = " ".join(pool[int(i)] for i in rng.integers(len(pool), size=n_words))
        if ci % 5 == 4:
            text = "today today " + text                          # the same pieces twice in a row
        ids = tok.encode("<|zh|><|NEUTRAL|><|Speech|><|woitn|>" + text)
        path = []
        for j, t in enumerate(ids):                               # a CTC path spelling `ids`
            if j and (ids[j - 1] == t or rng.random() < 0.5):
                path += [0] * int(rng.integers(1, 4))
            path += [t] * int(rng.integers(1, 4))
        path = [0] * int(rng.integers(0, 3)) + path + [0] * int(rng.integers(0, 4))
        T = len(path)
        lp = torch.full((1, T, V), -8.0) + 0.5 * torch.randn(1, T, V, generator=g)
        for t_, c in enumerate(path):
            lp[0, t_, c] = 4.0 + float(rng.random())
        lp = torch.log_softmax(lp, -1)
        model.ctc.log_softmax = lambda enc, _lp=lp: _lp.clone()
        feats = torch.randn(1, T - 4, 560, generator=g) * 0.5     # the encoder output is ignored; T = speech frames + 4 queries
        with torch.no_grad():
            res, _ = model.inference(feats, data_lengths=torch.tensor([T - 4]), key=["k"], tokenizer=tok, frontend=None,
                                     device="cpu", data_type="fbank", language="auto", output_timestamp=True)
        r = res[0]
        ts = [[float(a), float(b)] for a, b in r.get("timestamp", [])]
        tables[f"logp_{ci}"] = lp[0].numpy().astype(np.float32)
        injected.append(dict(text=r["text"], timestamp=ts, words=r.get("words"), has_ts="timestamp" in r))
    model.ctc.log_softmax = real_ls
    # ---- end to end with the real network
    B, T = 3, 45
    lens = torch.tensor([45, 31, 14], dtype=torch.int32)
    feats = (torch.randn(B, (T + 2) // 3, 560, generator=g) * 0.8).repeat_interleave(3, dim=1)[:, :T]
    for b in range(B):
        feats[b, lens[b]:] = 0
    with torch.no_grad():
        res, _ = model.inference(feats.clone(), data_lengths=lens.clone().long(), key=[f"u{i}" for i in range(B)], tokenizer=tok,
                                 frontend=None, device="cpu", data_type="fbank", language="auto", output_timestamp=True)
    e2e = [dict(text=r["text"], timestamp=[[float(a), float(b)] for a, b in r.get("timestamp", [])], words=r.get("words"),
                has_ts="timestamp" in r) for r in res]
    out = os.path.join(GOLD, "sensevoice_ts.npz")
    np.savez_compressed(out, config=json.dumps(cfg), seed=29, ctc_blank_bias_add=1.0, injected=json.dumps(injected, ensure_ascii=False),
                        feats=feats.numpy(), lens=lens.numpy(), e2e=json.dumps(e2e, ensure_ascii=False), **tables)
    print("wrote", out, os.path.getsize(out), "bytes; vocab", V, "bpe model", os.path.getsize(bpe + ".model"), "bytes")
    print("e.g.", injected[0]["text"], injected[0]["words"], injected[0]["timestamp"][:3])
    print("e2e", [(e["text"][:30], len(e["timestamp"])) for e in e2e])


if __name__ == "__main__":
    main()
