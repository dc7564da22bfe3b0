#!/usr/bin/env python3
"""Golden vectors for the CT-Transformer punctuation path, produced by the REFERENCE's own class
(funasr/models/ct_transformer/model.py `CTTransformer`): TEST INFRASTRUCTURE, build container only; writes
tests/golden/punc.npz.
  * network: punc_forward logits of the reference class on seeded weights (d_model 256, 8 heads of 32, 4 blocks) for a
    ragged batch of id sequences -- also pins oracle/punc_oracle.py;
  * sentence assembly: the reference's inference() driven with an INJECTED network (oracle.punc_oracle.injected_marks) on
    Chinese, English and mixed texts from 1 to ~260 words, incl. texts without any sentence end (comma cut after 200
    carried words) and other mini-sentence sizes;
  * end to end: the reference's inference() with the real seeded network on a few texts.
"""
import json
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
from oracle import punc_oracle, ref_import  # noqa: E402

ENC = dict(input_size=256, output_size=256, attention_heads=8, linear_units=1024, num_blocks=4, kernel_size=11, sanm_shfit=0)
CJK = list(dict.fromkeys("今天天气真不错我们一起去公园散步吧欢迎大家来体验语音识别模型的效果如何请告诉你他她它"))
ENW = ["hello", "world", "the", "quick", "brown", "fox", "i", "am", "fine", "thanks", "ok", "iphone", "don't", "gpu", "a"]
VOCAB = ["<unk>"] + CJK + ENW + ["<pad>"]


def random_text(rng, n, kind):
    words = []
    for _ in range(n):
        r = rng.random()
        if kind == "zh" or (kind == "mix" and r < 0.55):
            words.append(CJK[int(rng.integers(len(CJK)))])
        elif r < 0.97:
            words.append(ENW[int(rng.integers(len(ENW)))])
        else:
            words.append("zzz")                                   # out of vocabulary -> <unk>
    out = ""
    for i, w in enumerate(words):                                 # ASCII words need blanks, CJK runs are written together
        if i and (len(w[0].encode()) == 1 or len(words[i - 1][0].encode()) == 1):
            out += " "
        out += w
    return out


def main():
    ref_import.install()
    import funasr.models.sanm.encoder  # noqa: F401  (registers SANMEncoder)
    from funasr.models.ct_transformer.model import CTTransformer
    from funasr.tokenizer.char_tokenizer import CharTokenizer
    torch.set_num_threads(4)
    enc_conf = dict(input_size=256, output_size=256, attention_heads=8, linear_units=1024, num_blocks=4, dropout_rate=0.1,
                    positional_dropout_rate=0.1, attention_dropout_rate=0.0, input_layer="pe",
                    pos_enc_class="SinusoidalPositionEncoder", normalize_before=True, kernel_size=11, sanm_shfit=0,
                    selfattention_layer_type="sanm", padding_idx=0)
    model = CTTransformer(encoder="SANMEncoder", encoder_conf=enc_conf, vocab_size=len(VOCAB), punc_list=punc_oracle.PUNC_LIST,
                          embed_unit=256, att_unit=256, ignore_id=0, sentence_end_id=3).eval()
    sd = punc_oracle.synthetic_state_dict(len(VOCAB), ENC, seed=9)
    missing, unexpected = model.load_state_dict(sd, strict=False)
    assert not unexpected and not [m for m in missing if "embed.1" not in m], (missing, unexpected)
    tok = CharTokenizer(token_list=VOCAB, unk_symbol="<unk>")
    rng = np.random.default_rng(12)
    # ---- network
    ids = torch.from_numpy(rng.integers(1, len(VOCAB) - 1, size=(3, 57)).astype(np.int64))
    lens = torch.tensor([57, 20, 3], dtype=torch.int32)
    with torch.no_grad():
        y, _ = model.punc_forward(ids, lens)
    mine = punc_oracle.punc_forward(ids, lens, sd, ENC)
    err = max((mine[b, : lens[b]] - y[b, : lens[b]]).abs().max().item() for b in range(3))
    assert err < 2e-5, f"oracle differs from the reference network by {err}"
    # ---- sentence assembly with an injected network
    cases = []
    real_forward = model.punc_forward
    for ci in range(60):
        kind = ("zh", "en", "mix")[ci % 3]
        n = int(rng.choice([1, 2, 5, 19, 20, 21, 40, 63, 130, 260])) if ci < 50 else int(rng.integers(1, 90))
        text = random_text(rng, n, kind)
        never_end = ci % 7 == 3
        split = 20 if ci % 5 else int(rng.choice([5, 8, 33]))

        def fake(text, text_lengths, _ne=never_end, **kw):
            m = punc_oracle.injected_marks(text[0].cpu().numpy(), _ne)
            return torch.nn.functional.one_hot(torch.from_numpy(m), 6).float()[None], None
        model.punc_forward = fake
        res, _ = model.inference([text], key=["k"], tokenizer=tok, device="cpu", split_size=split)
        cases.append(dict(text=text, never_end=never_end, split_size=split, out=res[0]["text"],
                          punc_array=[int(x) for x in res[0]["punc_array"].tolist()]))
    model.punc_forward = real_forward
    # ---- end to end with the real network
    e2e = []
    for text in ("今天天气真不错我们一起去公园散步吧", "hello world i am fine thanks", "欢迎大家来体验 iphone 的语音识别效果如何 ok thanks",
                 random_text(rng, 75, "mix"), random_text(rng, 230, "zh")):
        with torch.no_grad():
            res, _ = model.inference([text], key=["k"], tokenizer=tok, device="cpu")
        e2e.append(dict(text=text, out=res[0]["text"], punc_array=[int(x) for x in res[0]["punc_array"].tolist()]))
    out = os.path.join(os.path.dirname(HERE), "tests", "golden", "punc.npz")
    np.savez_compressed(out, ids=ids.numpy(), lens=lens.numpy(), logits=y.numpy(), seed=9, vocab=json.dumps(VOCAB, ensure_ascii=False),
                        enc_cfg=json.dumps(ENC), cases=json.dumps(cases, ensure_ascii=False), e2e=json.dumps(e2e, ensure_ascii=False))
    print(f"wrote {out}: oracle vs reference network {err:.2e}; {len(cases)} assembly cases, {len(e2e)} end-to-end texts")
    print("e.g.", e2e[2]["out"], "|", cases[2]["out"][:60])


if __name__ == "__main__":
    main()
