"""Fuzz the punctuation stage's HOST logic of the product (funasr_amd/ct_transformer.py: split_words, tokenisation, mini-sentence
windows with the carried unfinished sentence, mark post-processing, text assembly) against the REFERENCE's own
CTTransformer.inference (funasr/models/ct_transformer/model.py:244-400) driven with the same INJECTED network
(punc_oracle.injected_marks) on random Chinese / English / mixed texts, lengths around every window boundary, random
split sizes (build container only; TEST INFRASTRUCTURE). tests/golden/punc.npz pins 60 such cases; this sweeps the space.

    python -m oracle.fuzz_punc_assembly_vs_reference [n_cases]
"""
import json
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
from oracle import make_golden_punc as G  # noqa: E402
from oracle import punc_oracle, ref_import  # noqa: E402


def main(n_cases=2000):
    ref_import.install()
    import funasr.models.sanm.encoder  # noqa: F401
    from funasr.models.ct_transformer.model import CTTransformer
    from funasr.tokenizer.char_tokenizer import CharTokenizer as RefTok
    from funasr_amd.ct_transformer import assemble, split_words
    from funasr_amd.tokenizer import CharTokenizer
    enc_conf = dict(input_size=256, output_size=256, attention_heads=8, linear_units=1024, num_blocks=1, dropout_rate=0.1,
                    positional_dropout_rate=0.1, attention_dropout_rate=0.0, input_layer="pe",
                    pos_enc_class="SinusoidalPositionEncoder", normalize_before=True, kernel_size=11, sanm_shfit=0,
                    selfattention_layer_type="sanm", padding_idx=0)
    model = CTTransformer(encoder="SANMEncoder", encoder_conf=enc_conf, vocab_size=len(G.VOCAB), punc_list=punc_oracle.PUNC_LIST,
                          embed_unit=256, att_unit=256, ignore_id=0, sentence_end_id=3).eval()
    rtok = RefTok(token_list=G.VOCAB, unk_symbol="<unk>")
    tok = CharTokenizer(token_list=G.VOCAB, unk_symbol="<unk>")
    rng = np.random.default_rng(77)
    bad = 0
    for ci in range(n_cases):
        kind = ("zh", "en", "mix")[ci % 3]
        split = int(rng.choice([3, 5, 8, 20, 20, 20, 33, 64]))
        n = int(rng.choice([1, 2, split - 1, split, split + 1, 2 * split - 1, 2 * split, 2 * split + 1, int(rng.integers(1, 300))]))
        sentence = G.random_text(rng, max(n, 1), kind)
        never_end = bool(rng.random() < 0.2)

        def fake(text, text_lengths, _ne=never_end, **kw):
            m = punc_oracle.injected_marks(text[0].cpu().numpy(), _ne)
            return torch.nn.functional.one_hot(torch.from_numpy(m), 6).float()[None], None
        model.punc_forward = fake
        res, _ = model.inference([sentence], key=["k"], tokenizer=rtok, device="cpu", split_size=split)
        words = split_words(sentence)
        got_text, marks = assemble(words, tok.encode(words), lambda x, ne=never_end: punc_oracle.injected_marks(x, ne),
                                   punc_oracle.PUNC_LIST, 3, split_size=split)
        if got_text != res[0]["text"] or [int(m) for m in marks] != [int(x) for x in res[0]["punc_array"].tolist()]:
            bad += 1
            print("DIFFERS", ci, dict(kind=kind, split=split, n=n, never_end=never_end), repr(sentence[:80]), "|", got_text[:80], "|", res[0]["text"][:80])
    out = dict(cases=n_cases, cases_with_a_difference=bad)
    print(json.dumps(out))
    return out


if __name__ == "__main__":
    main(int(sys.argv[1]) if len(sys.argv) > 1 else 2000)
