"""`generate_hotwords_list` of ContextualParaformer and SeacoParaformer (hotword strings, .txt files with one word per line,
the `seg_dict` next to the cmvn file that turns words into several tokens, empty and blank inputs) against the REFERENCE's own
methods (contextual_paraformer/model.py:518-607, seaco_paraformer/model.py:596-679) on random inputs. Build container only."""
import os
import random

import pytest

from oracle import make_golden_bicif as MB
from oracle import ref_import

pytestmark = pytest.mark.skipif(not ref_import.available(), reason="reference checkout not present (GPU box)")


class _Self:
    sos = 1


class _Frontend:
    def __init__(self, cmvn_file):
        self.cmvn_file = cmvn_file


def test_generate_hotwords_list_equals_the_reference(tmp_path):
    ref_import.install()
    from funasr.models.contextual_paraformer.model import ContextualParaformer as RefC
    from funasr.models.seaco_paraformer.model import SeacoParaformer as RefS
    from funasr.tokenizer.char_tokenizer import CharTokenizer as RefTok
    from funasr_amd.contextual_paraformer import ContextualParaformer
    from funasr_amd.seaco_paraformer import SeacoParaformer
    from funasr_amd.tokenizer import CharTokenizer
    with open(tmp_path / "seg_dict", "w", encoding="utf-8") as f:
        for ch in MB.VOCAB[3:-10]:
            f.write(f"{ch} {ch}\n")
        f.write("hello hel@@ lo\nworld wor@@ ld\nthe the\n")
    with_dict, without = _Frontend(os.path.join(str(tmp_path), "am.mvn")), _Frontend(None)
    rtok, tok = RefTok(token_list=MB.VOCAB, unk_symbol="<unk>"), CharTokenizer(token_list=MB.VOCAB, unk_symbol="<unk>")
    words = list(MB.VOCAB[3:-10]) + ["hello", "world", "the", "我们", "大地", "zzz", "国家大"]
    rng = random.Random(1)
    n = 0
    for trial in range(120):
        k = rng.choice([0, 1, 2, 5, 12])
        hw = " ".join(rng.choice(words) for _ in range(k)) if k else rng.choice([None, "", " "])
        forms = [hw]
        if hw and hw.strip():
            p = tmp_path / f"hw{trial}.txt"
            p.write_text("\n".join(hw.split()) + "\n", encoding="utf-8")
            forms.append(str(p))
        for form in forms:
            for ref_cls, our_cls in ((RefC, ContextualParaformer), (RefS, SeacoParaformer)):
                for fe in (with_dict, without):
                    want = ref_cls.generate_hotwords_list(_Self(), form, tokenizer=rtok, frontend=fe)
                    got = our_cls.generate_hotwords_list(_Self(), form, tokenizer=tok, frontend=fe)
                    assert got == want, (ref_cls.__name__, form, fe.cmvn_file, got, want)
                    n += 1
    assert n > 500
