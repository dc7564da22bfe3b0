"""Golden cases for funasr_amd/postprocess_hotwords.py made by the REFERENCE module (funasr/utils/postprocess_hotwords.py;
build container only; TEST INFRASTRUCTURE) -> tests/golden/postprocess_hotwords.json. `pypinyin` / `rapidfuzz` are not
installed here, so the fuzzy search is driven with stand-ins on both sides (`fake_pinyin`, `fake_ratio` below -- the test
imports them from this file's copy in tests/): what is pinned is the window search, the candidate selection and the
replacement, not the two libraries.

    python oracle/make_golden_hotwords.py
"""
import difflib
import importlib.util
import json
import os
import random
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))


def fake_pinyin(text, style=None, errors="ignore"):
    out = []
    for ch in text:
        if "一" <= ch <= "鿿":
            out.append("bpmfdtnl"[ord(ch) % 8] + "aoeiu"[ord(ch) % 5] + ("ng" if ord(ch) % 3 == 0 else ""))
        elif ch.isascii() and ch.isalnum():
            out.append(ch)
    return out


class FakeStyle:
    NORMAL = 0


class FakeFuzz:
    @staticmethod
    def ratio(a, b):
        return 100.0 * difflib.SequenceMatcher(None, a, b).ratio()


def load_reference():
    path = "/root/reference/funasr/utils/postprocess_hotwords.py"
    spec = importlib.util.spec_from_file_location("_ref_postprocess_hotwords", path)
    mod = importlib.util.module_from_spec(spec)
    sys.modules["_ref_postprocess_hotwords"] = mod
    spec.loader.exec_module(mod)
    mod._LAZY_PINYIN, mod._PINYIN_STYLE, mod._RAPIDFUZZ_FUZZ = fake_pinyin, FakeStyle, FakeFuzz
    return mod


CHARS = list("科大讯飞迅东方财富撒贝宁你康辉灰的语音识别很强我喜欢新闻主持节目") + ["a", "b", "AI", "3", " ", "，"]


def main():
    ref = load_reference()
    rng = random.Random(11)
    cases = []
    for _ in range(300):
        n_t = rng.randint(1, 4)
        targets = ["".join(rng.choice(CHARS[:30]) for _ in range(rng.randint(2, 4))) for _ in range(n_t)]
        explicit = {}
        for _ in range(rng.randint(0, 2)):
            w = "".join(rng.choice(CHARS[:30]) for _ in range(rng.randint(1, 3)))
            explicit[w] = "".join(rng.choice(CHARS[:30]) for _ in range(rng.randint(1, 3)))
        text = ""
        for _ in range(rng.randint(0, 6)):
            r = rng.random()
            if r < 0.3:                                    # a corrupted copy of a target
                t = list(rng.choice(targets))
                t[rng.randrange(len(t))] = rng.choice(CHARS[:30])
                text += "".join(t)
            elif r < 0.45 and explicit:
                text += rng.choice(list(explicit))
            else:
                text += "".join(rng.choice(CHARS) for _ in range(rng.randint(1, 5)))
        thr = rng.choice([0.6, 0.75, 0.85, 1.0])
        m = ref.PostprocessHotwordMatcher(explicit_map=explicit, fuzzy_targets=targets, threshold=thr)
        out, matches = m.apply_text(text)
        cases.append(dict(explicit=explicit, targets=targets, threshold=thr, text=text, out=out,
                          matches=[x.as_dict() for x in matches]))
    parse = []
    for src in (["科大讯飞", "东方财富"], {"科大迅飞": "科大讯飞", "东方财富": "东方财富"}, ["撒贝你=>撒贝宁", "康辉"],
                "# c\n科大讯飞\n科大迅飞=>科大讯飞\n a -> b \nx→\n", None, [None, "", " q "], {"": "z", "k": ""}):
        e, f = ref.parse_postprocess_hotwords(src)
        parse.append([src, e, f])
    path = os.path.join(os.path.dirname(HERE), "tests", "golden", "postprocess_hotwords.json")
    with open(path, "w", encoding="utf-8") as f:
        json.dump(dict(cases=cases, parse=parse), f, ensure_ascii=False, indent=0)
    print("wrote", path, len(cases), "cases;", sum(1 for c in cases if c["matches"]), "with matches;",
          sum(1 for c in cases if any(m["score"] < 1 for m in c["matches"])), "with fuzzy matches")


if __name__ == "__main__":
    main()
