"""Golden vectors for the CTC-rescored beam search, made by the REFERENCE's own `BeamSearchPara`
(funasr/models/paraformer/search.py) with `CTCPrefixScorer` / `LengthBonus` exactly as `Paraformer.init_beam_search`
wires them (funasr/models/paraformer/model.py:482-532). Build container only; TEST INFRASTRUCTURE.
Writes tests/golden/beam_search.npz: per case the decoder log-probs [N, V], the CTC log-probs [T, V], the settings and
the reference's n-best (token sequences incl. <sos>/<eos>, total scores).

    python oracle/make_golden_beam.py
"""
import json
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
from oracle import ref_import  # noqa: E402

CASES = [dict(V=50, N=6, T=14, beam=2, ctc=0.5, penalty=0.0, seed=1),
         dict(V=50, N=9, T=20, beam=3, ctc=0.3, penalty=0.0, seed=2),
         dict(V=80, N=7, T=16, beam=2, ctc=0.5, penalty=0.4, seed=3),
         dict(V=60, N=8, T=18, beam=5, ctc=0.7, penalty=0.0, seed=4),
         dict(V=40, N=10, T=25, beam=2, ctc=0.5, penalty=0.0, seed=5, eos_boost=3.0),   # <eos> wins early: ended hypotheses, end_detect
         dict(V=50, N=6, T=14, beam=3, ctc=0.0, penalty=0.6, seed=6),                   # no CTC scorer (weight 0 is dropped)
         dict(V=50, N=5, T=9, beam=2, ctc=1.0, penalty=0.0, seed=7, peaky=True)]        # CTC dominates, repeated labels


class _Ctc:
    def __init__(self, logp):
        self.logp = logp

    def log_softmax(self, x):
        return self.logp.unsqueeze(0)


def main():
    ref_import.install()
    from funasr.models.paraformer.search import BeamSearchPara
    from funasr.models.transformer.scorers.ctc import CTCPrefixScorer
    from funasr.models.transformer.scorers.length_bonus import LengthBonus
    out = {"cases": json.dumps(CASES)}
    for ci, c in enumerate(CASES):
        g = torch.Generator().manual_seed(c["seed"])
        V, N, T = c["V"], c["N"], c["T"]
        dec = torch.randn(N, V, generator=g) * 2.0
        ctc = torch.randn(T, V, generator=g) * (4.0 if c.get("peaky") else 2.0)
        ctc[:, 0] += 2.0                                        # blank is frequent, as in a trained CTC head
        if c.get("eos_boost"):
            dec[N // 2, 2] += c["eos_boost"] + 6.0
            ctc[:, 2] += 1.0
        if c.get("peaky"):                                       # the same label twice in a row: exercises `last in cs`
            dec[1] = dec[0]
        am = torch.log_softmax(dec, dim=-1)
        logp = torch.log_softmax(ctc, dim=-1)
        token_list = [str(i) for i in range(V)]
        scorers = {"ctc": CTCPrefixScorer(ctc=_Ctc(logp), eos=2), "length_bonus": LengthBonus(V), "ngram": None}
        weights = dict(decoder=1.0 - c["ctc"], ctc=c["ctc"], lm=0.0, ngram=0.0, length_bonus=c["penalty"])
        bs = BeamSearchPara(beam_size=c["beam"], weights=weights, scorers=scorers, sos=1, eos=2, vocab_size=V,
                            token_list=token_list, pre_beam_score_key="full")
        with torch.no_grad():
            nbest = bs(x=torch.zeros(T, 8), am_scores=am, maxlenratio=0.0, minlenratio=0.0)
        seqs = [[int(t) for t in h.yseq] for h in nbest]
        scores = [float(h.score) for h in nbest]
        print(f"case {ci}: {len(nbest)} hyps, best {seqs[0]} {scores[0]:.4f}")
        out[f"am_{ci}"] = am.numpy()
        out[f"ctc_{ci}"] = logp.numpy()
        out[f"nbest_{ci}"] = json.dumps(seqs)
        out[f"scores_{ci}"] = np.asarray(scores, dtype=np.float64)
    path = os.path.join(os.path.dirname(HERE), "tests", "golden", "beam_search.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()
