"""Fuzz the host beam search of the product (funasr_amd/beam_search.py: BeamSearchPara + CTC prefix scorer + length bonus)
against the REFERENCE's own BeamSearchPara / CTCPrefixScorer / LengthBonus (funasr/models/paraformer/search.py, wired as
Paraformer.init_beam_search does, model.py:482-532) on random decoder / CTC log-probabilities, vocabulary sizes, beam sizes,
CTC weights and penalties (build container only; TEST INFRASTRUCTURE). tests/golden/beam_search.npz pins seven cases; this
sweeps the space: the n-best must agree hypothesis for hypothesis, scores to float32 round-off.

    python -m oracle.fuzz_beam_vs_reference [n_cases]
"""
import json
import os
import sys

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
from oracle import ref_import  # noqa: E402
from oracle.make_golden_beam import _Ctc  # noqa: E402


def main(n_cases=300):
    ref_import.install()
    from funasr.models.paraformer.search import BeamSearchPara as RefBeam
    from funasr.models.transformer.scorers.ctc import CTCPrefixScorer
    from funasr.models.transformer.scorers.length_bonus import LengthBonus
    from funasr_amd.beam_search import BeamSearchPara
    g = torch.Generator().manual_seed(5)
    bad, worst, hyps = 0, 0.0, 0
    for ci in range(n_cases):
        V = int(torch.randint(8, 120, (1,), generator=g))
        N = int(torch.randint(1, 14, (1,), generator=g))
        T = int(torch.randint(N, 3 * N + 6, (1,), generator=g))
        beam = int(torch.randint(1, 7, (1,), generator=g))
        ctc_w = [0.0, 0.3, 0.5, 0.7, 1.0][int(torch.randint(0, 5, (1,), generator=g))]
        penalty = [0.0, 0.0, 0.4, 1.0][int(torch.randint(0, 4, (1,), generator=g))]
        sharp = float(torch.rand(1, generator=g)) * 5.0 + 0.5
        dec = torch.randn(N, V, generator=g) * sharp
        ctc = torch.randn(T, V, generator=g) * sharp
        ctc[:, 0] += 2.0
        if ci % 6 == 0:
            dec[N // 2, 2] += 8.0                                   # <eos> wins early: ended hypotheses
        if ci % 7 == 0 and N > 1:
            dec[1] = dec[0]                                          # repeated labels
        am, logp = torch.log_softmax(dec, -1), torch.log_softmax(ctc, -1)
        scorers = {"ctc": CTCPrefixScorer(ctc=_Ctc(logp), eos=2), "length_bonus": LengthBonus(V), "ngram": None}
        weights = dict(decoder=1.0 - ctc_w, ctc=ctc_w, lm=0.0, ngram=0.0, length_bonus=penalty)
        rb = RefBeam(beam_size=beam, weights=weights, scorers=scorers, sos=1, eos=2, vocab_size=V,
                     token_list=[str(i) for i in range(V)], pre_beam_score_key="full")
        with torch.no_grad():
            want = rb(x=torch.zeros(T, 8), am_scores=am, maxlenratio=0.0, minlenratio=0.0)
        got = BeamSearchPara(beam_size=beam, vocab_size=V, sos=1, eos=2, ctc_weight=ctc_w, length_bonus_weight=penalty)(am, logp.numpy())
        ws, gs = [[int(t) for t in h.yseq] for h in want], [list(h.yseq) for h in got]
        hyps += len(ws)
        if ws != gs:
            bad += 1
            print("NBEST DIFFERS", ci, dict(V=V, N=N, T=T, beam=beam, ctc=ctc_w, penalty=penalty), ws[:2], gs[:2])
            continue
        for a, b in zip(want, got):
            worst = max(worst, abs(float(a.score) - float(b.score)))
    out = dict(cases=n_cases, hypotheses=hyps, cases_with_a_different_nbest=bad, max_abs_score_diff=worst)
    print(json.dumps(out))
    return out


if __name__ == "__main__":
    main(int(sys.argv[1]) if len(sys.argv) > 1 else 300)
