"""Host beam search with CTC prefix scoring (funasr_amd/beam_search.py) against the fixture recorded from the REFERENCE's own
BeamSearchPara + CTCPrefixScorer + LengthBonus (tests/golden/beam_search.npz, oracle/make_golden_beam.py): every hypothesis
of the n-best, in order, token for token; total scores to float32 round-off."""
import json
import os

import numpy as np
import torch

from funasr_amd.beam_search import BeamSearchPara

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def test_nbest_equals_reference_beam_search():
    g = np.load(os.path.join(GOLD, "beam_search.npz"), allow_pickle=False)
    cases = json.loads(str(g["cases"]))
    for ci, c in enumerate(cases):
        bs = BeamSearchPara(beam_size=c["beam"], vocab_size=c["V"], sos=1, eos=2, ctc_weight=c["ctc"],
                            length_bonus_weight=c["penalty"])
        nbest = bs(torch.from_numpy(g[f"am_{ci}"]), g[f"ctc_{ci}"])
        want = json.loads(str(g[f"nbest_{ci}"]))
        assert [h.yseq for h in nbest] == want, (ci, [h.yseq for h in nbest], want)
        assert np.allclose([h.score for h in nbest], g[f"scores_{ci}"], rtol=0, atol=2e-4), ci


def test_without_any_scorer_it_is_top1_per_position():
    """decoding_ctc_weight = 0 and penalty = 0 leave no scorer (search.py:72-75): the best hypothesis is the greedy path"""
    g = torch.Generator().manual_seed(0)
    logits = torch.randn(7, 30, generator=g)
    logits[:, 2] -= 50.0                     # <eos> never competes: a hypothesis that picks it ends early (search.py:413-421)
    am = torch.log_softmax(logits, -1)
    best = BeamSearchPara(beam_size=3, vocab_size=30, sos=1, eos=2)(am)[0]
    assert best.yseq == [1] + am.argmax(-1).tolist() + [2]
