"""Beam search with CTC prefix scoring for Paraformer's non-autoregressive decoder output (host bookkeeping).

Mirrors what `Paraformer.inference` runs when `decoding_ctc_weight > 0` on a model that has a CTC head
(funasr/models/paraformer/model.py:482-532 `init_beam_search`, :554-562 dispatch, :629-637 the call):
`BeamSearchPara` (funasr/models/paraformer/search.py:35-451) with the scorers `CTCPrefixScorer`
(funasr/models/transformer/scorers/ctc.py:11-96 over `CTCPrefixScore`, scorers/ctc_prefix_score.py:255-345) and
`LengthBonus` (scorers/length_bonus.py:12-34), plus `end_detect` (funasr/metrics/common.py:19-49).

The device supplies the two score matrices -- the decoder's log-softmax [N, V] and the CTC head's log-softmax [T, V]
(`ops.log_softmax` over the HIP GEMM outputs); everything in this file is the reference's per-hypothesis bookkeeping
on the host: O(N * beam * (pre_beam * T + V)) scalar work, which the reference runs in numpy / CPU torch as well.
Details kept as the reference has them: the decoder score enters with weight 1 (search.py:291 adds `am_score` before
any weighting), `lm` / `ngram` have no scorer object and therefore never contribute (model.py:505-508), the candidate
list is re-sorted after every expanded hypothesis (search.py:319-321), <eos> is appended at the last position (:407-411).
"""
from __future__ import annotations

from dataclasses import dataclass, field
from typing import Dict, List, Optional, Tuple

import numpy as np
import torch

LOGZERO = -10000000000.0


@dataclass
class Hypothesis:
    yseq: List[int]
    score: float = 0.0
    scores: Dict[str, float] = field(default_factory=dict)
    ctc_state: Optional[Tuple[float, np.ndarray]] = None     # (previous prefix score, r [T, 2])


class CTCPrefixScore:
    """scorers/ctc_prefix_score.py:255-345 (Algorithm 2 of Watanabe et al., several next labels at once), numpy float32"""

    def __init__(self, logp: np.ndarray, blank: int, eos: int):
        self.x, self.blank, self.eos, self.T = logp, blank, eos, len(logp)

    def initial_state(self) -> np.ndarray:
        r = np.full((self.T, 2), LOGZERO, dtype=np.float32)
        r[0, 1] = self.x[0, self.blank]
        for i in range(1, self.T):
            r[i, 1] = r[i - 1, 1] + self.x[i, self.blank]
        return r

    def __call__(self, y: List[int], cs: np.ndarray, r_prev: np.ndarray):
        out_len = len(y) - 1                                  # ignore sos
        # (the reference leaves the rows before out_len - 1 uninitialised; they are never read -- filled here so that no
        # garbage NaN ever reaches logaddexp)
        r = np.full((self.T, 2, len(cs)), LOGZERO, dtype=np.float32)
        xs = self.x[:, cs]
        if out_len == 0:
            r[0, 0] = xs[0]
            r[0, 1] = LOGZERO
        else:
            r[out_len - 1] = LOGZERO
        r_sum = np.logaddexp(r_prev[:, 0], r_prev[:, 1])
        last = y[-1]
        if out_len > 0 and last in cs:
            log_phi = np.ndarray((self.T, len(cs)), dtype=np.float32)
            for i in range(len(cs)):
                log_phi[:, i] = r_sum if cs[i] != last else r_prev[:, 1]
        else:
            log_phi = r_sum
        start = max(out_len, 1)
        log_psi = r[start - 1, 0]
        for t in range(start, self.T):
            r[t, 0] = np.logaddexp(r[t - 1, 0], log_phi[t - 1]) + xs[t]
            r[t, 1] = np.logaddexp(r[t - 1, 0], r[t - 1, 1]) + self.x[t, self.blank]
            log_psi = np.logaddexp(log_psi, log_phi[t - 1] + xs[t])
        eos_pos = np.where(cs == self.eos)[0]
        if len(eos_pos) > 0:
            log_psi[eos_pos] = r_sum[-1]
        blank_pos = np.where(cs == self.blank)[0]
        if len(blank_pos) > 0:
            log_psi[blank_pos] = LOGZERO
        return log_psi, np.rollaxis(r, 2)


def end_detect(ended: List[Hypothesis], i: int, M: int = 3, D_end: float = float(np.log(1 * np.exp(-10)))) -> bool:
    """funasr/metrics/common.py:19-49"""
    if not ended:
        return False
    best = max(ended, key=lambda h: h.score)
    count = 0
    for m in range(M):
        same = [h for h in ended if len(h.yseq) == i - m]
        if same and max(same, key=lambda h: h.score).score - best.score < D_end:
            count += 1
    return count == M


class BeamSearchPara:
    """search.py:35-451 for the scorer set `init_beam_search` builds: ctc (partial scorer, weight `decoding_ctc_weight`),
    length_bonus (full scorer, weight `penalty`); scorers with weight 0 are dropped (:72-75)."""

    def __init__(self, beam_size: int, vocab_size: int, sos: int, eos: int, ctc_weight: float = 0.0,
                 length_bonus_weight: float = 0.0, blank: int = 0, pre_beam_ratio: float = 1.5, pre_beam: bool = True):
        self.beam_size, self.n_vocab, self.sos, self.eos, self.blank = beam_size, vocab_size, sos, eos, blank
        self.w_ctc, self.w_len = float(ctc_weight), float(length_bonus_weight)
        self.pre_beam_size = int(pre_beam_ratio * beam_size)
        # pre_beam_score_key = "full" unless the model's ctc_weight is 1.0 (model.py:526)
        self.do_pre_beam = pre_beam and self.pre_beam_size < vocab_size and self.w_ctc != 0

    def _beam(self, weighted: torch.Tensor, ids: torch.Tensor):
        """search.py:205-231"""
        if weighted.size(0) == ids.size(0):
            top = weighted.topk(self.beam_size)[1]
            return top, top
        tmp = weighted[ids]
        weighted[:] = -float("inf")
        weighted[ids] = tmp
        return weighted.topk(self.beam_size)[1], weighted[ids].topk(self.beam_size)[1]

    def _search(self, running: List[Hypothesis], am: torch.Tensor, ctc: Optional[CTCPrefixScore]) -> List[Hypothesis]:
        """one position (search.py:276-322)"""
        best: List[Hypothesis] = []
        part_ids = torch.arange(self.n_vocab)                  # no pre-beam
        for hyp in running:
            weighted = torch.zeros(self.n_vocab, dtype=am.dtype)
            weighted += am
            if self.w_len != 0:
                weighted += self.w_len * torch.ones(self.n_vocab, dtype=am.dtype)     # LengthBonus.score: 1 per token
            part_scores = None
            new_state = None
            if self.w_ctc != 0 and ctc is not None:
                if self.do_pre_beam:
                    part_ids = torch.topk(weighted, self.pre_beam_size)[1]
                prev_score, r_prev = hyp.ctc_state
                presub, new_r = ctc(hyp.yseq, part_ids.numpy(), r_prev)
                part_scores = torch.as_tensor(presub - prev_score, dtype=am.dtype)
                new_state = (presub, new_r)
                weighted[part_ids] += self.w_ctc * part_scores
            weighted += hyp.score
            for j, pj in zip(*self._beam(weighted, part_ids)):
                j, pj = int(j), int(pj)
                scores = dict(hyp.scores)
                if self.w_len != 0:
                    scores["length_bonus"] = scores.get("length_bonus", 0.0) + 1.0
                st = None
                if part_scores is not None:
                    scores["ctc"] = scores.get("ctc", 0.0) + float(part_scores[pj])
                    st = (new_state[0][pj], new_state[1][pj])
                best.append(Hypothesis(yseq=hyp.yseq + [j], score=float(weighted[j]), scores=scores, ctc_state=st))
            best = sorted(best, key=lambda h: h.score, reverse=True)[: min(len(best), self.beam_size)]
        return best

    def __call__(self, am_scores: torch.Tensor, ctc_logp: Optional[np.ndarray] = None, maxlenratio: float = 0.0,
                 minlenratio: float = 0.0) -> List[Hypothesis]:
        """am_scores [N, V] decoder log-probs (host), ctc_logp [T, V] CTC log-probs (host numpy) -> n-best, best first.
        yseq includes <sos> ... <eos> like the reference's Hypothesis.yseq."""
        am_scores = am_scores.detach().to("cpu", torch.float32)
        maxlen = am_scores.shape[0]
        ctc = None
        init_state = None
        if self.w_ctc != 0 and ctc_logp is not None:
            ctc = CTCPrefixScore(np.asarray(ctc_logp, dtype=np.float32), self.blank, self.eos)
            init_state = (0.0, ctc.initial_state())
        init_scores = {}
        if self.w_ctc != 0 and ctc is not None:
            init_scores["ctc"] = 0.0
        if self.w_len != 0:
            init_scores["length_bonus"] = 0.0
        running = [Hypothesis(yseq=[self.sos], score=0.0, scores=init_scores, ctc_state=init_state)]
        ended: List[Hypothesis] = []
        for i in range(maxlen):
            best = self._search(running, am_scores[i], ctc)
            if i == maxlen - 1:                                 # search.py:407-411
                best = [Hypothesis(h.yseq + [self.eos], h.score, h.scores, h.ctc_state) for h in best]
            running = []
            for h in best:                                      # final_score() of both scorers is 0 (scorer_interface.py)
                (ended if h.yseq[-1] == self.eos else running).append(h)
            if maxlenratio == 0.0 and end_detect(ended, i):
                break
            if not running:
                break
        return sorted(ended, key=lambda h: h.score, reverse=True)
