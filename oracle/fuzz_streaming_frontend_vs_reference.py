"""Fuzz the oracle's online frontend (streaming_oracle.frontend_step: left-over samples, LFR splice cache, final flush)
against the REFERENCE's WavFrontendOnline.forward on random piece sequences (build container only; TEST INFRASTRUCTURE):
piece lengths from a few samples to several chunks, pieces too short for a frame or for an LFR row, the final flush with
and without new frames. Both sides call the same kaldi fbank restatement (torchaudio is absent), so the comparison is exact.

    python -m oracle.fuzz_streaming_frontend_vs_reference [n_sessions]
"""
import json
import os
import sys

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
from funasr_amd import synth  # noqa: E402
from oracle import make_golden_streaming as G  # noqa: E402
from oracle import paraformer_oracle as O  # noqa: E402
from oracle import streaming_oracle as S  # noqa: E402


def main(n_sessions=200):
    _, frontend, _, _ = G.build()
    cmvn = O.load_cmvn(os.path.join(G.GOLD, "am.mvn"))
    g = torch.Generator().manual_seed(7)
    worst, bad, calls, rows, empty_calls, ref_raised = 0.0, 0, 0, 0, 0, 0
    for si in range(n_sessions):
        n_pieces = int(torch.randint(1, 9, (1,), generator=g))
        kind = int(torch.randint(0, 4, (1,), generator=g))
        lens = []
        for _ in range(n_pieces):
            if kind == 0:
                lens.append(int(torch.randint(1, 700, (1,), generator=g)))             # mostly too short for anything
            elif kind == 1:
                lens.append(int(torch.randint(300, 4000, (1,), generator=g)))
            elif kind == 2:
                lens.append(960 * int(torch.randint(1, 21, (1,), generator=g)))        # the strides of ParaformerStreaming.inference
            else:
                lens.append(int(torch.randint(1, 20000, (1,), generator=g)))
        wav = synth.speech_like(sum(lens) + 1, seed=si)[:sum(lens)]
        wav = (wav * 32768.0).round().clamp(-32768, 32767) / 32768.0
        rc, oc = {}, S.frontend_init()
        off = 0
        for pi, n in enumerate(lens):
            piece = wav[off:off + n]
            off += n
            fin = pi == n_pieces - 1
            of = S.frontend_step(piece.clone(), oc, cmvn, fin)
            calls += 1
            try:
                with torch.no_grad():
                    rf, rl = frontend(piece[None].clone(), torch.tensor([n]), cache=rc, is_final=fin)
            except RuntimeError as e:
                # final flush of a session that never produced a frame: the reference stacks an EMPTY splice cache and raises
                # (wav_frontend.py:600); the oracle (and the product mirror) return no rows
                assert "non-empty TensorList" in str(e) and fin and of.shape[0] == 0, (si, pi, lens, str(e))
                ref_raised += 1
                break
            r_rows = 0 if rf.numel() == 0 else rf.shape[1]
            if r_rows == 0:
                empty_calls += 1
            if r_rows != of.shape[0]:
                bad += 1
                print("ROWS DIFFER", si, pi, lens, fin, r_rows, of.shape[0])
                break
            if r_rows:
                rows += r_rows
                d = (rf[0] - of).abs().max().item()
                worst = max(worst, d)
                if d > 0:
                    bad += 1
                    print("VALUES DIFFER", si, pi, lens, fin, d)
                    break
    out = dict(sessions=n_sessions, calls=calls, calls_without_output=empty_calls, rows_compared=rows, sessions_with_a_difference=bad, max_abs_diff=worst,
               sessions_where_the_reference_raises_on_an_empty_final_flush=ref_raised)
    print(json.dumps(out))
    return out


if __name__ == "__main__":
    main(int(sys.argv[1]) if len(sys.argv) > 1 else 200)
