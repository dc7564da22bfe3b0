"""Goldens for the `normalize` modules (normalize_classes UtteranceMVN / GlobalMVN) from the reference's OWN classes
(funasr/models/normalize/utterance_mvn.py, global_mvn.py), imported from /root/reference. TEST INFRASTRUCTURE ONLY.

    python oracle/make_golden_normalize.py        # writes tests/golden/normalize.npz (+ normalize_stats.npy, the GlobalMVN stats file)
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, ROOT)

from oracle import ref_import  # noqa: E402


def main():
    ref_import.install()
    from funasr.models.normalize.global_mvn import GlobalMVN
    from funasr.models.normalize.utterance_mvn import UtteranceMVN

    gold = os.path.join(ROOT, "tests", "golden")
    g = torch.Generator().manual_seed(77)
    B, T, D = 3, 41, 80
    x = torch.randn(B, T, D, generator=g) * 3.0 + torch.randn(1, 1, D, generator=g) * 5.0
    lens = torch.tensor([41, 17, 30], dtype=torch.int32)
    for b in range(B):
        x[b, lens[b]:] = 0                                      # "assumed zero padded" (utterance_mvn.py:62)
    out = dict(x=x.numpy(), lens=lens.numpy())
    for means in (True, False):
        for vars_ in (True, False):
            y, _ = UtteranceMVN(norm_means=means, norm_vars=vars_)(x.clone(), lens)
            out[f"utt_m{int(means)}_v{int(vars_)}"] = y.numpy()
    # a Kaldi-style stats matrix [2, D + 1]: sums | count, sums of squares | 0 (global_mvn.py:42-45)
    rng = np.random.default_rng(5)
    count = 12345.0
    mean = rng.normal(0.0, 4.0, D)
    var = rng.uniform(0.5, 9.0, D)
    stats = np.zeros((2, D + 1), dtype=np.float64)
    stats[0, :-1], stats[0, -1] = mean * count, count
    stats[1, :-1] = (var + mean * mean) * count
    stats_path = os.path.join(gold, "normalize_stats.npy")
    np.save(stats_path, stats)
    for means in (True, False):
        for vars_ in (True, False):
            m = GlobalMVN(stats_path, norm_means=means, norm_vars=vars_)
            y, _ = m(x.clone(), lens)
            out[f"glob_m{int(means)}_v{int(vars_)}"] = y.numpy()
    out["glob_mean"], out["glob_std"] = m.mean.numpy(), m.std.numpy()
    path = os.path.join(gold, "normalize.npz")
    np.savez_compressed(path, **out)
    print(f"wrote {path} ({os.path.getsize(path) / 1024:.1f} KB) and {stats_path}")


if __name__ == "__main__":
    main()
