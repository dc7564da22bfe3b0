"""The `normalize` modules (normalize_classes UtteranceMVN / GlobalMVN; funasr/models/normalize/*.py), applied by Paraformer.encode
between the frontend and the encoder (funasr/models/paraformer/model.py:305-306).

CPU: the oracle's restatement against goldens written by the reference's OWN classes (oracle/make_golden_normalize.py).
GPU: the HIP kernels (csrc/normalize.hip, through the C ABI) against the same goldens -- GlobalMVN bit-exact (element-wise), UtteranceMVN
within float32 round-off of the column sums (the reference sums in float32, the kernel in float64) -- and the model wiring on both routes.
"""
import json
import os

import numpy as np
import pytest
import torch

from funasr_amd import synth
from oracle import paraformer_oracle as O

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
CASES = [(m, v) for m in (True, False) for v in (True, False)]


def gold(name):
    return np.load(os.path.join(GOLD, name + ".npz"), allow_pickle=False)


def t(a):
    return torch.from_numpy(np.asarray(a))


@pytest.mark.parametrize("means,vars_", CASES)
def test_oracle_utterance_mvn_equals_reference(means, vars_):
    g = gold("normalize")
    y = O.utterance_mvn(t(g["x"]), t(g["lens"]), norm_means=means, norm_vars=vars_)
    # the same torch ops as the reference; ATen's float32 column sums depend on the host's thread count / vector width, hence ulps
    assert (y - t(g[f"utt_m{int(means)}_v{int(vars_)}"])).abs().max().item() <= 1e-5


@pytest.mark.parametrize("means,vars_", CASES)
def test_oracle_global_mvn_equals_reference(means, vars_):
    g = gold("normalize")
    y = O.global_mvn(t(g["x"]), t(g["lens"]), t(g["glob_mean"]), t(g["glob_std"]), norm_means=means, norm_vars=vars_)
    assert np.array_equal(y.numpy(), g[f"glob_m{int(means)}_v{int(vars_)}"])


def test_registry_holds_the_reference_keys():
    from funasr_amd import normalize  # noqa: F401
    from funasr_amd.register import tables
    assert tables.normalize_classes["UtteranceMVN"].__name__ == "UtteranceMVN"
    assert tables.normalize_classes["GlobalMVN"].__name__ == "GlobalMVN"
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        tables.normalize_classes["UtteranceMVN"]()(torch.zeros(1, 4, 8), [4])


@pytest.mark.gpu
@pytest.mark.parametrize("means,vars_", CASES)
def test_hip_utterance_mvn_vs_reference_golden(cuda, means, vars_):
    from funasr_amd.normalize import UtteranceMVN
    g = gold("normalize")
    x = t(g["x"]).to(cuda)
    y, lens = UtteranceMVN(norm_means=means, norm_vars=vars_)(x, t(g["lens"]))
    assert y.data_ptr() == x.data_ptr()                          # in place, like the reference at inference
    ref = t(g[f"utt_m{int(means)}_v{int(vars_)}"])
    d = (y.cpu() - ref).abs().max().item()
    assert d <= 2e-5, d                                          # values are O(10): a few float32 ulp of the column sums
    if not means:                                                # rows past an utterance's length stay zero without the mean shift
        for b, n in enumerate(g["lens"].tolist()):
            assert (y[b, n:] == 0).all()


@pytest.mark.gpu
@pytest.mark.parametrize("means,vars_", CASES)
def test_hip_global_mvn_bit_exact_vs_reference_golden(cuda, means, vars_):
    from funasr_amd.normalize import GlobalMVN
    g = gold("normalize")
    m = GlobalMVN(os.path.join(GOLD, "normalize_stats.npy"), norm_means=means, norm_vars=vars_)
    y, _ = m(t(g["x"]).to(cuda), t(g["lens"]))
    assert np.array_equal(y.cpu().numpy(), g[f"glob_m{int(means)}_v{int(vars_)}"])


@pytest.mark.gpu
@pytest.mark.parametrize("one_call", [True, False])
def test_paraformer_applies_normalize_in_front_of_the_encoder(cuda, one_call):
    """Paraformer(normalize="UtteranceMVN") on raw features == the same weights without `normalize` on features normalised beforehand
    (model.py:304-306), on the one-call route (pf_paraformer_begin / _finish) and on the module-by-module chain."""
    from funasr_amd.paraformer import Paraformer
    cfg = synth.tiny(synth.PARAFORMER_LARGE)
    sd = synth.paraformer_state_dict(cfg, seed=3, cif_bias=synth.BENCH_CIF_BIAS)

    def build(**kw):
        ec = dict(cfg["encoder"]); input_size = ec.pop("input_size")
        dc = dict(cfg["decoder"]); vocab = dc.pop("vocab_size"); dc.pop("encoder_output_size", None)
        m = Paraformer(encoder="SANMEncoder", encoder_conf=dict(ec, input_layer="pe"), decoder="ParaformerSANMDecoder", decoder_conf=dc,
                       predictor="CifPredictorV2", predictor_conf=dict(cfg["predictor"]), ctc_weight=0.0, input_size=input_size,
                       vocab_size=vocab, **kw)
        m.load_state_dict(sd, strict=False)
        m._one_call = one_call
        return m.to(cuda)

    g = torch.Generator().manual_seed(9)
    feats = torch.randn(3, 50, 560, generator=g) * 0.7 + 0.4
    lens = torch.tensor([50, 31, 44], dtype=torch.int32)
    for b in range(3):
        feats[b, lens[b]:] = 0
    plain, normed = build(), build(normalize="UtteranceMVN", normalize_conf=dict(norm_means=True, norm_vars=True))
    assert type(normed.normalize).__name__ == "UtteranceMVN"
    from funasr_amd.normalize import UtteranceMVN
    pre, _ = UtteranceMVN(norm_means=True, norm_vars=True)(feats.clone().to(cuda), lens)       # the kernel the golden tests above check
    assert (pre.cpu() - O.utterance_mvn(feats, lens, True, True)).abs().max().item() <= 2e-5
    ref = plain.recognize_features(pre, lens)
    got = normed.recognize_features(feats.clone().to(cuda), lens)
    raw = plain.recognize_features(feats.clone().to(cuda), lens)
    assert got["token_num"] == ref["token_num"] and got["ids"] == ref["ids"]
    assert raw["ids"] != got["ids"]                              # the normalisation is not a no-op on these features
