"""Shared by the CPU pin of the oracle (tests/test_oracle.py) and the GPU pin of the HIP path (tests/test_parity_gpu.py):
the fixture recorded from the reference's own modules at the headline configuration (oracle/make_golden_full.py)."""
import json
import os

import numpy as np
import torch

from funasr_amd import synth

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def gold(name):
    return np.load(os.path.join(GOLD, name + ".npz"), allow_pickle=False)


def t(a):
    return torch.from_numpy(np.asarray(a))


def full_config_fixture():
    """tests/golden/full_config.npz (oracle/make_golden_full.py): the reference's OWN SANMEncoder + CifPredictorV2 +
    ParaformerSANMDecoder at the headline configuration (50 + 16 blocks, vocabulary 8404) on two 30 s clips, T = 500"""
    g = gold("full_config")
    cfg = json.loads(str(g["cfg"]))
    sd = synth.paraformer_state_dict(cfg, seed=int(g["seed"]), cif_bias=float(g["cif_bias"]))
    feats = t(g["feats_q"]).to(torch.float32) * float(g["feats_scale"])
    return g, cfg, sd, feats, t(g["lens"])


def check_against_full_config(g, res, enc_tol, hid_tol, alpha_tol, flip_gap=1e-4):
    """shared by the CPU (oracle) and GPU (HIP) pins: integer results bit-exact (fire frames, token counts); arg-max ids equal
    except where the REFERENCE's own top-2 logits are closer than `flip_gap`; strided activation samples within the bars.
    `res`: enc [B,T,D], alphas, peaks, token_num, raw_ids (lists), hidden [B,N,D] -- torch CPU tensors. Returns the measured
    differences."""
    B = g["feats_q"].shape[0]
    assert [int(x) for x in res["token_num"]] == g["token_num"].tolist()
    fire_ref = np.floor(g["peaks"]) >= 1
    fire = np.floor(res["peaks"].numpy()[:, : fire_ref.shape[1]]) >= 1
    assert np.array_equal(fire, fire_ref), "CIF fire frames differ from the reference modules'"
    d_alpha = float((res["alphas"][:, : g["alphas"].shape[1]] - t(g["alphas"])).abs().max())
    es, hs = int(g["enc_stride"]), int(g["hidden_stride"])
    d_enc = float((res["enc"][:, ::es][:, : g["enc_rows"].shape[1]] - t(g["enc_rows"])).abs().max())
    last = int(g["olens"].max()) - 1
    d_enc = max(d_enc, float((res["enc"][:, last] - t(g["enc_last_row"])).abs().max()))
    d_hid, flips = 0.0, []
    for b in range(B):
        n = int(g["token_num"][b])
        rows = torch.arange(0, n, hs)
        d_hid = max(d_hid, float((res["hidden"][b, rows] - t(g["hidden_rows"])[b, : len(rows)]).abs().max()))
        gap = g["top2_logits"][b, :n, 0] - g["top2_logits"][b, :n, 1]
        for pos, (x, y) in enumerate(zip(res["raw_ids"][b][:n], g["raw_ids"][b, :n].tolist())):
            if x != y:
                flips.append((b, pos, float(gap[pos])))
    assert d_alpha < alpha_tol and d_enc < enc_tol and d_hid < hid_tol, (d_alpha, d_enc, d_hid)
    assert all(gp < flip_gap for _, _, gp in flips), f"token ids differ where the reference's top-2 gap is not a near-tie: {flips}"
    return dict(alpha=d_alpha, enc=d_enc, hidden=d_hid, flips=flips)
