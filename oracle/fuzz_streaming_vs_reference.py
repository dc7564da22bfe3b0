"""Fuzz the streaming oracle against the REFERENCE's own ParaformerStreaming.inference (build container only; TEST
INFRASTRUCTURE): random chunk geometries, look-back settings, clip lengths and weight seeds; per chunk the oracle -- driven
with the online features the reference's frontend produced -- must return the reference's token ids, position counter,
encoder window and carried CIF state. The committed fixtures (make_golden_streaming.py) pin six sessions; this sweeps the
geometry space the GPU streaming tests rely on the oracle for (tests/test_streaming_gpu.py, tests/_stream_f16x2_cases.py).

    python -m oracle.fuzz_streaming_vs_reference [n_cases]
"""
import json
import os
import sys

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
from funasr_amd import synth  # noqa: E402
from oracle import make_golden_streaming as G  # noqa: E402
from oracle import streaming_oracle as S  # noqa: E402


def main(n_cases=12):
    model, frontend, cfg, enc_conf = G.build()
    g = torch.Generator().manual_seed(2024)
    worst = dict(enc=0.0, cif_alpha=0.0, cif_hidden=0.0)
    bad, chunks, sessions = 0, 0, []
    for ci in range(n_cases):
        left = int(torch.randint(0, 2, (1,), generator=g)) * 5
        cur = int(torch.randint(4, 21, (1,), generator=g))
        right = int(torch.randint(0, cur // 2 + 1, (1,), generator=g))
        enc_lb, dec_lb = int(torch.randint(0, 5, (1,), generator=g)), int(torch.randint(0, 3, (1,), generator=g))
        seed = 500 + ci
        sd = synth.paraformer_state_dict(cfg, seed=seed, cif_bias=float(torch.rand(1, generator=g)) * 0.8)
        sd_ref = dict(sd)
        sd_ref["decoder.embed.0.weight"] = torch.zeros(cfg["decoder"]["vocab_size"], 512)
        model.load_state_dict(sd_ref, strict=False)
        n_total = int(torch.randint(3 * cur * 960, 7 * cur * 960, (1,), generator=g))
        wav = synth.speech_like(n_total, seed=seed)
        wav = (wav * 32768.0).round().clamp(-32768, 32767) / 32768.0
        n1 = int(torch.randint(960, n_total - 960, (1,), generator=g))
        chunk = [left, cur, right]
        records = G.run(model, frontend, enc_conf, wav, n1, chunk, enc_lb, dec_lb)
        st = S.model_init(cfg, tuple(chunk), enc_lb, dec_lb)
        ok = True
        for r in records:
            if r["tail"]:
                st["tail_chunk"] = True
                feats = st["feats"]
            else:
                feats = torch.from_numpy(r["feats"])
            trace = []
            with torch.no_grad():
                ids = S.generate_chunk(feats, st, sd, cfg, r["is_final"], trace)
            chunks += 1
            if ids != r["tokens"] or st["start_idx"] != r["start_idx"]:
                ok = False
            worst["enc"] = max(worst["enc"], (trace[0]["enc"] - torch.from_numpy(r["enc"])).abs().max().item())
            worst["cif_alpha"] = max(worst["cif_alpha"], abs(float(st["cif_alphas"].reshape(-1)[0]) - float(r["cif_alphas"][0])))
            worst["cif_hidden"] = max(worst["cif_hidden"], (st["cif_hidden"].reshape(-1) - torch.from_numpy(r["cif_hidden"])).abs().max().item())
        bad += 0 if ok else 1
        sessions.append(dict(chunk=chunk, enc_lb=enc_lb, dec_lb=dec_lb, chunks=len(records), tokens=sum(len(r["tokens"]) for r in records), ok=ok))
    out = dict(sessions=len(sessions), chunks=chunks, sessions_with_different_ids_or_positions=bad, max_abs_diff=worst, detail=sessions)
    print(json.dumps(out))
    return out


if __name__ == "__main__":
    main(int(sys.argv[1]) if len(sys.argv) > 1 else 12)
