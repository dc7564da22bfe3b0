#!/usr/bin/env python3
"""Reproducer for the 1-rank vs 2-rank sweep difference: the same clip after different predecessors / workspace poisons."""
import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from funasr_amd import synth, _lib
from funasr_amd.paraformer import Paraformer
from funasr_amd.wav_frontend import WavFrontend
dev = torch.device("cuda:0")
cfg = synth.PARAFORMER_LARGE
model = Paraformer.from_config(cfg)
model.load_state_dict(synth.paraformer_state_dict(cfg, seed=0, cif_bias=synth.BENCH_CIF_BIAS), strict=False)
model = model.to(dev)
for kv in sys.argv[1:]:
    k, _, v = kv.partition("=")
    if k == "precision": model.set_precision(v)
    else: model.encoder.set_option(k, int(v))
sh, sc = synth.synthetic_cmvn(560)
fe = WavFrontend(cmvn=torch.stack([sh, sc]), lfr_m=7, lfr_n=6, dither=0.0, device=dev)
lens = {5: 98859, 24: 70000, 41: 90000, 45: 235000, 4: 104924}
pool = [synth.speech_like(int(14.7 * 16000) + 1, seed=1000 + i) for i in range(16)]
clip = lambda i: pool[i % 16].roll(31 * i)[: lens[i]]
lib = _lib.load()


def run(i, inter=True):
    w = clip(i).to(dev)[None]
    feats, fl = fe(w, [lens[i]])
    r = model.recognize_features(feats, fl, return_intermediate=inter)
    torch.cuda.synchronize()
    return r, feats


def poison(b):
    for mod, fn in ((model.encoder, lib.pf_encoder_debug_poison), (model.decoder, lib.pf_decoder_debug_poison), (model.predictor, lib.pf_predictor_debug_poison)):
        if getattr(mod, "_handle", None) is not None:
            fn(mod._handle, b)


run(45)
base, f0 = run(5)
rows = []
for tag, pre in (("after24", lambda: run(24)), ("after41", lambda: run(41)), ("after4", lambda: run(4)), ("poison7B", lambda: poison(0x7B)), ("poisonFF", lambda: poison(0xFF)),
                 ("poison00", lambda: poison(0)), ("after41_noninter", lambda: run(41, False)), ("repeat", lambda: None)):
    for rep in range(3):
        pre()
        r, f = run(5)
        rows.append({"case": tag, "rep": rep, "feats_equal": bool(torch.equal(f, f0)), "enc_equal": bool(torch.equal(r["enc"], base["enc"])),
                     "enc_maxdiff": float((r["enc"] - base["enc"]).abs().max()), "alphas_equal": bool(torch.equal(r["alphas"], base["alphas"])),
                     "tok": r["token_num"], "ids_equal": r["raw_ids"] == base["raw_ids"]})
        r2 = model.recognize_features(f, [f.shape[1]])          # the packed / production call
        rows[-1]["prod_ids_equal_base"] = r2["raw_ids"] == base["raw_ids"]
for r in rows:
    print(json.dumps(r))
