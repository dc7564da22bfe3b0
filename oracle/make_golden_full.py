"""The HEADLINE configuration pinned to the reference's own modules.  TEST INFRASTRUCTURE ONLY (build container: needs
/root/reference).        python -m oracle.make_golden_full      ->  tests/golden/full_config.npz

Every other fixture written from the reference's nn.Modules is tiny (3 + 2 blocks, T <= 61, vocabulary 97,
oracle/make_golden.py). This one runs the reference's `SANMEncoder` (funasr/models/sanm/encoder.py:392-461),
`CifPredictorV2` (funasr/models/paraformer/cif_predictor.py:253-314, with cif_v1 :853-908 and cif_wo_hidden_v1 :818-850)
and `ParaformerSANMDecoder` (funasr/models/paraformer/decoder.py:397-449) at synth.PARAFORMER_LARGE -- 50 + 16 blocks,
d_model 512, FFN 2048, vocabulary 8404 -- with the bench's own weights (synth.paraformer_state_dict(seed 0,
BENCH_CIF_BIAS), loaded with strict=True) on the first two 30 s clips of the bench batch (T = 500 LFR frames each), the
way `Paraformer.inference` chains them (funasr/models/paraformer/model.py:534-697: encode -> calc_predictor -> round ->
decoder -> log_softmax -> arg-max).

Inputs: the features are the oracle frontend's (fbank pinned to the reference-vendored kaldi-native-fbank) on
synth.speech_like(480000, seed 0 / 1), rounded to multiples of 2^-10 and stored as int16 so the fixture carries the EXACT
input bits in 1 MB (what feeds the modules is feats_q * 2^-10 in fp32; the modules do not care that it is on a grid).
Outputs kept (bit-exact items in full, activations as a strided sample -- the fixture stays ~1.5 MB):
  olens, alphas [2, 501], cif_peak [2, 501] (fires), token_num, raw arg-max ids of the random-init output layer and, per
  position, the reference's own top-2 logit values (so a test can tell a real difference from a near-tie of a flat
  8404-way distribution), encoder rows ::8 and the last row, decoder hidden rows ::4, the CIF embeddings' rows ::8.
"""
from __future__ import annotations

import json
import os
import sys
import time

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, ROOT)

from funasr_amd import synth  # noqa: E402
from oracle import paraformer_oracle as O  # noqa: E402
from oracle import ref_import  # noqa: E402

GOLD = os.path.join(ROOT, "tests", "golden")
Q = 2.0 ** -10
ENC_STRIDE, HID_STRIDE, EMB_STRIDE = 8, 4, 8


def quantised_features(n_clips: int = 2, seconds: float = 30.0):
    shift, scale = synth.synthetic_cmvn(560)
    cmvn = torch.stack([shift, scale])
    clips = [synth.speech_like(int(seconds * 16000), seed=i) for i in range(n_clips)]     # bench rank 0, clips 0 and 1
    feats, flens = O.wav_frontend(clips, cmvn)
    q = torch.round(feats / Q).clamp_(-32768, 32767).to(torch.int16)
    return q, flens


def build_reference_modules(cfg, sd):
    R = ref_import.modules()
    ec, pc, dc = cfg["encoder"], cfg["predictor"], cfg["decoder"]
    enc = R["SANMEncoder"](input_size=ec["input_size"], output_size=ec["output_size"], attention_heads=ec["attention_heads"],
                           linear_units=ec["linear_units"], num_blocks=ec["num_blocks"], dropout_rate=0.1, input_layer="pe",
                           pos_enc_class=R["SinusoidalPositionEncoder"], normalize_before=True, kernel_size=ec["kernel_size"],
                           sanm_shfit=ec["sanm_shfit"], selfattention_layer_type="sanm").eval()
    pred = R["CifPredictorV2"](idim=pc["idim"], l_order=pc["l_order"], r_order=pc["r_order"], threshold=pc["threshold"],
                               tail_threshold=pc["tail_threshold"]).eval()
    dec = R["ParaformerSANMDecoder"](vocab_size=dc["vocab_size"], encoder_output_size=dc["encoder_output_size"],
                                     attention_heads=dc["attention_heads"], linear_units=dc["linear_units"],
                                     num_blocks=dc["num_blocks"], att_layer_num=dc["att_layer_num"],
                                     kernel_size=dc["kernel_size"], sanm_shfit=dc["sanm_shfit"]).eval()
    enc.load_state_dict({k[len("encoder."):]: v for k, v in sd.items() if k.startswith("encoder.")}, strict=True)
    pred.load_state_dict({k[len("predictor."):]: v for k, v in sd.items() if k.startswith("predictor.")}, strict=True)
    dd = {k[len("decoder."):]: v for k, v in sd.items() if k.startswith("decoder.")}
    dd["embed.0.weight"] = torch.zeros(dc["vocab_size"], dc["encoder_output_size"])       # training-only table (decoder.py:314-317)
    dec.load_state_dict(dd, strict=True)
    return enc, pred, dec


def run_reference(enc, pred, dec, feats, flens):
    """Paraformer.inference's chain (model.py:585-640) on the reference modules"""
    with torch.no_grad():
        eo, el, _ = enc(feats.clone(), flens)
        mask = (torch.arange(eo.shape[1])[None, :] < el[:, None]).float()[:, None, :]
        emb, tok, alphas, peaks = pred(eo, None, mask)
        tok = tok.round().long()
        logits, hidden, _ = dec(eo, el, emb, tok, return_hidden=True, return_both=True)     # decoder.py:439-446
    return dict(enc=eo, olens=el, embeds=emb, token_num=tok, alphas=alphas, peaks=peaks, logits=logits, hidden=hidden)


def main():
    torch.manual_seed(0)
    torch.set_num_threads(max(1, min(8, len(os.sched_getaffinity(0)))))
    cfg = synth.PARAFORMER_LARGE
    sd = synth.paraformer_state_dict(cfg, seed=0, cif_bias=synth.BENCH_CIF_BIAS)
    q, flens = quantised_features()
    feats = q.to(torch.float32) * Q
    enc, pred, dec = build_reference_modules(cfg, sd)
    t0 = time.perf_counter()
    r = run_reference(enc, pred, dec, feats, flens)
    print(f"reference modules: {time.perf_counter() - t0:.1f} s for {feats.shape[0]} x {feats.shape[1]} frames")
    B = feats.shape[0]
    N = int(r["token_num"].max())
    ids = -np.ones((B, N), dtype=np.int64)
    top2 = np.zeros((B, N, 2), dtype=np.float32)
    for b in range(B):
        n = int(r["token_num"][b])
        lp = torch.log_softmax(r["logits"][b, :n], dim=-1)                        # model.py:628-636
        ids[b, :n] = lp.argmax(-1).numpy()
        top2[b, :n] = torch.topk(r["logits"][b, :n], 2, dim=-1).values.numpy()
    last = int(r["olens"].max()) - 1
    path = os.path.join(GOLD, "full_config.npz")
    np.savez_compressed(
        path, feats_q=q.numpy(), feats_scale=np.float64(Q), lens=flens.numpy().astype(np.int32),
        olens=r["olens"].numpy(), alphas=r["alphas"].numpy(), peaks=r["peaks"].numpy(), token_num=r["token_num"].numpy(),
        raw_ids=ids, top2_logits=top2,
        enc_rows=r["enc"][:, ::ENC_STRIDE].numpy(), enc_last_row=r["enc"][:, last].numpy(), enc_stride=np.int64(ENC_STRIDE),
        hidden_rows=r["hidden"][:, ::HID_STRIDE].numpy(), hidden_stride=np.int64(HID_STRIDE),
        embeds_rows=r["embeds"][:, ::EMB_STRIDE].numpy(), embeds_stride=np.int64(EMB_STRIDE),
        logit_rows=r["logits"][:, ::32, ::7].numpy(),
        seed=np.int64(0), cif_bias=np.float64(synth.BENCH_CIF_BIAS), cfg=json.dumps(cfg))
    print(f"wrote {path} ({os.path.getsize(path) / 1024:.0f} KB); tokens {r['token_num'].tolist()}, "
          f"min top-2 gap {float((top2[..., 0] - top2[..., 1])[ids >= 0].min()):.2e}")


if __name__ == "__main__":
    main()
