#!/usr/bin/env python3
"""Shape fuzz on the GPU for the timestamp / hotword models: random batch sizes, frame counts, ragged lengths and hotword
lists through BiCifParaformer / SeacoParaformer (HIP) against the CPU oracles -- token ids, CIF fires, upsampled weights
and the fires of the timestamp head. Not part of the test run (needs the MI355X): `python tools/fuzz_gpu_bicif_vs_oracle.py
[seed] [cases]`."""
import copy
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from funasr_amd import synth                                    # noqa: E402
from funasr_amd.seaco_paraformer import SeacoParaformer         # noqa: E402
from oracle import seaco_oracle as SO                           # noqa: E402

V3 = dict(idim=512, threshold=1.0, l_order=1, r_order=1, tail_threshold=0.45, smooth_factor=1.0, noise_threshold=0.0,
          smooth_factor2=0.25, noise_threshold2=0.01, upsample_times=3, use_cif1_cnn=False, upsample_type="cnn_blstm")


def build(cfg, no_bias):
    ec = dict(cfg["encoder"])
    input_size = ec.pop("input_size")
    dc = dict(cfg["decoder"])
    vocab = dc.pop("vocab_size")
    dc.pop("encoder_output_size", None)
    sc = {k: v for k, v in cfg["seaco_decoder"].items() if k not in ("vocab_size", "encoder_output_size", "att_layer_num")}
    return SeacoParaformer(encoder="SANMEncoder", encoder_conf=dict(ec, input_layer="pe"), decoder="ParaformerSANMDecoder",
                           decoder_conf=dc, seaco_decoder="ParaformerSANMDecoder",
                           seaco_decoder_conf=dict(sc, use_output_layer=False, wo_input_layer=True), predictor="CifPredictorV3",
                           predictor_conf=dict(cfg["predictor"]), ctc_weight=0.0, input_size=input_size, vocab_size=vocab,
                           inner_dim=512, bias_encoder_type="lstm", NO_BIAS=no_bias)


def fires(x, thr=1.0 - 1e-4):
    return [torch.nonzero(r >= thr).flatten().tolist() for r in x]


def main():
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(int(sys.argv[1]) if len(sys.argv) > 1 else 0)
    n_cases = int(sys.argv[2]) if len(sys.argv) > 2 else 8
    bad, worst = 0, 0.0
    for ci in range(n_cases):
        vocab = int(torch.randint(30, 200, (1,), generator=g))
        cfg = synth.tiny(synth.PARAFORMER_LARGE, enc_blocks=int(torch.randint(1, 3, (1,), generator=g)),
                         dec_blocks=int(torch.randint(1, 3, (1,), generator=g)), vocab=vocab)
        cfg["predictor"] = dict(V3, upsample_type=("cnn_blstm", "cnn")[ci % 4 == 3])
        cfg["seaco_decoder"] = copy.deepcopy(SO.SEACO_DECODER)
        no_bias = 5
        sd = SO.seaco_state_dict(cfg, 700 + ci, no_bias)
        if cfg["predictor"]["upsample_type"] == "cnn":
            sd = {k: v for k, v in sd.items() if ".blstm." not in k}
            sd["predictor.cif_output2.weight"] = sd["predictor.cif_output2.weight"][:, :512].contiguous()
        model = build(cfg, no_bias)
        model.load_state_dict(sd, strict=True)
        model = model.to(dev)
        B = int(torch.randint(1, 7, (1,), generator=g))
        T = int(torch.randint(3, 160, (1,), generator=g))
        lens = torch.randint(1, T + 1, (B,), generator=g, dtype=torch.int32)
        lens[0] = T
        x = torch.randn(B, T, 560, generator=g) * 0.7
        for b in range(B):
            x[b, lens[b]:] = 0
        hw = None
        if ci % 2:
            hw = [torch.randint(3, vocab, (int(torch.randint(1, 6, (1,), generator=g)),), generator=g).tolist()
                  for _ in range(int(torch.randint(1, 12, (1,), generator=g)))] + [[1]]
        ref = SO.seaco_greedy(x, lens, hw, sd, cfg, no_bias)
        for mode in ("fp32", "bf16x3"):
            model.set_precision(mode)
            model.hotword_list = hw
            res = model.recognize_features(x.to(dev), lens)
            ok = res["raw_ids"] == ref["raw_ids"] and res["token_num"] == ref["token_num"].tolist()
            if max(res["token_num"]) >= 1:
                ua = res["us_alphas_host"]
                err = max(float((ua[b, : 3 * int(lens[b])] - ref["us_alphas"][b, : 3 * int(lens[b])]).abs().max()) for b in range(B))
                worst = max(worst, err)
                ok = ok and err < 1e-4 and fires(res["us_peaks_host"]) == fires(ref["us_peaks"])
            if not ok:
                bad += 1
                print(f"case {ci} mode {mode} B={B} T={T} lens={lens.tolist()} hotwords={None if hw is None else len(hw)} differs")
    print(f"{n_cases} cases: failures {bad}, max |us_alphas - oracle| {worst:.1e}")


if __name__ == "__main__":
    main()
