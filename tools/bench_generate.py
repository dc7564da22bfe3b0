#!/usr/bin/env python
"""`AutoModel.generate` over a directory of wav FILES, full-size Paraformer-large (random-init weights, synthetic speech-like clips with
AISHELL-like durations): the user-facing loop (funasr/auto/auto_model.py:790-840 -- load, features, forward, text, one batch after the
other) against this package's overlapped form of the same loop (funasr_amd/auto_model.py: begin(i + 1) | launch(i) | end(i - 1)).
Both return the same records; the line says how many audio seconds per second each delivers INCLUDING file reading, resampling-free
decoding, padding, the H2D copy, features, text.

    python tools/bench_generate.py --clips 2000 --batch-size 64 [--dir /tmp/clips]
"""
import argparse
import json
import os
import shutil
import sys
import tempfile
import time

import torch
import yaml

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))


def model_dir(path, cfg, model="Paraformer"):
    ec, dc, pc = cfg["encoder"], cfg["decoder"], cfg["predictor"]
    vocab = ["<blank>", "<s>", "</s>"] + [chr(0x4E00 + i) for i in range(dc["vocab_size"] - 4)] + ["<unk>"]
    conf = {
        "model": model,
        "model_conf": {"ctc_weight": 0.0, "predictor_weight": 1.0, "predictor_bias": 1},
        "encoder": "SANMEncoder",
        "encoder_conf": {"output_size": ec["output_size"], "attention_heads": ec["attention_heads"], "linear_units": ec["linear_units"],
                         "num_blocks": ec["num_blocks"], "input_layer": "pe", "pos_enc_class": "SinusoidalPositionEncoder",
                         "normalize_before": True, "kernel_size": ec["kernel_size"], "sanm_shfit": ec["sanm_shfit"],
                         "selfattention_layer_type": "sanm"},
        "decoder": "ParaformerSANMDecoder",
        "decoder_conf": {"attention_heads": dc["attention_heads"], "linear_units": dc["linear_units"], "num_blocks": dc["num_blocks"],
                         "att_layer_num": dc["att_layer_num"], "kernel_size": dc["kernel_size"], "sanm_shfit": dc["sanm_shfit"]},
        "predictor": "CifPredictorV2" if model == "Paraformer" else "CifPredictorV3",
        "predictor_conf": dict({"idim": pc["idim"], "threshold": 1.0, "l_order": 1, "r_order": 1, "tail_threshold": 0.45},
                               **({} if model == "Paraformer" else {"upsample_times": 3, "use_cif1_cnn": False, "upsample_type": "cnn_blstm"})),
        "frontend": "WavFrontend",
        "frontend_conf": {"fs": 16000, "window": "hamming", "n_mels": 80, "frame_length": 25, "frame_shift": 10, "lfr_m": 7, "lfr_n": 6},
        "tokenizer": "CharTokenizer",
        "tokenizer_conf": {"unk_symbol": "<unk>", "split_with_space": True},
    }
    os.makedirs(path, exist_ok=True)
    with open(os.path.join(path, "config.yaml"), "w", encoding="utf-8") as f:
        yaml.safe_dump(conf, f, allow_unicode=True)
    with open(os.path.join(path, "tokens.json"), "w", encoding="utf-8") as f:
        json.dump(vocab, f, ensure_ascii=False)
    shutil.copy(os.path.join(ROOT, "tests", "golden", "am.mvn"), os.path.join(path, "am.mvn"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--clips", type=int, default=2000)
    ap.add_argument("--batch-size", type=int, default=64, help="clips per batch (AutoModel batch_size)")
    ap.add_argument("--dir", default=None)
    ap.add_argument("--repeats", type=int, default=2)
    ap.add_argument("--batch-size-rows", type=int, default=0, help="> 0: generate(batch_size_rows=N) over the list in its ORIGINAL (unsorted) "
                    "order -- this package's own option: batches planned by encoder rows from the WAV headers, records back in input order")
    ap.add_argument("--model", default="Paraformer", choices=["Paraformer", "BiCifParaformer"], help="BiCifParaformer: text + token timestamps "
                    "on every call (the model behind the paraformer-zh alias), CifPredictorV3 with random timestamp-head weights")
    ap.add_argument("--profile", action="store_true", help="cProfile of one overlapped pass (top functions by own time) to stderr")
    args = ap.parse_args()

    from funasr_amd import synth
    from funasr_amd.auto_model import AutoModel
    from sweep import durations                                     # the AISHELL-like duration draw of the corpus sweep
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from _model_dir import write_wav

    work = args.dir or tempfile.mkdtemp(prefix="pf_generate_")
    mdir = os.path.join(work, "model")
    cfg = synth.PARAFORMER_LARGE
    model_dir(mdir, cfg, args.model)
    durs = durations(args.clips)
    pool = [synth.speech_like(int(14.7 * 16000) + 1, seed=1000 + i) for i in range(16)]
    paths = []
    t0 = time.perf_counter()
    for i, d in enumerate(durs):
        p = os.path.join(work, f"clip{i:05d}.wav")
        write_wav(p, pool[i % 16].roll(31 * i)[: int(d * 16000)])
        paths.append(p)
    # the reference sorts nothing here: generate() takes the list in its order; sorted by duration so that a batch pads little
    # (what examples/aishell's data preparation achieves with its length-sorted jsonl)
    if args.batch_size_rows <= 0:
        order = sorted(range(len(paths)), key=lambda i: -durs[i])
        paths = [paths[i] for i in order]
    total_s = float(sum(durs))
    print(f"[generate] {args.clips} wav files, {total_s / 3600:.2f} h, written in {time.perf_counter() - t0:.1f} s under {work}", file=sys.stderr)

    am = AutoModel(model=mdir, device="cuda:0", batch_size=args.batch_size, disable_pbar=True)
    missing, _ = am.model.load_state_dict(synth.paraformer_state_dict(cfg, seed=0, cif_bias=synth.BENCH_CIF_BIAS), strict=False)
    if args.model != "Paraformer":
        g = torch.Generator().manual_seed(5)
        sd = am.model.state_dict()
        for k in missing:
            if k.startswith("predictor."):
                sd[k].copy_(torch.randn(sd[k].shape, generator=g) * 0.05)
    am.model.to("cuda:0")
    am.generate(input=paths[: 2 * args.batch_size])                  # warm-up: buffers at their final size, files in the page cache
    am.generate(input=paths[-2 * args.batch_size:], pipeline=False)
    out = {"metric": f"AutoModel.generate over wav files, audio-seconds/s ({args.model}-large, f16x2)", "clips": args.clips,
           "audio_hours": round(total_s / 3600, 2), "batch_size": args.batch_size, "runs": []}
    ref = None
    for r in range(args.repeats):
        rows_kw = {"batch_size_rows": args.batch_size_rows} if args.batch_size_rows > 0 else {}
        for name, kw in (("overlapped", dict(rows_kw)), ("plain loop", dict(rows_kw, pipeline=False))) + (
                (("count batches, unsorted list", {}),) if rows_kw else ()):
            torch.cuda.synchronize()
            t = time.perf_counter()
            res = am.generate(input=paths, **kw)
            torch.cuda.synchronize()
            dt = time.perf_counter() - t
            ref = ref if ref is not None else res
            out["runs"].append({"loop": name, "wall_s": round(dt, 3), "audio_s_per_s": round(total_s / dt, 1), "records_equal_first_run": res == ref, "texts_equal_first_run": sum(a.get("text") == b.get("text") for a, b in zip(res, ref)),
                                "last_batch": {k: am.speed_stats.get(k) for k in ("load_data", "extract_feat", "forward")}})
    if args.profile:
        import cProfile
        import pstats
        pr = cProfile.Profile()
        pr.enable()
        am.generate(input=paths)
        pr.disable()
        pstats.Stats(pr, stream=sys.stderr).sort_stats("tottime").print_stats(28)
    best = {n: max(x["audio_s_per_s"] for x in out["runs"] if x["loop"] == n) for n in {x["loop"] for x in out["runs"]}}
    out["batch_size_rows"] = args.batch_size_rows or None
    out["count_batches_unsorted"] = best.get("count batches, unsorted list")
    out["value"] = best["overlapped"]
    out["plain_loop"] = best["plain loop"]
    out["gain"] = round(best["overlapped"] / best["plain loop"], 3)
    out["tokens"] = sum(len(r.get("text", "").split()) for r in ref)
    print(json.dumps(out), flush=True)
    if args.dir is None:
        shutil.rmtree(work, ignore_errors=True)


if __name__ == "__main__":
    main()
