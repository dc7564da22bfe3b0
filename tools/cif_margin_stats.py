#!/usr/bin/env python3
"""Statistics behind the "CIF fire indices bit-exact" claim (SURVEY 7 / cif_predictor.py:835-846): over many clips, how close
do the prefix sums of the CIF weights come to an integer (the fire decision margin), and how far apart are the GPU's and the
CPU oracle's prefix sums at the same frame? A fire index can only differ where |ps_gpu - ps_cpu| exceeds the margin.

GPU: the product path (frontend -> 50-block encoder -> CifPredictorV2) in its default mode; CPU: oracle/paraformer_oracle.py
on the same clips (fp32 ATen kernels, prefix sums in float64 like cif_wo_hidden_v1). Prints one JSON object.

Round 6 (--ids on both phases): the consequence at TOKEN-ID level. Both sides decode every clip twice -- with the random-init output
layer and with a confident one (synth.confident_output_layer, calibrated ONCE on the GPU's first batch and shipped in the dump, so
both sides use the same weights) -- and the report counts clips whose ids differ, the token error rate between the two paths, and
what happens in exactly those clips whose fire indices differ."""
import argparse, json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

ap = argparse.ArgumentParser()
ap.add_argument("--clips", type=int, default=1024)
ap.add_argument("--seconds", type=float, default=10.0)
ap.add_argument("--gpu-batch", type=int, default=128)
ap.add_argument("--cpu-batch", type=int, default=16)
ap.add_argument("--precision", default=None)
ap.add_argument("--dump", default=None, help="GPU phase only: write alphas / fires / token counts of the clips to this file (torch.save)")
ap.add_argument("--compare", default=None, help="CPU phase only: read a --dump file (made on a GPU box) and run the oracle on the same clips here")
ap.add_argument("--oracle-features", action="store_true", help="GPU phase: feed the ORACLE frontend's features (computed on the host) to the GPU "
                "encoder, so that the comparison isolates the neural path from the fbank FFT's fp32 round-off")
ap.add_argument("--ids", action="store_true", help="also decode: token ids of both paths (random-init and confident output layer), clips with different ids, TER")
ap.add_argument("--first", type=int, default=0, help="--compare: first clip of the slice this process takes")
ap.add_argument("--count", type=int, default=0, help="--compare: clips in the slice (0 = all); slices run as parallel processes")
args = ap.parse_args()

from funasr_amd import synth
from funasr_amd.paraformer import Paraformer
from funasr_amd.wav_frontend import WavFrontend
from oracle import paraformer_oracle as O

def host_cores() -> int:
    """min(affinity mask, cgroup quota): an OpenMP pool larger than the container's CPU quota runs ~100x slower"""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        with open("/sys/fs/cgroup/cpu.max") as f:
            quota, period = f.read().split()[:2]
        if quota != "max":
            n = min(n, max(1, int(float(quota) / float(period))))
    except (OSError, ValueError):
        pass
    return max(1, n)


torch.set_num_threads(host_cores())
cfg = synth.PARAFORMER_LARGE
sd = synth.paraformer_state_dict(cfg, seed=0, cif_bias=synth.BENCH_CIF_BIAS)
shift, scale = synth.synthetic_cmvn(560)
cmvn = torch.stack([shift, scale])
if args.compare:
    blob = torch.load(args.compare)
    args.clips, args.seconds = blob["clips"], blob["seconds"]
    g_alpha, g_fire, g_tok, t_gpu, mode_name = blob["alphas"], blob["fires"], blob["tok"], blob["gpu_seconds"], blob["mode"]
    g_ids_rand, g_ids_conf, conf_layer = blob.get("ids_random"), blob.get("ids_confident"), blob.get("confident_layer")
    if args.ids and g_ids_rand is None:
        raise SystemExit("--ids: the dump was made without --ids")
n = int(args.seconds * 16000)
lo = args.first if args.compare else 0
hi = min(args.clips, lo + args.count) if (args.compare and args.count > 0) else args.clips
clips = {i: synth.speech_like(n, seed=20000 + i) for i in range(lo, hi)}
if not args.compare:
    dev = torch.device("cuda:0")
    model = Paraformer.from_config(cfg)
    model.load_state_dict(sd, strict=False)
    model = model.to(dev).set_precision(args.precision)
    mode_name = model.encoder._mode()
    fe = WavFrontend(cmvn=cmvn, lfr_m=7, lfr_n=6, dither=0.0, device=dev)
    t0 = time.time()
    g_alpha, g_fire, g_tok = [], [], []
    g_ids_rand, g_ids_conf, conf_layer, kept = [], [], None, []
    for b0 in range(0, args.clips, args.gpu_batch):
        batch = [clips[i] for i in range(b0, min(args.clips, b0 + args.gpu_batch))]
        if args.oracle_features:
            feats, flens = O.wav_frontend(batch, cmvn)
            feats = feats.to(dev)
        else:
            feats, flens = fe(torch.stack(batch).to(dev), [n] * len(batch))
        res = model.recognize_features(feats, flens, return_intermediate=True)
        g_alpha.append(res["alphas"].cpu()); g_fire.append(torch.floor(res["peaks"].cpu()) >= 1); g_tok += res["token_num"]
        if args.ids:
            g_ids_rand += res["raw_ids"]
            kept.append((feats, flens))
    g_alpha, g_fire = torch.cat(g_alpha), torch.cat(g_fire)
    t_gpu = time.time() - t0
    if args.ids:
        conf_layer, conf_stats = synth.make_paraformer_confident(model, *kept[0])      # calibrated on the first batch only
        for feats, flens in kept:
            g_ids_conf += model.recognize_features(feats, flens)["raw_ids"]
        del kept
    if args.dump:
        torch.save(dict(clips=args.clips, seconds=args.seconds, alphas=g_alpha, fires=g_fire, tok=g_tok, gpu_seconds=t_gpu,
                        ids_random=g_ids_rand if args.ids else None, ids_confident=g_ids_conf if args.ids else None, confident_layer=conf_layer,
                        mode=mode_name + (" on the oracle frontend's features" if args.oracle_features else "")), args.dump)
        print(json.dumps({"dumped": args.dump, "clips": args.clips, "seconds": args.seconds, "gpu_seconds": round(t_gpu, 2), "mode": mode_name}))
        sys.exit(0)

def edit_distance(a, b):
    prev = list(range(len(b) + 1))
    for i, x in enumerate(a, 1):
        cur = [i]
        for j, y in enumerate(b, 1):
            cur.append(min(prev[j] + 1, cur[j - 1] + 1, prev[j - 1] + (x != y)))
        prev = cur
    return prev[-1]


idstat = {k: dict(clips_with_different_ids=0, token_errors=0, ref_tokens=0, different_ids_among_fire_mismatch_clips=0,
                  different_ids_among_fire_equal_clips=0, clips=[]) for k in ("random_output_layer", "confident_output_layer")}
t0 = time.time()
margins, deltas, a_err = [], [], []
fire_mismatch_clips = token_count_mismatch = at_risk = frames = 0
worst_ratio = float("inf")
with torch.no_grad():
    for b0 in range(lo, hi, args.cpu_batch):
        f, fl = O.wav_frontend([clips[i] for i in range(b0, min(hi, b0 + args.cpu_batch))], cmvn)
        enc, olens = O.sanm_encoder(f, fl, sd, cfg["encoder"], "encoder.")
        embeds, token_num, alphas, peaks = O.cif_predictor(enc, olens, sd, cfg["predictor"], "predictor.")
        c_ids = None
        if args.ids:
            tokr = token_num.round().long()
            logits, hidden = O.paraformer_decoder(enc, olens, embeds, tokr, sd, cfg["decoder"], "decoder.", return_hidden=True)
            Wc, bc = conf_layer["decoder.output_layer.weight"], conf_layer["decoder.output_layer.bias"]
            c_ids = {"random_output_layer": [logits[j, : int(tokr[j])].argmax(-1).tolist() for j in range(alphas.shape[0])],
                     "confident_output_layer": [(hidden[j, : int(tokr[j])] @ Wc.T + bc).argmax(-1).tolist() for j in range(alphas.shape[0])]}
        for j in range(alphas.shape[0]):
            i = b0 + j
            a_c = alphas[j]
            a_g = g_alpha[i, : a_c.numel()]
            ps_c, ps_g = torch.cumsum(a_c.double(), 0), torch.cumsum(a_g.double(), 0)
            fr = ps_c - torch.floor(ps_c)
            m = torch.minimum(fr, 1 - fr)[ps_c > 0.5]
            d = (ps_g - ps_c).abs()[ps_c > 0.5]
            margins.append(m); deltas.append(d); a_err.append((a_g - a_c).abs().max())
            frames += m.numel()
            at_risk += int((m < 4 * d).sum())
            worst_ratio = min(worst_ratio, float((m / d.clamp_min(1e-12)).min()))
            fc = torch.floor(peaks[j]) >= 1
            fire_diff = not torch.equal(fc, g_fire[i, : fc.numel()])
            fire_mismatch_clips += int(fire_diff)
            if c_ids is not None:
                for kind, g_ids in (("random_output_layer", g_ids_rand), ("confident_output_layer", g_ids_conf)):
                    ref, hyp, st = c_ids[kind][j], g_ids[i], idstat[kind]
                    st["ref_tokens"] += len(ref)
                    if ref != hyp:
                        e = edit_distance(ref, hyp)
                        st["clips_with_different_ids"] += 1; st["token_errors"] += e
                        st["different_ids_among_fire_mismatch_clips" if fire_diff else "different_ids_among_fire_equal_clips"] += 1
                        st["clips"].append({"clip": i, "fire_indices_differ": fire_diff, "edit_distance": e, "tokens": len(ref)})
            token_count_mismatch += int(int(token_num[j].round()) != g_tok[i])
t_cpu = time.time() - t0
margins, deltas = torch.cat(margins), torch.cat(deltas)
edges = [0.0, 1e-7, 1e-6, 1e-5, 1e-4, 1e-3, 1e-2, 1e-1, 0.5000001]


def hist(x):
    return {f"[{edges[k]:g}, {edges[k + 1]:g})": int(((x >= edges[k]) & (x < edges[k + 1])).sum()) for k in range(len(edges) - 1)}


print(json.dumps({
    "what": "CIF fire decision margins vs GPU-CPU prefix-sum differences", "clips": hi - lo, "first_clip": lo, "clip_seconds": args.seconds,
    "mode": mode_name, "frames_compared": frames, "tokens": int(sum(g_tok[lo:hi])),
    "clips_with_different_fire_indices": fire_mismatch_clips, "clips_with_different_token_count": token_count_mismatch,
    "alpha_max_abs_diff": float(torch.stack(a_err).max()),
    "prefix_sum_abs_diff": {"max": float(deltas.max()), "median": float(deltas.median()), "p99": float(deltas.quantile(0.99))},
    "margin_to_integer": {"min": float(margins.min()), "p0.1": float(margins.quantile(0.001)), "median": float(margins.median())},
    "frames_with_margin_below_4x_prefix_sum_diff": at_risk, "min_margin_over_diff_ratio": worst_ratio,
    "margin_histogram": hist(margins), "prefix_sum_diff_histogram": hist(deltas),
    "expected_frames_within_diff_of_an_integer": float((2 * deltas).sum()),     # uniform fractional parts: P(margin < d) = 2 d
    "token_ids": ({k: dict(v, token_error_rate=(v["token_errors"] / max(1, v["ref_tokens"]))) for k, v in idstat.items()} if args.ids else None),
    "gpu_seconds": round(t_gpu, 2), "cpu_oracle_seconds": round(t_cpu, 1), "cpu_threads": torch.get_num_threads()}))
