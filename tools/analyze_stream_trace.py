#!/usr/bin/env python3
"""Where does a hipGraph-replayed streaming step spend its time: in kernels, or between them?
Reads a rocprofv3 --kernel-trace CSV (Start_Timestamp / End_Timestamp in ns), takes the LAST `--steps` repetitions of the step's
kernel sequence (from one stream_embed_kernel to the next, cut at the first gap above --idle-us) and prints per step: wall time from first
start to last end, the sum of kernel durations, the sum and the distribution of the gaps between consecutive kernels, and the
durations per kernel name.   usage: analyze_stream_trace.py kernel_trace.csv [--idle-us 200]"""
import argparse, collections, csv, json, re, statistics

ap = argparse.ArgumentParser()
ap.add_argument("csv")
ap.add_argument("--idle-us", type=float, default=200.0)
a = ap.parse_args()
rows = []
with open(a.csv) as f:
    for r in csv.DictReader(f):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
rows.sort()
# a step = the dispatches from one stream_embed_kernel (first kernel of a step) up to the next one, cut at the first idle gap
starts = [i for i, r in enumerate(rows) if "stream_embed_kernel" in r[2]]
steps = []
for i, j in zip(starts, starts[1:] + [len(rows)]):
    b = [rows[i]]
    for r in rows[i + 1:j]:
        if (r[0] - b[-1][1]) / 1e3 > a.idle_us:
            break
        b.append(r)
    steps.append(b)
sizes = collections.Counter(len(b) for b in steps)
n = sizes.most_common(1)[0][0]
steps = [b for b in steps if len(b) == n][-20:]
out = {"kernels_per_step": n, "steps_analysed": len(steps)}
wall = [(b[-1][1] - b[0][0]) / 1e3 for b in steps]
busy = [sum(e - s for s, e, _ in b) / 1e3 for b in steps]
gaps = [[(y[0] - x[1]) / 1e3 for x, y in zip(b, b[1:])] for b in steps]
out["wall_us_median"] = round(statistics.median(wall), 1)
out["kernel_time_us_median"] = round(statistics.median(busy), 1)
out["gap_time_us_median"] = round(statistics.median(sum(g) for g in gaps), 1)
allg = sorted(g for gs in gaps for g in gs)
out["gap_us"] = {"median": round(statistics.median(allg), 2), "p10": round(allg[len(allg) // 10], 2), "p90": round(allg[len(allg) * 9 // 10], 2), "max": round(allg[-1], 2)}
dur = collections.defaultdict(list)
for b in steps:
    for s, e, k in b:
        dur[re.sub(r"^void ", "", k.replace("(anonymous namespace)::", "")).split("(")[0][:80]].append((e - s) / 1e3)
out["per_kernel"] = {k: {"per_step": round(len(v) / len(steps), 1), "avg_us": round(sum(v) / len(v), 2), "us_per_step": round(sum(v) / len(steps), 1)}
                     for k, v in sorted(dur.items(), key=lambda kv: -sum(kv[1]))}
print(json.dumps(out, indent=1))
