#!/usr/bin/env python3
"""Condense rocprofv3 output of tools/profile.sh into one markdown summary (per-kernel time table from
--kernel-trace --stats, per-kernel HBM bytes per launch from the FETCH_SIZE / WRITE_SIZE PMC passes).

HBM bytes follow /opt/skills/guides/MI355X_MICROARCH.md (HBM section): FETCH_SIZE / WRITE_SIZE are reported in KiB
(x1024); on gfx950 FETCH_SIZE under-counts wide coalesced reads by exactly 2x, so the read side is doubled
("corrected"); WRITE_SIZE is taken as reported (uncalibrated)."""
import csv
import glob
import os
import sys
from collections import defaultdict


def find(root, pattern):
    hits = glob.glob(os.path.join(root, "**", pattern), recursive=True)
    return hits[0] if hits else None


def short(name):
    for pre in ("void ", "pf::(anonymous namespace)::", "(anonymous namespace)::"):
        name = name.replace(pre, "")
    name = name.split("(")[0].replace("pf::", "")
    return name[:70]


def stats_table(root):
    f = find(os.path.join(root, "stats"), "*kernel_stats.csv")
    rows = []
    if not f:
        return rows
    with open(f) as fh:
        for r in csv.DictReader(fh):
            rows.append(r)
    return rows


def pmc_table(root, sub, counter):
    f = find(os.path.join(root, sub), "*counter_collection.csv")
    agg = defaultdict(lambda: [0.0, 0])
    if not f:
        return agg
    with open(f) as fh:
        for r in csv.DictReader(fh):
            if r.get("Counter_Name") != counter:
                continue
            k = short(r.get("Kernel_Name", "?"))
            agg[k][0] += float(r.get("Counter_Value", 0.0))
            agg[k][1] += 1
    return agg


def main():
    root, tag = sys.argv[1], sys.argv[2]
    cmd = " ".join(sys.argv[3:]) or "python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-secondary --no-bf16"
    print(f"# rocprofv3 summary `{tag}`  ({cmd})\n")
    rows = stats_table(root)
    print("## kernel time (rocprofv3 --kernel-trace --stats)\n")
    print("| kernel | calls | total ms | avg us | % |")
    print("|---|---|---|---|---|")
    # (bench.py's live power-limited-peak measurement launches tools/micro/mfma_peak.so's kernel AFTER the timed region: not a kernel
    #  of the step, listed apart; the percentages below are of the remaining kernels)
    tool = [r for r in rows if short(r.get("Name", "")).startswith("mfma_peak_kernel")]
    rows = [r for r in rows if r not in tool]
    total = sum(float(r.get("TotalDurationNs", 0)) for r in rows) or 1.0
    for r in rows[:24]:
        name = short(r.get("Name", r.get("KernelName", "?")))
        calls = r.get("Calls", "?")
        tot = float(r.get("TotalDurationNs", 0)) / 1e6
        avg = float(r.get("AverageNs", 0)) / 1e3
        print(f"| {name} | {calls} | {tot:.2f} | {avg:.1f} | {100.0 * float(r.get('TotalDurationNs', 0)) / total:.2f} |")
    print("\n(absmax_kernel, rowl1_bound_kernel and most split2_kernel launches are the one-time weight preparation of the f16x2 mode at model "
          "load -- exponent scans, a-priori output bounds, plane splits -- outside the timed region; a step launches absmax_kernel twice)")
    for r in tool:
        print(f"\n(outside the step: {short(r.get('Name', '?'))}, {r.get('Calls', '?')} launches, {float(r.get('TotalDurationNs', 0)) / 1e6:.0f} ms -- "
              f"bench.py's live measurement of the power-limited MFMA peak, run after the timed region)")
    fetch = pmc_table(root, "pmc_fetch", "FETCH_SIZE")
    write = pmc_table(root, "pmc_write", "WRITE_SIZE")
    print("\n## HBM traffic per launch (separate --pmc passes; KiB counters x1024; FETCH_SIZE doubled per the gfx950 note)\n")
    print("| kernel | launches | read MB/launch (corrected) | write MB/launch |")
    print("|---|---|---|---|")
    keys = sorted(set(fetch) | set(write), key=lambda k: -(fetch[k][0] if k in fetch else 0))
    traffic = {}
    keys = [k for k in keys if not k.startswith("mfma_peak_kernel")]
    for k in keys[:24]:
        fr = fetch[k][0] * 1024 * 2 / max(fetch[k][1], 1) / 1e6 if k in fetch else float("nan")
        wr = write[k][0] * 1024 / max(write[k][1], 1) / 1e6 if k in write else float("nan")
        n = fetch[k][1] if k in fetch else write[k][1]
        print(f"| {k} | {n} | {fr:.3f} | {wr:.3f} |")
        if k in fetch and k in write:
            traffic[k] = dict(launches=n, read_bytes_per_launch=fr * 1e6, write_bytes_per_launch=wr * 1e6)
    # per-step figure: ONLY this library's kernels (the output-layer calibration outside the clock runs torch / rocBLAS kernels), and
    # only meaningful when the run held ONE mode (tools/profile.sh passes --no-secondary --no-bf16; r04q mixed five modes and two
    # other workloads into one quotient). Steps = attention_f16x2 launches / 66 (50 self- + 16 cross-attentions per offline step).
    foreign = ("at::", "Cijk_", "__amd_rocclr", "rocprim", "hipcub", "elementwise", "void at")
    own = lambda k: not any(k.startswith(f) or f in k[:24] for f in foreign) and not k.startswith("mfma_peak_kernel")
    tot_r = sum(fetch[k][0] for k in fetch if own(k)) * 1024 * 2 / 1e9
    tot_w = sum(write[k][0] for k in write if own(k)) * 1024 / 1e9
    oth = (sum(fetch[k][0] for k in fetch if not own(k)) * 2 + sum(write[k][0] for k in write if not own(k))) * 1024 / 1e9
    n_attn = max((fetch[k][1] for k in fetch if k.startswith("attention_f16x2")), default=0)
    modes = sorted({k.split("<")[0] for k in fetch if k.startswith(("attention_f32", "attention_bf16", "attention_split3", "gemm_split3"))})
    print(f"\nthis library's kernels: {tot_r:.1f} GB read (corrected) + {tot_w:.1f} GB written (torch / rocBLAS / runtime kernels outside "
          f"the clock: {oth:.1f} GB more); the run holds {n_attn} attention_f16x2 launches = {n_attn / 66.0:.2f} steps of 50 self- + 16 "
          f"cross-attentions => {(tot_r + tot_w) / max(n_attn / 66.0, 1e-9):.1f} GB of HBM traffic per step"
          + (f"  -- NOT a per-step figure of the main mode: kernels of other modes are in this run ({', '.join(modes)})" if modes else ""))
    # matrix-pipe occupancy: SQ_VALU_MFMA_BUSY_CYCLES counts cycles summed over the SIMDs (32 per v_mfma_*_32x32x16);
    # against duration x 1024 SIMDs x the 2.4 GHz peak clock it is the fraction of the dense-MFMA peak the kernel used
    mfma = pmc_table(root, "pmc_mfma", "SQ_VALU_MFMA_BUSY_CYCLES")
    if mfma:
        avg_ns = {short(r.get("Name", "?")): float(r.get("AverageNs", 0)) for r in rows}
        print("\n## matrix pipe (separate --pmc pass: SQ_VALU_MFMA_BUSY_CYCLES, summed over the SIMDs)\n")
        print("| kernel | launches | MFMA busy cycles / launch | avg us (stats pass) | busy / (1024 SIMDs x 2.4 GHz x duration) |")
        print("|---|---|---|---|---|")
        for k in sorted(mfma, key=lambda k: -mfma[k][0])[:12]:
            per = mfma[k][0] / max(mfma[k][1], 1)
            ns = avg_ns.get(k, 0.0)
            frac = per / (1024 * 2.4 * ns) if ns > 0 else float("nan")
            print(f"| {k} | {mfma[k][1]} | {per:.3e} | {ns / 1e3:.1f} | {frac:.3f} |")
    import json
    out = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "profiles", f"{tag}_traffic.json")
    try:
        with open(out, "w") as f:
            json.dump(dict(tag=tag, note="FETCH_SIZE KiB x1024 x2 (gfx950 under-count of wide reads), WRITE_SIZE KiB x1024; "
                                        "separate rocprofv3 --pmc passes of `bench.py --steps 3 --warmup 1 --no-cpu-baseline`",
                           kernels=traffic), f, indent=1)
    except OSError:
        pass


if __name__ == "__main__":
    main()
