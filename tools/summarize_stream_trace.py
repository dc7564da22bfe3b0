#!/usr/bin/env python3
"""Condense a rocprofv3 kernel trace of tools/bench_streaming.py (tools/profile_stream.sh) into launches per step, the mean
kernel duration and the gap between consecutive kernels of the replayed steps, and a per-kernel table."""
import collections
import csv
import glob
import re
import sys


def main():
    f = glob.glob(sys.argv[1] + "/trace/**/*kernel_trace.csv", recursive=True)[0]
    rows = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in csv.DictReader(open(f)))
    rows = rows[int(len(rows) * 0.6):]                      # the timed replays are the tail of the run
    by = collections.defaultdict(lambda: [0, 0])
    gap = busy = 0
    for i, (s, e, n) in enumerate(rows):
        k = re.sub(r"\(anonymous namespace\)::", "", n)
        k = re.sub(r"^void ", "", k).split("(")[0][:80]
        by[k][0] += 1
        by[k][1] += e - s
        busy += e - s
        if i:
            gap += max(0, s - rows[i - 1][1])
    steps = max(1, max((c for k, (c, t) in by.items() if "cif_chunk" in k), default=1))
    span = rows[-1][1] - rows[0][0]
    print(f"steps {steps}  launches/step {len(rows) / steps:.1f}  span/step {span / steps / 1e3:.0f} us  kernel time/step "
          f"{busy / steps / 1e3:.0f} us  mean kernel {busy / len(rows) / 1e3:.2f} us  mean gap {gap / len(rows) / 1e3:.2f} us")
    for k, (c, t) in sorted(by.items(), key=lambda kv: -kv[1][1])[:24]:
        print(f"{t / busy * 100:5.1f}%  n/step {c / steps:6.1f}  avg {t / c / 1e3:6.2f} us  per step {t / steps / 1e3:7.1f} us  {k}")


if __name__ == "__main__":
    main()
