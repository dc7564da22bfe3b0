#!/bin/bash
# rocprofv3 kernel-trace stats of the SenseVoiceSmall workload (BASELINE configs[2]: 128 x 10 s); summary table on stdout
set -u
TAG=${1:-r03}
OUT=$PWD/gpurun_out/prof_sv_$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
CMD="python $PWD/tools/bench_sensevoice.py --modes f16x2 --cpu-clips 0 --steps 4 --warmup 1"
cd /tmp
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats" -o sv -- $CMD > "$OUT/stats.log" 2>&1
cd - > /dev/null
python - "$OUT" <<'PY'
import csv, glob, os, sys
f = glob.glob(os.path.join(sys.argv[1], "stats", "**", "*kernel_stats.csv"), recursive=True)
print("| kernel | calls | total ms | avg us | % |\n|---|---|---|---|---|")
if f:
    for r in list(csv.DictReader(open(f[0])))[:16]:
        n = r["Name"].replace("void ", "").replace("pf::(anonymous namespace)::", "").split("(")[0][:70]
        print(f"| {n} | {r['Calls']} | {float(r['TotalDurationNs'])/1e6:.2f} | {float(r['AverageNs'])/1e3:.1f} | {r['Percentage']} |")
PY
find "$OUT" -name '*kernel_trace.csv' -size +8M -delete
tail -3 "$OUT/stats.log"
