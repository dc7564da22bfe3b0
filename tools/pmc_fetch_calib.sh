#!/bin/bash
# usage (on the GPU box): tools/pmc_fetch_calib.sh TAG  ->  gpurun_out/pmc_calib_TAG/summary.txt
set -u
TAG=${1:-r04}
OUT=$PWD/gpurun_out/pmc_calib_$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
CMD="python $PWD/tools/pmc_fetch_calib.py"
cd /tmp
timeout 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d "$OUT/pmc" -o c -- $CMD > "$OUT/pmc.log" 2>&1
echo "rc=$?" >> "$OUT/pmc.log"
cd - > /dev/null
python - "$OUT" <<'PY' | tee "$OUT/summary.txt"
import csv, glob, json, os, sys
from collections import defaultdict
root = sys.argv[1]
known = {}
for line in open(os.path.join(root, "pmc.log"), errors="replace"):
    if line.startswith("KNOWN "): known = json.loads(line[6:])
f = glob.glob(os.path.join(root, "pmc", "**", "*counter_collection.csv"), recursive=True)
agg = defaultdict(lambda: [0.0, 0])
for r in csv.DictReader(open(f[0])):
    if r["Counter_Name"] != "FETCH_SIZE": continue
    k = r["Kernel_Name"].replace("void ", "").replace("pf::(anonymous namespace)::", "").split("(")[0][:70] + " grid=" + r.get("Grid_Size", r.get("Grid_Size_X", "?"))
    agg[k][0] += float(r["Counter_Value"]); agg[k][1] += 1
print("raw FETCH_SIZE (KiB x 1024) per launch, by kernel and grid:")
for k, (v, n) in sorted(agg.items(), key=lambda kv: -kv[1][0]):
    if v / n * 1024 > 2e7: print(f"  {v / n * 1024 / 1e6:9.1f} MB  n={n:3d}  {k}")
print("known read volumes (MB):", {k: round(v["read_bytes"] / 1e6, 1) for k, v in known.items()})
PY
