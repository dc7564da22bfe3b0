#!/bin/bash
# SQ issue / wait counters of the f16x2 GEMM kernels (one PMC pass, 8 SQ slots + GRBM), per kernel averages on stdout
set -u
TAG=${1:-r03}
OUT=$PWD/gpurun_out/pmc_gemm_$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
CMD="python $PWD/tools/pmc_gemm.py"
cd /tmp
timeout 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d "$OUT/pmc" -o g -- $CMD > "$OUT/pmc.log" 2>&1
echo "pmc rc=$?" >> "$OUT/pmc.log"
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats" -o g -- $CMD > "$OUT/stats.log" 2>&1
cd - > /dev/null
python - "$OUT" <<'PY' | tee "$OUT/summary.md"
import csv, glob, os, sys
from collections import defaultdict
root = sys.argv[1]
def short(n): return n.replace("void ", "").replace("pf::(anonymous namespace)::", "").split("(")[0][:64]
f = glob.glob(os.path.join(root, "pmc", "**", "*counter_collection.csv"), recursive=True)
agg = defaultdict(lambda: defaultdict(lambda: [0.0, 0]))
if f:
    for r in csv.DictReader(open(f[0])):
        k = short(r.get("Kernel_Name", "?"))
        if "gemm_f16x2" not in k: continue
        c = agg[k][r["Counter_Name"]]; c[0] += float(r["Counter_Value"]); c[1] += 1
dur = {}
s = glob.glob(os.path.join(root, "stats", "**", "*kernel_stats.csv"), recursive=True)
if s:
    for r in csv.DictReader(open(s[0])):
        dur[short(r["Name"])] = float(r["AverageNs"]) / 1e3
names = ["SQ_WAVE_CYCLES", "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_WAIT_INST_LDS", "SQ_VALU_MFMA_BUSY_CYCLES", "SQ_LDS_BANK_CONFLICT", "SQ_LDS_IDX_ACTIVE", "GRBM_GUI_ACTIVE"]
print("| kernel | avg us | " + " | ".join(names) + " |")
print("|---|---|" + "---|" * len(names))
for k, cs in agg.items():
    print(f"| {k} | {dur.get(k, 0):.1f} | " + " | ".join(f"{cs[n][0] / max(cs[n][1], 1):.4g}" for n in names) + " |")
PY
find "$OUT" -name '*kernel_trace.csv' -size +8M -delete
tail -3 "$OUT/pmc.log"
