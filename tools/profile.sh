#!/bin/bash
# Run on the GPU box (through gpurun): rocprofv3 kernel-trace stats + separate PMC passes for HBM traffic and the
# matrix-pipe busy cycles of the bench workload. Outputs under gpurun_out/prof_$TAG; tools/summarize_prof.py condenses them into profiles/.
# usage: tools/profile.sh TAG [bench args...]
set -u
TAG=${1:-r01}; shift || true
OUT=$PWD/gpurun_out/prof_$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
# main mode only, ONE stream (kernel durations next to another stream's kernels are not the kernels'): a profile that mixes the other
# arithmetic modes and the secondary workloads is not evidence for the headline line
BENCH="python $PWD/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-secondary --no-bf16 --no-decoder-stream $*"
cd /tmp
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats" -o bench -- $BENCH > "$OUT/stats.log" 2>&1
echo "stats rc=$?" >> "$OUT/stats.log"
timeout 400 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d "$OUT/pmc_fetch" -o bench -- $BENCH > "$OUT/pmc_fetch.log" 2>&1
echo "pmc_fetch rc=$?" >> "$OUT/pmc_fetch.log"
timeout 400 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d "$OUT/pmc_write" -o bench -- $BENCH > "$OUT/pmc_write.log" 2>&1
echo "pmc_write rc=$?" >> "$OUT/pmc_write.log"
timeout 400 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES --kernel-trace --output-format csv -d "$OUT/pmc_mfma" -o bench -- $BENCH > "$OUT/pmc_mfma.log" 2>&1
echo "pmc_mfma rc=$?" >> "$OUT/pmc_mfma.log"
cd - > /dev/null
python tools/summarize_prof.py "$OUT" "$TAG" "python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-secondary --no-bf16 --no-decoder-stream $*" > "$OUT/summary.md" 2>&1
cp "profiles/${TAG}_traffic.json" "$OUT/" 2>/dev/null     # (written next to the other profiles: only gpurun_out/ travels back)
# keep the merged-back payload small: per-dispatch traces can be tens of MB
find "$OUT" -name '*kernel_trace.csv' -size +8M -delete
find "$OUT" -name '*counter_collection.csv' -size +8M -delete
tail -40 "$OUT/summary.md"
