#!/bin/bash
# third short call: per-kernel time of the S = 64 streaming step in both arithmetic modes (eager, so every launch is visible)
cd "$(dirname "$0")/.." || exit 1
R=$PWD
O=$R/gpurun_out/r03x
mkdir -p $O
export TMPDIR=/tmp
cd /tmp
for prec in f16x2 fp32; do
  timeout 120 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_$prec -o s64 -- python $R/tools/bench_streaming.py --streams 64 --precision $prec --graph 0 --steps 10 --warmup 2 > $O/prof_$prec.log 2>&1
  echo "rc=$?" >> $O/prof_$prec.log
  find $O/prof_$prec -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/stream64_${prec}_kernel_stats.csv
  find $O/prof_$prec -name '*kernel_trace.csv' -size +8M -delete
done
head -25 $O/stream64_f16x2_kernel_stats.csv | cut -c1-200
