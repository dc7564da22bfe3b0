#!/bin/bash
# second short call: where the f16x2 streaming step overtakes the fp32 one, and what its floor at S = 64 is made of
cd "$(dirname "$0")/.." || exit 1
O=gpurun_out/r03w
mkdir -p $O
export TMPDIR=/tmp
( timeout 150 python tools/bench_streaming.py --streams 32 96 128 192 --precision fp32 f16x2 --graph 1 --steps 20 --warmup 4 > $O/bench_streaming.jsonl 2> $O/bench_streaming.err; echo "rc=$?" >> $O/bench_streaming.err )
( cd /tmp && timeout 150 rocprofv3 --kernel-trace --stats -d $OLDPWD/$O/prof -- python $OLDPWD/tools/bench_streaming.py --streams 64 --precision f16x2 --graph 0 --steps 10 --warmup 2 > $OLDPWD/$O/prof_run.log 2>&1; echo "rc=$?" >> $OLDPWD/$O/prof_run.log )
find $O/prof -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/stream64_f16x2_kernel_stats.csv
find $O/prof -type f ! -name "*stats.csv" -size +2M -delete
cat $O/bench_streaming.jsonl | cut -c1-400
head -30 $O/stream64_f16x2_kernel_stats.csv
