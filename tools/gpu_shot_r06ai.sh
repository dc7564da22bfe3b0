#!/bin/bash
# round 6, shot ai: SenseVoiceSmall 128 x 10 s under rocprofv3, kernels by (template, grid): where the CTC head's vocabulary GEMM stands
set -u
R=$PWD; OUT=$R/gpurun_out/r06ai; mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp && timeout 500 rocprofv3 --kernel-trace --output-format csv -d $OUT/trace -o sv -- python $R/tools/bench_sensevoice.py --steps 3 --warmup 1 > $OUT/bench.log 2>&1
cd $R; F=$(find $OUT/trace -name "*kernel_trace.csv" | head -1); python tools/trace_by_grid.py $F > $OUT/by_grid.txt 2>&1
find $OUT -name "*.csv" -size +8M -delete
tail -3 $OUT/bench.log; head -24 $OUT/by_grid.txt
