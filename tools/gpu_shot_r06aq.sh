#!/bin/bash
# round 6, shot aq: the bf16-operand mode (BASELINE configs[1] names bf16; the line's `bf16_mode` field) under rocprofv3, kernels by launch shape
set -u
R=$PWD; OUT=$R/gpurun_out/r06aq; mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp && timeout 500 rocprofv3 --kernel-trace --output-format csv -d $OUT/trace -o bench -- python $R/bench.py --precision bf16 --steps 3 --warmup 1 --no-cpu-baseline --no-secondary --no-bf16 --no-decoder-stream > $OUT/bench.log 2>&1
cd $R; F=$(find $OUT/trace -name "*kernel_trace.csv" | head -1); python tools/trace_by_grid.py $F > $OUT/by_grid.txt 2>&1
find $OUT -name "*.csv" -size +8M -delete
tail -2 $OUT/bench.log | cut -c1-400; head -30 $OUT/by_grid.txt
