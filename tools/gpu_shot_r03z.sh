#!/bin/bash
# the driver's own bench command (N = 1) at the round's final state, stdout JSON line + stderr stage trace kept
cd "$(dirname "$0")/.." || exit 1
O=gpurun_out/r03z
mkdir -p $O
export TMPDIR=/tmp
( timeout 200 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err; echo "rc=$?" >> $O/bench.err )
tail -n 30 $O/bench.err; cut -c1-600 $O/bench.json
