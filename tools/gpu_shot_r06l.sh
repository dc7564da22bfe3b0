#!/bin/bash
# round 6: rocprofv3 kernel stats + PMC passes of the main mode on the final tree, then the driver's bench command, then the GPU suite
cd "$(dirname "$0")/.." || exit 1
O=gpurun_out/r06m
mkdir -p $O
export TMPDIR=/tmp
bash tools/profile.sh r06 > $O/profile.log 2>&1
tail -n 5 $O/profile.log
cp profiles/r06_* $O/ 2>/dev/null
( timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err; echo "rc=$?" >> $O/bench.err )
tail -n 4 $O/bench.err; cut -c1-300 $O/bench.json



