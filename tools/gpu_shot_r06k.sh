#!/bin/bash
# round 6: the decoder on a second stream (one replica): test, then same-call A/B of the three loops inside one bench run
cd "$(dirname "$0")/.." || exit 1
O=gpurun_out/r06k
mkdir -p $O
export TMPDIR=/tmp
( timeout 300 python -m pytest tests/test_parity_gpu.py -x -q -m gpu -k "two_phase or one_call" > $O/tests.log 2>&1; echo "rc=$?" >> $O/tests.log ); tail -n 4 $O/tests.log
( timeout 400 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-bf16 > $O/bench.json 2> $O/bench.err; echo "rc=$?" >> $O/bench.err )
python - <<PY
import json
d = json.load(open("$O/bench.json"))
print("value", d["value"], "ms", d["ms_per_step"], "interleave", d.get("interleave"), "two_replicas", d.get("two_replicas", {}).get("value"), "pcie", d.get("value_pcie_inclusive"), "sv", d.get("sensevoice", {}).get("value"))
PY
