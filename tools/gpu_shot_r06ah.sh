#!/bin/bash
# round 6, shot ah: the whole GPU suite on the product library after the constructor corners, smoke, the driver's bench command
set -u
OUT=gpurun_out/${SHOT:-r06ah}; mkdir -p $OUT
timeout 1500 python -m pytest tests -q -m gpu > $OUT/gputest.txt 2>&1; echo "rc=$?" >> $OUT/gputest.txt; tail -n 6 $OUT/gputest.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $OUT/smoke.txt 2>&1; tail -n 2 $OUT/smoke.txt
export SHOT=${SHOT:-r06ah}
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc=$?"; python - <<'PY'
import json
import os
d=json.loads(open('gpurun_out/' + os.environ.get('SHOT', 'r06ah') + '/bench.json').read().strip().splitlines()[-1])
print({k:d[k] for k in ('metric','value','ms_per_step','n_gpus','dtype')}, d['roofline']['frac'], d['cpu_baseline']['value'], d['cpu_baseline']['kind'])
PY
