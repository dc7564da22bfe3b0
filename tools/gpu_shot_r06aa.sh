#!/bin/bash
# round 6, shot aa: linear_out's epilogue operands prefetched during the K loop (PF_ROW_PREFETCH), A/B in one call
set -u
OUT=gpurun_out/r06aa; mkdir -p $OUT
PF_ROW_PREFETCH=1 timeout 600 python -m pytest tests/test_kernels_f16x2_gpu.py tests/test_parity_gpu.py -x -q -m gpu -k "row or fsmn or encoder or full" > $OUT/pytest.txt 2>&1; tail -n 3 $OUT/pytest.txt
for rep in 1; do for v in 0 2 3 0; do
  PF_ROW_PREFETCH=$v timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-secondary --no-bf16 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
site=[s for s in d['kernels']['by_call_site'] if 'linear_out' in s['site']][0]
print('prefetch=$v', d['value'], d['ms_per_step'], 'linear_out us', site['us_per_launch'], 'sclk', d['sclk_mhz_mean'])"
done; done | tee $OUT/ab.txt
