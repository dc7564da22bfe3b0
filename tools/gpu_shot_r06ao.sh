#!/bin/bash
# round 6, shot ao: A/B of the decoder stream's HIP priority in the bench loop (same box, alternating)
OUT=gpurun_out/r06ao; mkdir -p $OUT
for rep in 1 2 3; do for pr in none -1; do
  if [ $pr = none ]; then unset PF_MAIN_STREAM_PRIORITY; else export PF_MAIN_STREAM_PRIORITY=$pr; fi
  timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-secondary --no-bf16 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('main-stream priority', '$pr', d['value'], d['ms_per_step'], d['sclk_mhz_mean'], d['power_w_mean'])" | tee -a $OUT/ab_main.txt
done; done
