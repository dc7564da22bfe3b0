#!/bin/bash
# round 6: where does the persistent shape pay inside the engine? Same-call A/B: headline step and SenseVoice with the QKV form on it
cd "$(dirname "$0")/.." || exit 1
O=gpurun_out/r06i
mkdir -p $O
export TMPDIR=/tmp
for mode in 0 1 2; do
  ( PF_QKV_PS=$mode timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-secondary --no-bf16 > $O/bench_qkvps$mode.json 2> $O/bench_qkvps$mode.err; echo "rc=$?" >> $O/bench_qkvps$mode.err )
  python - <<PY
import json
try:
    d = json.load(open("$O/bench_qkvps$mode.json"))
    print("headline PF_QKV_PS=$mode", d["value"], d["ms_per_step"], d["roofline"]["frac"], d.get("sclk_mhz_mean"), d.get("power_w_mean"), [(k["site"], k["us_per_launch"]) for k in d["kernels"]["by_call_site"][:5]])
except Exception as e:
    print("failed", e)
PY
done
for mode in 0 1 2; do
  ( PF_QKV_PS=$mode timeout 300 python tools/bench_sensevoice.py --modes f16x2 --steps 10 --warmup 3 --cpu-clips 0 > $O/sv_qkvps$mode.json 2> $O/sv_qkvps$mode.err; echo "rc=$?" >> $O/sv_qkvps$mode.err )
  echo "sensevoice PF_QKV_PS=$mode"; tail -n 3 $O/sv_qkvps$mode.json | cut -c1-600
done
for opt in "gemm_tile=10" "w2_tile=10"; do
  ( timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-secondary --no-bf16 --enc-option $opt > $O/bench_$opt.json 2> $O/bench_$opt.err; echo "rc=$?" >> $O/bench_$opt.err )
  python - <<PY
import json
try:
    d = json.load(open("$O/bench_$opt.json"))
    print("headline $opt", d["value"], d["ms_per_step"], d["roofline"]["frac"], d.get("sclk_mhz_mean"), d.get("power_w_mean"), [(k["site"], k["us_per_launch"]) for k in d["kernels"]["by_call_site"][:5]])
except Exception as e:
    print("failed", e)
PY
done
