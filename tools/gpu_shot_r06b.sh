#!/bin/bash
# round 6: the persistent wave-specialised f16x2 GEMM (tile 10) -- bitwise parity against the shapes in use, us per launch, then the
# step with it switched on for the encoder's tile GEMMs (same call A/B)
cd "$(dirname "$0")/.." || exit 1
O=gpurun_out/r06b
mkdir -p $O
export TMPDIR=/tmp
( timeout 300 python tools/bench_ps.py parity > $O/ps_parity.jsonl 2> $O/ps_parity.err; echo "rc=$?" >> $O/ps_parity.err )
tail -n 4 $O/ps_parity.jsonl; tail -n 3 $O/ps_parity.err
( timeout 400 python tools/bench_ps.py time > $O/ps_time.jsonl 2> $O/ps_time.err; echo "rc=$?" >> $O/ps_time.err )
cat $O/ps_time.jsonl; tail -n 2 $O/ps_time.err
for opt in "" "--enc-option gemm_tile=10" "--enc-option gemm_tile=10 --enc-option w2_tile=10" "--enc-option w2_tile=10"; do
  tag=$(echo "$opt" | tr -c 'a-z0-9_=' '_')
  ( timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-secondary --no-bf16 $opt > $O/bench_$tag.json 2> $O/bench_$tag.err; echo "rc=$?" >> $O/bench_$tag.err )
  echo "== $opt"; python - <<PY
import json
try:
    d = json.load(open("$O/bench_$tag.json"))
    print(d["value"], d["ms_per_step"], d["roofline"]["frac"], d.get("sclk_mhz_mean"), d.get("power_w_mean"), [(k["site"], k["us_per_launch"]) for k in d["kernels"]["by_call_site"][:6]] if "by_call_site" in d["kernels"] else "")
except Exception as e:
    print("failed", e)
PY
done
