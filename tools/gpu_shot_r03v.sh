#!/bin/bash
# One short GPU call (round 3 had ~6 GPU-minutes left): first hardware run of the streaming step's f16x2 form, most
# informative pieces first, every piece under its own timeout, logs written as they go (gpurun_out/ is merged back even
# when the call is cut off).
cd "$(dirname "$0")/.." || exit 1
O=gpurun_out/r03v
mkdir -p $O
export TMPDIR=/tmp
( timeout 170 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "rc=$?" >> $O/smoke.log )
for c in golden_graph many_streams_vs_fp32_step batch_independence_and_graph geometries weight_reload golden_eager oracle_geometry_many_tokens; do
  ( timeout 100 python tests/_stream_f16x2_cases.py $c > $O/case_$c.log 2>&1; echo "rc=$?" >> $O/case_$c.log )
done
( timeout 120 python -m pytest tests/test_streaming_gpu.py -m gpu -x -q > $O/pytest_streaming.log 2>&1; echo "rc=$?" >> $O/pytest_streaming.log )
( timeout 200 python tools/bench_streaming.py --streams 64 256 8 --precision fp32 f16x2 --graph 1 --steps 20 --warmup 4 > $O/bench_streaming.jsonl 2> $O/bench_streaming.err; echo "rc=$?" >> $O/bench_streaming.err )
( timeout 200 python -m pytest tests/test_parity_gpu.py -m gpu -x -q > $O/pytest_parity.log 2>&1; echo "rc=$?" >> $O/pytest_parity.log )
tail -n 3 $O/*.log $O/bench_streaming.jsonl
