#!/bin/bash
# last short call of round 3: the split-K form of the streaming step's w_2 and the 128 x 128 plane-output shape
cd "$(dirname "$0")/.." || exit 1
O=gpurun_out/r03s
mkdir -p $O
export TMPDIR=/tmp
( timeout 40 python -m pytest tests/test_kernels_f16x2_gpu.py -m gpu -x -q -k "split_k or (plane_output and not 32768)" > $O/pytest_kernels.log 2>&1; echo "rc=$?" >> $O/pytest_kernels.log )
( timeout 40 python -m pytest tests/test_streaming_f16x2_gpu.py tests/test_streaming_gpu.py -m gpu -x -q > $O/pytest_streaming.log 2>&1; echo "rc=$?" >> $O/pytest_streaming.log )
( timeout 60 python tools/bench_streaming.py --streams 64 256 128 8 32 --precision f16x2 --graph 1 --steps 20 --warmup 4 > $O/bench_streaming.jsonl 2> $O/bench_streaming.err; echo "rc=$?" >> $O/bench_streaming.err )
( timeout 40 python -m pytest tests/test_parity_gpu.py -m gpu -x -q > $O/pytest_parity.log 2>&1; echo "rc=$?" >> $O/pytest_parity.log )
tail -n 4 $O/*.log; cut -c1-330 $O/bench_streaming.jsonl
