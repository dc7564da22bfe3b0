#!/bin/bash
# round 6, first contact: the driver's bench command on the new tree (two-phase loop, gathered cif_conv1d), the new GPU tests, the
# 512 x 30 s CIF dump with token ids, then the whole GPU suite. Bench FIRST (a run right after the suite sees hot HBM).
cd "$(dirname "$0")/.." || exit 1
O=gpurun_out/r06a
mkdir -p $O
export TMPDIR=/tmp
( timeout 400 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err; echo "rc=$?" >> $O/bench.err )
tail -n 5 $O/bench.err; cut -c1-400 $O/bench.json
( timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 --no-interleave --no-cpu-baseline --no-secondary --no-bf16 > $O/bench_sequential.json 2> $O/bench_sequential.err; echo "rc=$?" >> $O/bench_sequential.err )
cut -c1-200 $O/bench_sequential.json
( timeout 600 python -m pytest tests/test_kernels_gpu.py tests/test_parity_gpu.py tests/test_abi.py -x -q -m gpu -k "conv1d or two_phase or one_call or abi or predictor or pipeline" > $O/new_tests.log 2>&1; echo "rc=$?" >> $O/new_tests.log )
tail -n 5 $O/new_tests.log
( timeout 600 python tools/cif_margin_stats.py --clips 512 --seconds 30 --gpu-batch 128 --ids --dump $O/cif_dump_512x30s.pt > $O/cif_dump.log 2>&1; echo "rc=$?" >> $O/cif_dump.log )
tail -n 3 $O/cif_dump.log
( timeout 2400 python -m pytest tests -x -q -m gpu > $O/gputest.log 2>&1; echo "rc=$?" >> $O/gputest.log )
tail -n 8 $O/gputest.log
