#!/bin/bash
# round 6: the pruned product library -- the whole GPU suite with durations; then the f16x2 kernel tests once more on the measurement
# library (the measured-and-off shapes must still return the product's bits)
cd "$(dirname "$0")/.." || exit 1
O=gpurun_out/r06j
mkdir -p $O
export TMPDIR=/tmp
( timeout 1500 python -m pytest tests -x -q -m gpu --durations=25 > $O/gputest.log 2>&1; echo "rc=$?" >> $O/gputest.log )
tail -n 40 $O/gputest.log
( PF_LIB_PATH=$PWD/funasr_amd/libparaformer_hip_measure.so timeout 900 python -m pytest tests/test_kernels_f16x2_gpu.py -x -q -m gpu --durations=8 > $O/gputest_measure.log 2>&1; echo "rc=$?" >> $O/gputest_measure.log )
tail -n 15 $O/gputest_measure.log
