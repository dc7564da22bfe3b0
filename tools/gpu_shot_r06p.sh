#!/bin/bash
# round 6, shot p: AutoModel.generate over many batches -- the overlapped loop against the plain one (records equal; throughput over wav files)
set -u
OUT=gpurun_out/r06p; mkdir -p $OUT
timeout 900 python -m pytest tests/test_auto_model.py tests/test_parity_gpu.py -x -q -m gpu > $OUT/pytest.txt 2>&1; echo "pytest rc=$?" >> $OUT/pytest.txt
tail -n 5 $OUT/pytest.txt
timeout 900 python tools/bench_generate.py --clips 2000 --batch-size 64 > $OUT/generate_2000x64.json 2> $OUT/generate.err; tail -n 3 $OUT/generate.err; cat $OUT/generate_2000x64.json
