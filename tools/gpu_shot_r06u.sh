#!/bin/bash
# round 6, shot u: VAD pipeline with segment batches shared across recordings (rows budget)
set -u
OUT=gpurun_out/r06u; mkdir -p $OUT
timeout 900 python -m pytest tests/test_auto_model.py tests/test_reference_vad_pipeline_differential.py tests/test_vad_gpu.py -x -q -m gpu > $OUT/pytest.txt 2>&1; echo "rc=$?" >> $OUT/pytest.txt; tail -n 12 $OUT/pytest.txt
timeout 900 python tools/bench_longform.py --recordings 200 --minutes 1 --batch-size-rows 32768 > $OUT/calls_200x1min_rows.json 2> $OUT/err1.txt; tail -n 3 $OUT/err1.txt; cat $OUT/calls_200x1min_rows.json
timeout 900 python tools/bench_longform.py --recordings 200 --minutes 1 > $OUT/calls_200x1min_300s.json 2> $OUT/err2.txt; tail -n 3 $OUT/err2.txt; cat $OUT/calls_200x1min_300s.json
timeout 900 python tools/bench_longform.py --recordings 4 --minutes 20 --batch-size-rows 32768 > $OUT/longform_rows.json 2> $OUT/err3.txt; tail -n 3 $OUT/err3.txt; cat $OUT/longform_rows.json
