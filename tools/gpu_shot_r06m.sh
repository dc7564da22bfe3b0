#!/bin/bash
# round 6, shot m: the 10 000-clip corpus sweep (BASELINE configs[3], its 1-GPU part) with the bench's two-phase loop, beside round 5's loop;
# the two loops' hypotheses compared clip by clip.
set -u
OUT=gpurun_out/r06m; mkdir -p $OUT
for i in 1 2; do
  timeout 600 python tools/sweep.py --clips 10000 --dump $OUT/two_phase.json > $OUT/sweep_two_phase_$i.json 2> $OUT/sweep_two_phase_$i.err
  timeout 600 python tools/sweep.py --clips 10000 --no-decoder-stream --dump $OUT/one_stream.json > $OUT/sweep_one_stream_$i.json 2> $OUT/sweep_one_stream_$i.err
done
python - <<'PY'
import json
a = json.load(open("gpurun_out/r06m/two_phase.json")); b = json.load(open("gpurun_out/r06m/one_stream.json"))
print(json.dumps({"clips": len(a), "clips_with_different_ids": sum(x != y for x, y in zip(a, b))}))
PY
tail -n 2 $OUT/sweep_*.json
