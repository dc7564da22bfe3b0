#!/bin/bash
# Run on the GPU box (through gpurun): rocprofv3 kernel trace of the streaming step (tools/bench_streaming.py), condensed to
# per-kernel counts / durations per step and the mean gap between consecutive kernels of a replayed step.
# usage: tools/profile_stream.sh TAG STREAMS [PRECISION]
set -u
TAG=${1:-r04}; S=${2:-1}; PREC=${3:-fp32}
OUT=$PWD/gpurun_out/prof_stream_$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
CMD="python $PWD/tools/bench_streaming.py --streams $S --graph 1 --precision $PREC --steps 40 --warmup 5"
cd /tmp
timeout 400 rocprofv3 --kernel-trace --output-format csv -d "$OUT/trace" -o s -- $CMD > "$OUT/trace.log" 2>&1
echo "rc=$?" >> "$OUT/trace.log"
cd - > /dev/null
python tools/analyze_stream_trace.py "$(find "$OUT/trace" -name '*kernel_trace.csv' | head -1)" --idle-us 100 | tee "$OUT/summary.json"
find "$OUT" -name '*kernel_trace.csv' -size +8M -delete
