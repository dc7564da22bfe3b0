#!/bin/bash
# round 6: the persistent f16x2 GEMM with the paired three-instruction loaders: parity, us per launch with the measurement switches
cd "$(dirname "$0")/.." || exit 1
O=gpurun_out/r06h
mkdir -p $O
export TMPDIR=/tmp
( timeout 150 python tools/bench_ps.py parity > $O/ps_parity.jsonl 2> $O/ps_parity.err; echo "rc=$?" >> $O/ps_parity.err )
tail -n 2 $O/ps_parity.jsonl; tail -n 2 $O/ps_parity.err
( timeout 200 python tools/bench_ps.py time > $O/ps_time.jsonl 2> $O/ps_time.err; echo "rc=$?" >> $O/ps_time.err )
cat $O/ps_time.jsonl; tail -n 2 $O/ps_time.err
