#!/bin/bash
# Run on the GPU box: the two-ranks-on-one-GPU sweep WITHOUT serialising the GPU section (the configuration in which the frontend was
# seen to return a sporadically disturbed frame), N times with the fbank cross-check on and N times with it off; every dump is compared
# with the one-rank dump. usage: tools/two_rank_frontend_check.sh TAG N
set -u
TAG=${1:-r04}; N=${2:-3}
OUT=$PWD/gpurun_out/two_rank_$TAG
mkdir -p "$OUT"
export HSA_ENABLE_IPC_MODE_LEGACY=0
COMMON="--clips 48 --batch-seconds 1 --no-overlap"
python tools/sweep.py $COMMON --dump "$OUT/one.json" > "$OUT/one.log" 2>&1
for V in ${VERIFY_MODES:-1 0}; do
  for i in $(seq 1 $N); do
    PF_FRONTEND_VERIFY=$V python -m torch.distributed.run --nnodes=1 --nproc-per-node=2 --master-addr 127.0.0.1 --master-port $((29600 + V * 10 + i)) \
        tools/sweep.py $COMMON --dist-backend gloo --dump "$OUT/two_${V}_$i.json" > "$OUT/two_${V}_$i.log" 2>&1
    same=$(python -c "import json,sys; print(json.load(open('$OUT/one.json')) == json.load(open('$OUT/two_${V}_$i.json')))" 2>&1)
    faults=$(grep -h "disagreements seen" "$OUT/two_${V}_$i.log" | sed 's/.*rank \([0-9]\).*seen: \([0-9]*\)/r\1=\2/' | tr '\n' ' ')
    echo "verify=$V run $i: equals the one-rank dump: $same   cross-check disagreements: $faults" | tee -a "$OUT/summary.txt"
    grep -h "disagreements seen" "$OUT/two_${V}_$i.log" | grep -v "seen: 0" | sed 's/^/    /' | tee -a "$OUT/summary.txt"
  done
done
