#!/bin/bash
# flakiness check: the GPU suite N times on one box (the dither's seeds are drawn per call, several tests assert statistics)
cd "$(dirname "$0")/.."
OUT=gpurun_out/loop
mkdir -p $OUT
N=${1:-5}
for i in $(seq 1 $N); do
  timeout 900 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider > $OUT/run$i.log 2>&1
  echo "run $i: exit $? $(grep -E 'passed|failed' $OUT/run$i.log | tail -1)"
done
