#!/bin/bash
# round 6, visit G: the committed bench line (counter stamp of THIS tree) + the GPU suite three times in a row (flakiness)
cd "$(dirname "$0")/.."
OUT=gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
timeout 900 python bench.py --steps 20 --warmup 5 > $OUT/r06_bench.json 2> $OUT/bench.err; echo "bench exit $?"; cut -c1-260 $OUT/r06_bench.json
for i in 1 2 3; do timeout 1500 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider > $OUT/pytest_gpu_$i.log 2>&1; echo "suite $i exit $?: $(grep -E 'passed|failed' $OUT/pytest_gpu_$i.log | tail -1)"; done
