#!/bin/bash
# round 6, visit A: GPU suite + bench line (default, dither variants) + re-score statistics
cd "$(dirname "$0")/.."
OUT=gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider -x > $OUT/pytest_gpu.log 2>&1
echo "pytest exit $?"; tail -15 $OUT/pytest_gpu.log
timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $OUT/r06_bench_a.json 2> $OUT/bench.err; echo "bench exit $?"; cut -c1-3000 $OUT/r06_bench_a.json
MSAE_NO_SUBTRACT=1 timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-secondary > $OUT/r06_bench_a_nosub.json 2>> $OUT/bench.err; echo "bench nosub exit $?"; cut -c1-2500 $OUT/r06_bench_a_nosub.json
