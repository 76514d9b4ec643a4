#!/bin/bash
# round 6, visit B: band test first, GPU suite, bench line
cd "$(dirname "$0")/.."
OUT=gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_band.py -m gpu -q --tb=short -p no:cacheprovider -x -s > $OUT/pytest_band.log 2>&1; echo "band exit $?"; tail -25 $OUT/pytest_band.log
timeout 1500 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider -x > $OUT/pytest_gpu.log 2>&1
echo "pytest exit $?"; tail -8 $OUT/pytest_gpu.log
timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $OUT/r06_bench_b.json 2> $OUT/bench.err; echo "bench exit $?"; cut -c1-300 $OUT/r06_bench_b.json
