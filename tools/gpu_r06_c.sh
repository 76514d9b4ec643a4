#!/bin/bash
# round 6, visit C: GPU suite (new tests), bench line with the new records, small-T latencies
cd "$(dirname "$0")/.."
OUT=gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider -x > $OUT/pytest_gpu.log 2>&1
echo "pytest exit $?"; tail -8 $OUT/pytest_gpu.log | cut -c1-300
timeout 900 python bench.py --steps 20 --warmup 5 > $OUT/r06_bench_c.json 2> $OUT/bench.err; echo "bench exit $?"; cut -c1-200 $OUT/r06_bench_c.json; tail -3 $OUT/bench.err
timeout 300 python tools/latency_small_T.py > $OUT/r06_latency_small_T.txt 2>&1; grep "T=" $OUT/r06_latency_small_T.txt
timeout 600 python tools/sanity_shapes.py > $OUT/r06_other_shapes.txt 2>&1; grep "T=" $OUT/r06_other_shapes.txt
