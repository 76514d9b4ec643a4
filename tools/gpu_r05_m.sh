#!/bin/bash
cd "$(dirname "$0")/.."
OUT=gpurun_out/r05m
mkdir -p $OUT
timeout 500 python -m pytest tests/test_gpu_hostile.py -m gpu -q -x --tb=short -p no:cacheprovider -k "non_finite" > $OUT/pytest_nonfinite.log 2>&1; tail -5 $OUT/pytest_nonfinite.log | cut -c1-250
