#!/bin/bash
cd "$(dirname "$0")/.."
OUT=gpurun_out/r05f
mkdir -p $OUT
export TMPDIR=/tmp
python tools/f32_probe.py 10 2>&1 | tail -1 | tee $OUT/f32_rate.txt
timeout 900 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider -k "pre_acts or exact or oracle or fixture or small_T or fallback or degenerate" > $OUT/pytest_f32.log 2>&1; tail -4 $OUT/pytest_f32.log
