#!/bin/bash
cd "$(dirname "$0")/.."
OUT=gpurun_out/r05m
mkdir -p $OUT
timeout 1200 python tools/fuzz_ops.py 600 2 > $OUT/fuzz_ops.txt 2>&1; echo "exit $?"; grep -v amdgpu.ids $OUT/fuzz_ops.txt | tail -25 | cut -c1-300
