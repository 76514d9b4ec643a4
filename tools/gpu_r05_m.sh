#!/bin/bash
cd "$(dirname "$0")/.."
OUT=gpurun_out/r05m
mkdir -p $OUT
(timeout 900 python tools/fuzz_fused.py 2500 21 int8,bf16,fp8,certified,int8_rn; MSAE_FM=1 timeout 600 python tools/fuzz_fused.py 1200 22 int8,certified,fp8; MSAE_FM=0 timeout 600 python tools/fuzz_fused.py 1200 23 int8,certified,fp8) 2>&1 | grep -i "cases\|MISMATCH\|Error\|Traceback" | cut -c1-400 > $OUT/fuzz_modes.txt; cat $OUT/fuzz_modes.txt
