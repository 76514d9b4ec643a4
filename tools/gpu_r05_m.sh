#!/bin/bash
cd "$(dirname "$0")/.."
OUT=gpurun_out/r05m
mkdir -p $OUT
timeout 500 python tools/extreme_probe.py > $OUT/extreme.txt 2>&1; echo "exit $?"; tail -45 $OUT/extreme.txt | cut -c1-250
