#!/bin/bash
cd "$(dirname "$0")/.."
OUT=gpurun_out/r05m
mkdir -p $OUT
timeout 600 python tools/z_sweep.py --out $OUT/z_sweep_n32768.json > $OUT/z_sweep.txt 2>&1; echo "exit $?"
timeout 600 python tools/z_sweep.py --dither off --out $OUT/z_sweep_n32768_dither_off.json >> $OUT/z_sweep.txt 2>&1; echo "exit $?"
timeout 900 python tools/z_sweep.py --N 131072 --d 4096 --tokens 131072 --out $OUT/z_sweep_c2.json >> $OUT/z_sweep.txt 2>&1; echo "exit $?"
grep -v amdgpu.ids $OUT/z_sweep.txt | cut -c1-330
