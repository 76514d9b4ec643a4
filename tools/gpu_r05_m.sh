#!/bin/bash
cd "$(dirname "$0")/.."
OUT=gpurun_out
timeout 1500 python tools/soak_fused.py --tokens 16777216 --N 131072 --d 4096 --out $OUT/r05_soak_16M_trained_like_c2.json > $OUT/soak16.log 2>&1; echo "soak exit $?"; tail -1 $OUT/soak16.log | cut -c1-500
