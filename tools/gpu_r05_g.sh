#!/bin/bash
cd "$(dirname "$0")/.."
OUT=gpurun_out/r05g
mkdir -p $OUT
for w in 0 5; do echo "== wave $w (inside k-tile 8)"; MSAE_HIP_LIB=tools/bin/libmsae_tlk$w.so python tools/gemm_timeline.py --k 2>&1 | tail -7; done | tee $OUT/ktile_timeline.txt
echo "== per output tile"; MSAE_HIP_LIB=tools/bin/libmsae_tl.so python tools/gemm_timeline.py 2>&1 | tail -10 | tee $OUT/tile_timeline.txt
