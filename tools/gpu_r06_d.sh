#!/bin/bash
# round 6, visit D: the guarantee's evidence under the subtractive dither -- z sweep, soaks, fuzz
cd "$(dirname "$0")/.."
OUT=gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
R=r06
timeout 900 python tools/z_sweep.py --out $OUT/${R}_z_sweep_n32768.json > $OUT/${R}_z_sweep.txt 2>&1; echo "z sweep exit $?"; grep "'z'" $OUT/${R}_z_sweep.txt | cut -c1-200
timeout 900 python tools/z_sweep.py --tokens 131072 --N 131072 --d 4096 --out $OUT/${R}_z_sweep_c2.json >> $OUT/${R}_z_sweep.txt 2>&1; echo "z sweep c2 exit $?"; grep "'z'" $OUT/${R}_z_sweep.txt | tail -6 | cut -c1-200
timeout 1500 python tools/soak_fused.py --tokens 16777216 --N 131072 --d 4096 --out $OUT/${R}_soak_16M_trained_like_c2.json > $OUT/soak.log 2>&1; echo "soak 16M exit $?"; tail -1 $OUT/soak.log | cut -c1-500
timeout 900 python tools/soak_fused.py --tokens 1048576 --N 32768 --d 1024 --out $OUT/${R}_soak_1M_trained_like_n32768.json >> $OUT/soak.log 2>&1; echo "soak n32768 exit $?"
timeout 900 python tools/soak_fused.py --tokens 262144 --N 131072 --d 4096 --k 256 --out $OUT/${R}_soak_k256_trained_like_c2.json >> $OUT/soak.log 2>&1; echo "soak k256 exit $?"; tail -1 $OUT/soak.log | cut -c1-400
(MSAE_FM=1 timeout 400 python tools/fuzz_fused.py 1500 11; MSAE_FM=0 timeout 400 python tools/fuzz_fused.py 1500 11; timeout 400 python tools/fuzz_fused.py 1500 12; timeout 600 python tools/fuzz_fused.py 2500 21 int8,bf16,fp8,certified,int8_rn) 2>&1 | grep -i "cases" > $OUT/${R}_fuzz.txt; cat $OUT/${R}_fuzz.txt
timeout 600 python tools/fuzz_ops.py 2>&1 | tail -2 > $OUT/${R}_fuzz_ops.txt; cat $OUT/${R}_fuzz_ops.txt
