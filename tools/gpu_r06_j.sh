#!/bin/bash
# round 6, visit J: extended validation of the final tree -- fuzz with more seeds, the A/B switch, a wider soak, mid-size batches
cd "$(dirname "$0")/.."
OUT=gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
(for s in 41 42 43; do timeout 500 python tools/fuzz_fused.py 2000 $s; done; timeout 700 python tools/fuzz_fused.py 3000 44 int8,bf16,fp8,certified,int8_rn; MSAE_NO_SUBTRACT=1 timeout 500 python tools/fuzz_fused.py 1500 45) 2>&1 | grep -i "cases" | tee $OUT/r06_fuzz_final.txt
MSAE_NO_SUBTRACT=1 timeout 900 python -m pytest tests/test_gpu_hostile.py tests/test_gpu_parity.py -m gpu -q -x -p no:cacheprovider 2>&1 | tail -2
timeout 900 python tools/soak_fused.py --tokens 1048576 --N 262144 --d 4096 --out $OUT/r06_soak_1M_trained_like_n262144.json > $OUT/soak_w.log 2>&1; echo "soak N=262144 exit $?"; tail -1 $OUT/soak_w.log | cut -c1-300
timeout 900 python tools/soak_fused.py --tokens 1048576 --batch 2880 --N 131072 --d 4096 --out $OUT/r06_soak_1M_batch2880_c2.json > $OUT/soak_2880.log 2>&1; echo "soak batch 2880 exit $?"; tail -1 $OUT/soak_2880.log | cut -c1-300
timeout 600 python tools/fuzz_ops.py 2>&1 | tail -1
