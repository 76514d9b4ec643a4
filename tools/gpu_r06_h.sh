#!/bin/bash
# round 6, visit H: subtractive dither in the weight-stream pass (17 ... 256 tokens): band test, suite, small-T latencies, fuzz
cd "$(dirname "$0")/.."
OUT=gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_band.py -m gpu -q --tb=short -p no:cacheprovider -x -s > $OUT/pytest_band.log 2>&1; echo "band exit $?"; grep -E "subtractive|passed|failed|Error|assert" $OUT/pytest_band.log | head
timeout 1500 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider -x > $OUT/pytest_gpu.log 2>&1
echo "pytest exit $?"; tail -4 $OUT/pytest_gpu.log | cut -c1-300
timeout 300 python tools/latency_small_T.py > $OUT/r06_latency_small_T.txt 2>&1; grep "T=" $OUT/r06_latency_small_T.txt
MSAE_NO_SUBTRACT=1 timeout 300 python tools/latency_small_T.py 2>&1 | grep "T=" | sed 's/^/nosub /' | head -14
(timeout 400 python tools/fuzz_fused.py 1500 31; timeout 600 python tools/fuzz_fused.py 2500 32 int8,bf16,fp8,certified,int8_rn) 2>&1 | grep -i "cases"
timeout 600 python tools/soak_fused.py --tokens 262144 --batch 128 --N 32768 --d 1024 --out $OUT/r06_soak_256k_batch128_n32768.json > $OUT/soak_b128.log 2>&1; echo "soak batch 128 exit $?"; tail -1 $OUT/soak_b128.log | cut -c1-400
