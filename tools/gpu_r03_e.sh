#!/bin/bash
cd "$(dirname "$0")/.."
OUT=gpurun_out; R=r03
export TMPDIR=/tmp
echo "== parity, 64-byte ring =="
MSAE_GEMM_RING64=1 timeout 900 python -m pytest tests/test_gpu_hostile.py tests/test_gpu_parity.py -m gpu -q --tb=short -p no:cacheprovider -x > $OUT/${R}_pytest_ring64.log 2>&1
echo "pytest exit $?"; grep -E "passed|failed|error" $OUT/${R}_pytest_ring64.log | tail -3; grep -E "^FAILED|^ERROR" $OUT/${R}_pytest_ring64.log | head; tail -30 $OUT/${R}_pytest_ring64.log | cut -c1-200
echo "== A/B =="
bash tools/gpu_ab2.sh -r 2 "tile128||" "row128||MSAE_GEMM_ROWMAJOR=1" "tile128_stag1|tools/bin/libmsae_stag1.so|" "row128_stag1|tools/bin/libmsae_stag1.so|MSAE_GEMM_ROWMAJOR=1" "ring64||MSAE_GEMM_RING64=1" 2>&1 | tee $OUT/${R}_ab_ring64.txt
