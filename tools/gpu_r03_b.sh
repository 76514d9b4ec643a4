#!/bin/bash
# Round 3, visit b: whole suite again; the integer-prefilter epilogue (-DMSAE_EPI_INT) -- parity on every weight family,
# then A/B against the default build; re-score rows at k = 256.
cd "$(dirname "$0")/.."
OUT=gpurun_out; R=r03
mkdir -p $OUT
export TMPDIR=/tmp
echo "== pytest -m gpu =="
timeout 1500 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider --durations=10 > $OUT/${R}_pytest_gpu.log 2>&1
echo "pytest exit $?" | tee -a $OUT/${R}_pytest_gpu.log
grep -E "passed|failed|error" $OUT/${R}_pytest_gpu.log | tail -5
grep -E "^FAILED|^ERROR" $OUT/${R}_pytest_gpu.log | head -20
echo "== EPI_INT parity =="
MSAE_HIP_LIB=tools/bin/libmsae_epi.so timeout 900 python -m pytest tests/test_gpu_hostile.py tests/test_gpu_parity.py -m gpu -q --tb=short -p no:cacheprovider -k "fused or hostile or heterogeneous or small_T or degenerate or sharded or width" > $OUT/${R}_pytest_epi.log 2>&1
echo "epi pytest exit $?"; grep -E "passed|failed|error" $OUT/${R}_pytest_epi.log | tail -3; grep -E "^FAILED|^ERROR" $OUT/${R}_pytest_epi.log | head
echo "== A/B =="
bash tools/gpu_ab.sh tools/bin/libmsae_base.so tools/bin/libmsae_epi.so 2>&1 | tee $OUT/${R}_ab_epilogue_int.txt
echo "== rescore rows k=256 =="
K=256 MSAE_HIP_LIB=tools/bin/libmsae_dbg.so timeout 300 python tools/rescore_stats.py bench 2>&1 | grep -v amdgpu.ids | tee $OUT/${R}_rescore_stats_k256.txt
