#!/bin/bash
# Round 3, visit c: tile-major operands in the candidate GEMM (default) vs row-major (MSAE_GEMM_ROWMAJOR=1), same library.
cd "$(dirname "$0")/.."
OUT=gpurun_out; R=r03
mkdir -p $OUT/ab
export TMPDIR=/tmp
echo "== parity of the tile-major build =="
timeout 900 python -m pytest tests/test_gpu_hostile.py tests/test_gpu_parity.py -m gpu -q --tb=short -p no:cacheprovider > $OUT/${R}_pytest_tm.log 2>&1
echo "pytest exit $?"; grep -E "passed|failed|error" $OUT/${R}_pytest_tm.log | tail -3; grep -E "^FAILED|^ERROR" $OUT/${R}_pytest_tm.log | head
echo "== A/B =="
for rep in 1 2 3; do
  for mode in tile row; do
    if [ $mode = row ]; then export MSAE_GEMM_ROWMAJOR=1; else unset MSAE_GEMM_ROWMAJOR; fi
    timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $OUT/ab/tm_${mode}_$rep.json 2>/dev/null
    python - <<PY
import json
r=json.load(open("$OUT/ab/tm_${mode}_$rep.json"))
print("$mode-major rep $rep: step %.3f ms  " % r["ms_per_step"], {k: round(v,3) for k,v in r["stage_ms"].items()}, "verified", r["fast_path_verified_frac"])
PY
  done
done 2>&1 | tee $OUT/${R}_ab_tile_major.txt
unset MSAE_GEMM_ROWMAJOR
