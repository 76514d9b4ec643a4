#!/bin/bash
# Round 5, GPU visit A: full parity suite on the dithered default, dither on / off A/B of the bench step and of the re-score rows.
cd "$(dirname "$0")/.."
OUT=gpurun_out/r05a
mkdir -p $OUT
export TMPDIR=/tmp
echo "== pytest -m gpu =="
timeout 1200 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider -s > $OUT/pytest_gpu.log 2>&1
echo "pytest exit $?" | tee -a $OUT/pytest_gpu.log
grep -E "passed|failed|error" $OUT/pytest_gpu.log | tail -5
grep -E "residual-aligned|shape guard|soak 65536" $OUT/pytest_gpu.log
for dm in 1 0; do
  echo "== bench MSAE_DITHER=$dm"
  MSAE_DITHER=$dm timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>$OUT/bench_d$dm.err > $OUT/bench_d$dm.json < /dev/null
  python -c "
import json,sys
r=json.loads(open('$OUT/bench_d$dm.json').read().strip().splitlines()[-1]); print(round(r['ms_per_step'],3), {a: round(b,3) for a,b in r['stage_ms'].items()}, r.get('rows_rescored_per_token'), r.get('rescore_rounds_per_token'), r.get('fast_path_verified_frac'))"
  MSAE_DITHER=$dm timeout 300 python bench.py --k 256 --steps 10 --warmup 3 --no-cpu-baseline 2>>$OUT/bench_d$dm.err > $OUT/bench_k256_d$dm.json < /dev/null
  python -c "
import json,sys
r=json.loads(open('$OUT/bench_k256_d$dm.json').read().strip().splitlines()[-1]); print('k256', round(r['ms_per_step'],3), {a: round(b,3) for a,b in r['stage_ms'].items()}, r.get('rows_rescored_per_token'), r.get('rescore_rounds_per_token'), r.get('fast_path_verified_frac'))"
  MSAE_DITHER=$dm timeout 300 python tools/rescore_stats.py bench trained_like 2>&1 | grep "k=" | tee $OUT/rescore_stats_d$dm.txt
done
