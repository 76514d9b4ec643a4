#!/bin/bash
cd "$(dirname "$0")/.."
OUT=gpurun_out/r05k
mkdir -p $OUT
export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider > $OUT/pytest_gpu.log 2>&1; tail -6 $OUT/pytest_gpu.log | cut -c1-200
timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $OUT/bench.json 2> $OUT/bench.err; echo "bench exit $?"
python - <<'PY'
import json
r=json.loads(open('gpurun_out/r05k/bench.json').read().strip().splitlines()[-1])
print(round(r['ms_per_step'],3), {a: round(b,3) for a,b in r['stage_ms'].items()}, r.get('rows_rescored_per_token'))
print('coarse_fp8', json.dumps(r.get('coarse_fp8'))[:700])
PY
MSAE_COARSE=fp8 timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-secondary 2>/dev/null | python -c "
import json,sys
r=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('fp8 headline', round(r['ms_per_step'],3), {a: round(b,3) for a,b in r['stage_ms'].items()}, r.get('rows_rescored_per_token'), r['roofline']['frac'], r['fast_path_verified_frac'])"
timeout 600 python tools/soak_fused.py --tokens 262144 --N 131072 --d 4096 --coarse fp8 --out $OUT/soak_fp8_c2.json > $OUT/soak.log 2>&1; tail -1 $OUT/soak.log | cut -c1-330
