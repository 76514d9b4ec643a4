#!/bin/bash
cd "$(dirname "$0")/.."
OUT=gpurun_out
timeout 600 python bench.py --steps 20 --warmup 5 > $OUT/r05_bench.json 2> $OUT/bench.err; echo "bench exit $?"
python - <<PY
import json
r=json.loads(open('gpurun_out/r05_bench.json').read().strip().splitlines()[-1])
print(round(r['ms_per_step'],3), {a: round(b,3) for a,b in r['stage_ms'].items()}, r['roofline']['frac'], r['roofline']['counter_pass'].get('stale'))
PY
