#!/bin/bash
cd "$(dirname "$0")/.."
OUT=gpurun_out/r05m
mkdir -p $OUT
SECONDS=0; python bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err; echo "exit $?"
echo "wall seconds: $SECONDS"
python - <<PY
import json
r=json.loads(open('gpurun_out/r05m/bench_default.json').read().strip().splitlines()[-1])
print(r['steps'], r['warmup'], round(r['ms_per_step'],3), r['roofline']['counter_pass'].get('stale'), len(open('gpurun_out/r05m/bench_default.json').read().strip().splitlines()))
PY
