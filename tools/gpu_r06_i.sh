#!/bin/bash
# round 6, visit I: the bench line after the harness fix (clock inside the sampler context), warm-up sweep again
cd "$(dirname "$0")/.."
OUT=gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
for w in 5 200 5 50 5; do
  timeout 600 python bench.py --steps 20 --warmup $w --no-cpu-baseline --no-secondary > $OUT/r06_bench_w$w.json 2> $OUT/bench.err
  python - <<PY
import json
d=json.loads(open('$OUT/r06_bench_w$w.json').read().strip().splitlines()[-1])
print("warmup %4d: headline pass %.3f ms  instrumented pass (first) %.3f ms  stage sum %.3f " % ($w, d['ms_per_step'], d['ms_per_step_instrumented'], sum(d['stage_ms'].values())), {k: round(v,3) for k,v in d['stage_ms'].items()})
PY
  sleep 3
done
timeout 900 python bench.py --steps 20 --warmup 5 > $OUT/r06_bench.json 2> $OUT/bench.err; echo "bench exit $?"
