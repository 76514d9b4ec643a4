#!/bin/bash
# round 6, visit F: the headline against the length of the warm-up (power-management transient after idle)
cd "$(dirname "$0")/.."
OUT=gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
for w in 5 50 200 5 800 5; do
  timeout 600 python bench.py --steps 20 --warmup $w --no-cpu-baseline --no-secondary > $OUT/r06_bench_w$w.json 2> $OUT/bench.err
  python - <<PY
import json
d=json.loads(open('$OUT/r06_bench_w$w.json').read().strip().splitlines()[-1])
print("warmup %4d: step %.3f ms " % ($w, d['ms_per_step']), {k: round(v,3) for k,v in d['stage_ms'].items()}, d.get('clock',{}).get('effective_sclk_mhz'), d.get('clock',{}).get('avg_power_w'))
PY
  sleep 5
done
