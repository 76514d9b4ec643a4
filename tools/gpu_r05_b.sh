#!/bin/bash
# Round 5, GPU visit B: new bench line (secondary records), torchrun 1-rank forced collectives, bench-related tests
cd "$(dirname "$0")/.."
OUT=gpurun_out/r05b
mkdir -p $OUT
export TMPDIR=/tmp
timeout 600 python bench.py --steps 20 --warmup 5 > $OUT/bench.json 2> $OUT/bench.err; echo "bench exit $?"
python - <<'PY'
import json
r=json.loads(open('gpurun_out/r05b/bench.json').read().strip().splitlines()[-1])
print(round(r['ms_per_step'],3), {a: round(b,3) for a,b in r['stage_ms'].items()}, r.get('rows_rescored_per_token'))
for k in ('k256','zipf','exact_modes','dither_off','cpu_baseline'):
    print(k, json.dumps(r.get(k))[:600])
print('roofline', json.dumps(r['roofline'])[:900])
PY
tail -5 $OUT/bench.err
timeout 900 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider -k "bench or torchrun or launch" > $OUT/pytest_bench.log 2>&1; tail -5 $OUT/pytest_bench.log
