#!/bin/bash
# round 6, visit E: GPU suite, hook latency, bench, shard emulation, small-T latencies
cd "$(dirname "$0")/.."
OUT=gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider -x > $OUT/pytest_gpu.log 2>&1
echo "pytest exit $?"; tail -8 $OUT/pytest_gpu.log | cut -c1-300
timeout 300 python tools/latency_hook_S1.py > $OUT/r06_latency_hook_S1.txt 2>&1; grep "S=1" $OUT/r06_latency_hook_S1.txt
timeout 900 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-secondary > $OUT/r06_bench_e.json 2> $OUT/bench.err; echo "bench exit $?"; python - <<'PY'
import json
d=json.loads(open('gpurun_out/r06_bench_e.json').read().strip().splitlines()[-1])
print(d['ms_per_step'], {k: round(v,3) for k,v in d['stage_ms'].items()}, d['rows_rescored_per_token'])
PY
timeout 600 python tools/emulate_shard.py > $OUT/r06_emulate_shard.txt 2>&1; grep "G=" $OUT/r06_emulate_shard.txt
timeout 300 python tools/latency_small_T.py > $OUT/r06_latency_small_T.txt 2>&1; grep "T=" $OUT/r06_latency_small_T.txt | head -20
