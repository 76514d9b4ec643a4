#!/bin/bash
# wave-block LDS-DMA staging (gemm_stage_block): parity subset first, then the bench line
cd "$(dirname "$0")/.."
OUT=gpurun_out/r05l
mkdir -p $OUT
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_hostile.py -m gpu -q -x --tb=short -p no:cacheprovider > $OUT/pytest_subset.log 2>&1; tail -4 $OUT/pytest_subset.log | cut -c1-200
for i in 1 2; do
timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-secondary > $OUT/bench$i.json 2> $OUT/bench.err; echo "bench exit $?"
python - <<PY
import json
r=json.loads(open('gpurun_out/r05l/bench$i.json').read().strip().splitlines()[-1])
print(round(r['ms_per_step'],3), {a: round(b,3) for a,b in r['stage_ms'].items()}, r['roofline']['frac'], r['roofline'].get('kernel_ms'))
PY
done
