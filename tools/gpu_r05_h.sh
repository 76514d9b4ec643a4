#!/bin/bash
cd "$(dirname "$0")/.."
OUT=gpurun_out/r05h
mkdir -p $OUT
export TMPDIR=/tmp
for rep in 1 2; do
for lib in multimodal-sae_amd/msae/_lib/libmsae_hip.so tools/bin/libmsae_nodefer.so; do
  echo "== $lib"; MSAE_HIP_LIB=$lib timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-secondary 2>/dev/null < /dev/null | python -c "
import json,sys
r=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(r['ms_per_step'],3), {a: round(b,3) for a,b in r['stage_ms'].items()}, r.get('rows_rescored_per_token'), r['clock'].get('effective_sclk_mhz'))"
done; done 2>&1 | tee $OUT/ab_defer_flush.txt
echo "== per output tile"; MSAE_HIP_LIB=tools/bin/libmsae_tl.so python tools/gemm_timeline.py 2>&1 | tail -10 | tee $OUT/tile_timeline.txt
timeout 900 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider -x > $OUT/pytest_gpu.log 2>&1; tail -3 $OUT/pytest_gpu.log
