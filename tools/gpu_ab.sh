#!/bin/bash
# A/B of library variants on ONE box: bench.py stage times, alternating.  usage: gpu_ab.sh libA.so libB.so ...
cd "$(dirname "$0")/.."
OUT=gpurun_out/ab; mkdir -p $OUT
for rep in 1 2; do
  for lib in "$@"; do
    n=$(basename $lib .so)
    MSAE_HIP_LIB=$lib timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-secondary > $OUT/${n}_$rep.json 2>/dev/null
    python - <<PY
import json
r=json.load(open("$OUT/${n}_$rep.json"))
print("$n rep $rep: step %.3f ms  " % r["ms_per_step"], {k: round(v,3) for k,v in r["stage_ms"].items()}, "verified", r["fast_path_verified_frac"])
PY
  done
done
