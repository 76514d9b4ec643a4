#!/bin/bash
# A/B of build / environment variants on ONE box, alternating.  usage: gpu_ab2.sh [-r reps] "name|lib.so|ENV=V ENV2=V" ...
# (lib empty = the in-tree library).  Prints bench.py's step and stage times per variant and repetition.
cd "$(dirname "$0")/.."
REPS=2; if [ "$1" = "-r" ]; then REPS=$2; shift 2; fi
EXTRA=${AB_BENCH_ARGS:-}
OUT=gpurun_out/ab; mkdir -p $OUT
for rep in $(seq 1 $REPS); do
  for spec in "$@"; do
    name=$(echo "$spec" | cut -d'|' -f1); lib=$(echo "$spec" | cut -d'|' -f2); envs=$(echo "$spec" | cut -d'|' -f3)
    ( [ -n "$lib" ] && export MSAE_HIP_LIB=$lib; for e in $envs; do export $e; done
      timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-secondary $EXTRA > $OUT/${name}_$rep.json 2>$OUT/${name}_$rep.err )
    python - <<PY
import json
try:
    r=json.load(open("$OUT/${name}_$rep.json"))
    print("%-22s rep $rep: step %.3f ms  " % ("$name", r["ms_per_step"]), {k: round(v,3) for k,v in r["stage_ms"].items()}, "verified", r["fast_path_verified_frac"])
except Exception as e:
    print("$name rep $rep: FAILED", e, open("$OUT/${name}_$rep.err").read()[-600:])
PY
  done
done
