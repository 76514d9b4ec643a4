#!/bin/bash
# stage clocks of bench.py at a small batch for several library builds.  usage: sk_ab.sh tokens lib1 lib2 ...  ("-" = in-tree)
cd "$(dirname "$0")/.."
t=$1; shift
for l in "$@"; do
  ( [ "$l" != "-" ] && export MSAE_HIP_LIB=$l; python bench.py --tokens $t --steps 50 --warmup 10 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
r=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('T=$t %-32s' % '$l', round(r['ms_per_step'],3), {k: round(v,4) for k,v in r['stage_ms'].items()})" )
done
