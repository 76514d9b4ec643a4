#!/bin/bash
cd "$(dirname "$0")/.."
OUT=gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
for T in 256 257; do
rm -rf $OUT/prof_t$T
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OLDPWD/$OUT/prof_t$T -o t -- python $OLDPWD/tools/tN_trace.py $T > /dev/null 2>&1)
for f in $(find $OUT/prof_t$T -name "*kernel_stats.csv" | head -1); do echo "== T=$T"; python - <<PY
import csv
for r in list(csv.DictReader(open("$f")))[:14]:
    n=r['Name']
    if 'at::native' in n or 'rocclr' in n: continue
    print(f"{float(r['AverageNs'])/1e3:9.1f} us x{r['Calls']:>3}  {n[:120]}")
PY
done
done
true
