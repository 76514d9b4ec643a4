#!/bin/bash
# S = 1 path: parity tests, latency table, kernel timeline of one encode + decode step
cd "$(dirname "$0")/.."
OUT=gpurun_out/small; mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_hostile.py tests/test_gpu_parity.py -m gpu -x -q -k "small_T or entry_points or fused_encode_bit_exact or hook_edits or degenerate or soak_small or dropin or steering" 2>&1 | tail -15 > $OUT/pytest.log
cat $OUT/pytest.log
timeout 300 python tools/latency_small_T.py 2>&1 | grep "T=" | tee $OUT/latency.txt
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OLDPWD/$OUT/prof -o t1 -- python $OLDPWD/tools/t1_trace.py > $OLDPWD/$OUT/t1.log 2>&1
cd $OLDPWD
tail -2 $OUT/t1.log
f=$(find $OUT/prof -name "*kernel_stats.csv" | head -1)
python - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows[:16]:
    print("%-60s calls %5s avg %9.1f ns" % (r["Name"][:60], r["Calls"], float(r["AverageNs"])))
PY
cp "$f" $OUT/t1_kernel_stats.csv
