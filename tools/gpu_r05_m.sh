#!/bin/bash
# counters + bench line of the committed tree (the traffic file is stamped with the csrc hash)
cd "$(dirname "$0")/.."
OUT=gpurun_out
R=r05
export TMPDIR=/tmp
rm -rf $OUT/prof
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -o $R -- python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-secondary > $OUT/${R}_bench_under_rocprof.json 2> $OUT/rocprof.err; echo "rocprof exit $?"
find $OUT/prof -name "*kernel_trace*" -size +2M -delete
timeout 1500 bash tools/gpu_pmc.sh > $OUT/pmc.log 2>&1
python tools/pmc_traffic.py $OUT/pmc_summary.json $OUT/pmc_traffic.json $OUT/pmc/p4/p4_kernel_trace.csv > /dev/null 2>&1
cp $OUT/pmc_traffic.json profiles/pmc_traffic.json
timeout 600 python bench.py > $OUT/${R}_bench.json 2> $OUT/bench.err; echo "bench exit $?"
python - <<PY
import json
r=json.loads(open('gpurun_out/r05_bench.json').read().strip().splitlines()[-1])
print(round(r['ms_per_step'],3), {a: round(b,3) for a,b in r['stage_ms'].items()}, r['roofline']['frac'], r['roofline']['counter_pass'].get('stale'))
PY
