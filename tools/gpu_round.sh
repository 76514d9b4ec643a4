#!/bin/bash
# One GPU-box visit: parity tests, smoke, bench, rocprof kernel stats.  Everything lands in
# gpurun_out/ (merged back by gpurun).  Each step has its own timeout and never aborts the rest.
cd "$(dirname "$0")/.."
OUT=gpurun_out
mkdir -p $OUT
export TMPDIR=/tmp
echo "== host ==" > $OUT/host.txt
(nproc; grep -m1 "model name" /proc/cpuinfo; free -g | head -2; rocm-smi --showproductname 2>/dev/null | head -8) >> $OUT/host.txt 2>&1
echo "== pytest -m gpu ==" 
timeout 900 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider ${PYTEST_ARGS} > $OUT/pytest_gpu.log 2>&1
echo "pytest exit $?" | tee -a $OUT/pytest_gpu.log
tail -40 $OUT/pytest_gpu.log
echo "== smoke =="
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1; echo "smoke exit $?" | tee -a $OUT/smoke.log; tail -5 $OUT/smoke.log
echo "== bench =="
timeout 600 python bench.py --steps ${BENCH_STEPS:-10} --warmup 3 > $OUT/bench.json 2> $OUT/bench.err; echo "bench exit $?"; cat $OUT/bench.json; tail -5 $OUT/bench.err
echo "== rocprof =="
rm -rf $OUT/prof
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -o r01 -- python bench.py --steps 5 --warmup 2 --no-cpu-baseline > $OUT/bench_prof.json 2> $OUT/rocprof.err; echo "rocprof exit $?"
find $OUT/prof -name "*stats*" | head; 
for f in $(find $OUT/prof -name "*kernel_stats*.csv" | head -1); do head -25 $f; done
# drop the bulky per-dispatch trace, keep the stats
find $OUT/prof -name "*kernel_trace*" -size +2M -delete
echo "== training step (BASELINE configs[3]) =="
timeout 300 python tools/train_step_bench.py > $OUT/train_step.txt 2>&1; tail -1 $OUT/train_step.txt
rm -rf $OUT/prof_train
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_train -o tr -- python tools/train_step_bench.py > /dev/null 2> $OUT/rocprof_train.err; echo "rocprof train exit $?"
find $OUT/prof_train -name "*kernel_trace*" -size +2M -delete
