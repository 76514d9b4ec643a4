#!/bin/bash
# Round-2 first GPU visit: parity suite (incl. hostile weights), smoke, bench, re-score rows/rounds, soak.
cd "$(dirname "$0")/.."
OUT=gpurun_out
mkdir -p $OUT
export TMPDIR=/tmp
(nproc; grep -m1 "model name" /proc/cpuinfo; rocm-smi --showproductname 2>/dev/null | head -8) > $OUT/host.txt 2>&1
echo "== pytest -m gpu =="
timeout 1500 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider -s ${PYTEST_ARGS} > $OUT/pytest_gpu.log 2>&1
echo "pytest exit $?" | tee -a $OUT/pytest_gpu.log
grep -E "status \{|passed|failed|error|wrong|soak|stale" $OUT/pytest_gpu.log | tail -60
echo "== smoke =="
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1; echo "smoke exit $?" | tee -a $OUT/smoke.log; tail -3 $OUT/smoke.log
echo "== bench =="
timeout 600 python bench.py --steps 10 --warmup 3 > $OUT/bench.json 2> $OUT/bench.err; echo "bench exit $?"; cat $OUT/bench.json; tail -5 $OUT/bench.err
MSAE_COARSE=bf16 timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline > $OUT/bench_bf16.json 2>> $OUT/bench.err; echo "bench bf16 exit $?"; cat $OUT/bench_bf16.json
echo "== rescore stats =="
MSAE_HIP_LIB=tools/bin/libmsae_dbg.so timeout 600 python tools/rescore_stats.py bench trained_like lognorm > $OUT/rescore_stats.txt 2>&1; cat $OUT/rescore_stats.txt | tail -12
echo "== soak =="
timeout 900 python tools/soak_fused.py --tokens 1048576 --N 32768 --d 1024 --out $OUT/soak_1M_n32768.json > $OUT/soak.log 2>&1; echo "soak exit $?"; tail -2 $OUT/soak.log
timeout 900 python tools/soak_fused.py --tokens 131072 --N 131072 --d 4096 --out $OUT/soak_128k_c2.json >> $OUT/soak.log 2>&1; echo "soak c2 exit $?"; tail -1 $OUT/soak.log
