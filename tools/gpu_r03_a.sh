#!/bin/bash
# Round 3, first GPU visit: the whole -m gpu suite (new config-size / two-rank / sharded-steering / legacy-seam tests
# included), smoke, bench (k = 32 and k = 256), the tile-major operand-delivery probe.
cd "$(dirname "$0")/.."
OUT=gpurun_out; R=r03
mkdir -p $OUT
export TMPDIR=/tmp
(nproc; grep -m1 "model name" /proc/cpuinfo; free -g | head -2; rocm-smi --showproductname 2>/dev/null | head -8) > $OUT/${R}_host.txt 2>&1
echo "== pytest -m gpu =="
timeout 1500 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider --durations=15 > $OUT/${R}_pytest_gpu.log 2>&1
echo "pytest exit $?" | tee -a $OUT/${R}_pytest_gpu.log
grep -E "passed|failed|error" $OUT/${R}_pytest_gpu.log | tail -5
tail -40 $OUT/${R}_pytest_gpu.log | cut -c1-300
echo "== smoke =="
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1; echo "smoke exit $?"; tail -2 $OUT/smoke.log
echo "== bench =="
timeout 600 python bench.py --steps 20 --warmup 5 > $OUT/${R}_bench.json 2> $OUT/bench.err; echo "bench exit $?"; cut -c1-1500 $OUT/${R}_bench.json; tail -3 $OUT/bench.err
timeout 600 python bench.py --steps 10 --warmup 3 --k 256 --no-cpu-baseline > $OUT/${R}_bench_k256.json 2>> $OUT/bench.err; echo "bench k256 exit $?"; cut -c1-1500 $OUT/${R}_bench_k256.json
echo "== tile-major delivery probe =="
timeout 300 tools/bin/dma_depth 8192 131072 256 0 > $OUT/${R}_dma_tile_major.txt 2>&1; echo "probe exit $?"
timeout 300 tools/bin/dma_depth 8192 131072 256 1 > $OUT/${R}_dma_tile_major_random.txt 2>&1; echo "probe (random data) exit $?"
cat $OUT/${R}_dma_tile_major.txt
