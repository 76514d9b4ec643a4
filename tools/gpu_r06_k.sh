#!/bin/bash
# round 6, visit K: the LDS race fix (m / -E parked in the row parts only) -- N = 262144, soaks, suite
cd "$(dirname "$0")/.."
OUT=gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
DBG_BATCHES=16 python tools/debug_n262144.py 2>&1 | grep -E "batch|wrong tokens" | cut -c1-300
DBG_KIND=gauss DBG_BATCHES=16 python tools/debug_n262144.py 2>&1 | grep -E "batch|wrong tokens" | cut -c1-300
timeout 1500 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider -x > $OUT/pytest_gpu.log 2>&1; echo "pytest exit $?"; tail -2 $OUT/pytest_gpu.log | cut -c1-200
timeout 1200 python tools/soak_fused.py --tokens 4194304 --N 262144 --d 4096 --out $OUT/r06_soak_4M_trained_like_n262144.json > $OUT/soak_w.log 2>&1; echo "soak N=262144 exit $?"; tail -1 $OUT/soak_w.log | cut -c1-500
timeout 1200 python tools/soak_fused.py --tokens 4194304 --N 131072 --d 4096 --out $OUT/r06_soak_4M_trained_like_c2.json > $OUT/soak_c2.log 2>&1; echo "soak C2 exit $?"; tail -1 $OUT/soak_c2.log | cut -c1-500
timeout 900 python tools/soak_fused.py --tokens 1048576 --N 65536 --d 2048 --kind gauss --out $OUT/r06_soak_1M_gauss_n65536.json > $OUT/soak_g.log 2>&1; echo "soak gauss N=65536 exit $?"; tail -1 $OUT/soak_g.log | cut -c1-400
