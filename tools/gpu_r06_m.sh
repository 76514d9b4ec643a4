#!/bin/bash
cd "$(dirname "$0")/.."
OUT=gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider -x > $OUT/pytest_gpu.log 2>&1; echo "pytest exit $?"; tail -2 $OUT/pytest_gpu.log | cut -c1-200
timeout 300 python tools/latency_small_T.py 2>&1 | grep "T=" | cut -c1-40
timeout 300 python tools/stage_small_T.py 64 256 257 512 1024 2>&1 | grep "T=" | cut -c1-200
timeout 600 python bench.py --k 256 --steps 10 --warmup 3 --no-cpu-baseline --no-secondary 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('k256', d['ms_per_step'], {k: round(v,3) for k,v in d['stage_ms'].items()})"
(timeout 500 python tools/fuzz_fused.py 2000 71; timeout 300 python tools/soak_fused.py --tokens 262144 --N 131072 --d 4096 --k 256 --out $OUT/r06_soak_k256_trained_like_c2.json | tail -1 | cut -c1-300) 2>&1 | grep -E "cases|silent"
