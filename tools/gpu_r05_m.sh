#!/bin/bash
# last visit of the round: the GPU suite and the smoke on the committed tree
cd "$(dirname "$0")/.."
OUT=gpurun_out
timeout 1500 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider -s > $OUT/pytest_gpu.log 2>&1; echo "pytest exit $?" | tee -a $OUT/pytest_gpu.log
grep -E "passed|failed|error" $OUT/pytest_gpu.log | tail -3
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1; echo "smoke exit $?"; tail -1 $OUT/smoke.log
