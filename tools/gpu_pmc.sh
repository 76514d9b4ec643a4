#!/bin/bash
# PMC passes (each in its own run, kernel-trace only -- never combined with sys/hip traces).
cd "$(dirname "$0")/.."
OUT=gpurun_out/pmc
rm -rf $OUT; mkdir -p $OUT
export TMPDIR=/tmp
CMD="python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-secondary"
i=0
for ctrs in "FETCH_SIZE" "WRITE_SIZE" "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU_MFMA_MOPS_BF16 GRBM_GUI_ACTIVE" "TCC_HIT_sum TCC_MISS_sum"; do
  i=$((i+1))
  timeout 600 rocprofv3 --pmc $ctrs --kernel-trace --output-format csv -d $OUT/p$i -o p$i -- $CMD > $OUT/p$i.log 2>&1
  echo "pass $i ($ctrs) exit $?"
done
python tools/pmc_summary.py gpurun_out/pmc_summary.json $OUT/p1 $OUT/p2 $OUT/p3 $OUT/p4 $OUT/p5
find $OUT -name "*.csv" -size +1M -delete
