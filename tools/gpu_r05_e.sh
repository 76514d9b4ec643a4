#!/bin/bash
# Round 5, GPU visit E: pre_acts_f32_kernel profile (rate, clock/power, MFMA busy and wait counters)
cd "$(dirname "$0")/.."
OUT=gpurun_out/r05e
rm -rf $OUT; mkdir -p $OUT
export TMPDIR=/tmp
python tools/f32_probe.py 10 2>&1 | tail -1 | tee $OUT/f32_rate.txt
(for i in $(seq 1 40); do rocm-smi --showpower --showclocks 2>/dev/null | grep -E "sclk|Power" | tr '\n' ' '; echo; sleep 0.1; done) > $OUT/smi.txt 2>&1 &
python tools/f32_probe.py 40 2>&1 | tail -1
wait
grep -o "sclk[^)]*)\|[0-9.]* *W\|Power[^:]*: [0-9.]*" $OUT/smi.txt | head -0
sort $OUT/smi.txt | uniq -c | sort -rn | head -5
i=0
for ctrs in "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_LDS"; do
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $ctrs --kernel-trace --output-format csv -d $OUT/p$i -o p$i -- python tools/f32_probe.py 2 > $OUT/p$i.log 2>&1
  echo "pass $i exit $?"
done
python tools/pmc_summary.py $OUT/pmc_f32.json $OUT/p1 $OUT/p2 2>&1 | grep pre_acts
find $OUT -name "*.csv" -size +1M -delete
