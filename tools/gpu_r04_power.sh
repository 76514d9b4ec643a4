#!/bin/bash
# Round 4, verdict items 3 and 7: watts / shader clock / milliseconds of
#   (a) the product's candidate GEMM (tools/gemm_sweep, 256x256 tiles, 8 waves, LDS-DMA) against the 4-wave / 128x128-wave-tile
#       prototype (tools/gemm4w: AGPR-pinned accumulators, register staging, 2/3 of the fragment bytes) and its ablations,
#       ALL on round(N(0, 32)) int8 operands -- what a quantised residual stream and encoder rows look like to the multipliers;
#   (b) the pure MFMA loop on int8 and on e4m3 (v_mfma_scale_f32_32x32x64_f8f6f4, MX scale operands) with the product's operand
#       statistics: the ceiling of what an fp8 candidate GEMM could gain under the package power cap;
#   (c) the encode step with the bf16 candidate pass beside the int8 one.
# rocm-smi power / clock sampled beside each run (tools/smi_sample.sh).  Output: gpurun_out/r04_gemm4w_power.txt
cd "$(dirname "$0")/.."
OUT=gpurun_out; mkdir -p $OUT/power
R=$OUT/r04_gemm4w_power.txt
: > $R
S=tools/smi_sample.sh
echo "== (a) candidate GEMM, int8 k = 4096 B per row, T = 8192, N = 131072, operands round(N(0,32)) ==" >> $R
SWEEP_D=2048 SWEEP_CFG=1 SWEEP_SIGMA=32,32 bash $S product_8w 2 -- tools/bin/gemm_sweep 1200 >> $R 2>&1
for v in 3 4 5 6 0; do
  G4W_ONLY=$v G4W_SIGMA=32 G4W_REPS=1200 bash $S gemm4w_v$v 2 -- tools/bin/gemm4w >> $R 2>&1
done
SWEEP_D=2048 SWEEP_CFG=4 SWEEP_SIGMA=32,32 bash $S product_8w_nostaging 2 -- tools/bin/gemm_sweep 1200 >> $R 2>&1
echo "== (b) pure MFMA loop, 4 operand register sets, Gaussian operands ==" >> $R
MFMA_RATE_VARY=1 MFMA_RATE_GAUSS=i8 MFMA_RATE_ONLY_I8=1 MFMA_RATE_REPS=1200 bash $S mfma_int8_gauss 2 -- tools/bin/mfma_rate >> $R 2>&1
MFMA_RATE_VARY=1 MFMA_RATE_GAUSS=fp8 MFMA_RATE_ONLY_FP8=1 MFMA_RATE_REPS=1400 bash $S mfma_fp8_gauss 2 -- tools/bin/mfma_rate >> $R 2>&1
echo "== (c) encode + decode step (bench.py), int8 and bf16 candidate pass ==" >> $R
bash tools/power_probe.sh step_int8 "" 500 >> $R 2>&1
MSAE_COARSE=bf16 bash tools/power_probe.sh step_bf16 "" 350 >> $R 2>&1
cat $R
