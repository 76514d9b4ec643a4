#!/bin/sh
# usage: tools/kernel_regs.sh [pattern] [extra -D flags]  -> register / scratch usage of the kernels in encode_fused.hip
cd "$(dirname "$0")/.."
PAT="${1:-gemm_kernel}"; shift
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -munsafe-fp-atomics -Wno-unused-function -Wno-inline-asm \
  -Rpass-analysis=kernel-resource-usage "$@" -c multimodal-sae_amd/csrc/encode_fused.hip -o /tmp/ef_$$.o 2>/tmp/ef_$$.err
grep "remark:" /tmp/ef_$$.err | grep -A9 "$PAT" | grep -E "Function Name| VGPRs:|TotalSGPRs|ScratchSize|VGPRs Spill" | \
  sed 's/.*remark: *//; s/ \[-Rpass.*//' | paste - - - - - | sed 's/Function Name: //' | cut -c1-230
rm -f /tmp/ef_$$.o /tmp/ef_$$.err
