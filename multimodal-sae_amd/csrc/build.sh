#!/bin/sh
# Builds libmsae_hip.so for gfx950 in-tree (the .so travels to the GPU box with the snapshot).
# -ffp-contract=off: every fused multiply-add in the kernels is an explicit fma/MFMA, so device
# arithmetic matches oracle/sae_oracle.c operation for operation.
set -e
HERE="$(cd "$(dirname "$0")" && pwd)"
OUT="$HERE/../msae/_lib"
mkdir -p "$OUT"
HIPCC="${HIPCC:-/opt/rocm/bin/hipcc}"
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -munsafe-fp-atomics -Wall -Wno-unused-function -Wno-inline-asm"
OBJS=""
for f in capi decode topk sparsify encode_f32 encode_fused train; do
  "$HIPCC" $FLAGS -c "$HERE/$f.hip" -o "$OUT/$f.o" &
  OBJS="$OBJS $OUT/$f.o"
done
wait
"$HIPCC" --offload-arch=gfx950 -shared -fPIC $OBJS -o "$OUT/libmsae_hip.so"
rm -f $OBJS
echo "built $OUT/libmsae_hip.so"
