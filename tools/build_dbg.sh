#!/bin/sh
# Instrumented / ablation builds of the library (tools/bin/<MSAE_DBG_NAME>, not shipped): MSAE_DBG_FLAGS carries the -D flags of
# multimodal-sae_amd/csrc/tuning.h (e.g. -DMSAE_GEMM_TIMELINE, -DMSAE_RESCORE_TL, -DMSAE_ABL_NOEPI, -DMSAE_GEMM_STAGGER=0).
set -e
HERE="$(cd "$(dirname "$0")" && pwd)"
SRC="$HERE/../multimodal-sae_amd/csrc"
OUT="$HERE/bin"
OBJ="$OUT/obj_$$"
mkdir -p "$OBJ"
HIPCC="${HIPCC:-/opt/rocm/bin/hipcc}"
NAME="${MSAE_DBG_NAME:-libmsae_dbg.so}"
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -munsafe-fp-atomics -Wno-unused-function ${MSAE_DBG_FLAGS-}"
OBJS=""
for f in capi decode topk sparsify encode_f32 encode_fused train; do
  "$HIPCC" $FLAGS -c "$SRC/$f.hip" -o "$OBJ/$f.o" &
  OBJS="$OBJS $OBJ/$f.o"
done
wait
"$HIPCC" --offload-arch=gfx950 -shared -fPIC $OBJS -o "$OUT/$NAME"
rm -rf "$OBJ"
echo "built $OUT/$NAME"
