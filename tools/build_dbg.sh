#!/bin/sh
# Instrumented build of the library (tools/bin/libmsae_dbg.so, not shipped): -DMSAE_RESCORE_DEBUG makes the
# re-score kernel report (rounds << 24 | first-round rows << 12 | rows) in `status` of verified tokens.
set -e
HERE="$(cd "$(dirname "$0")" && pwd)"
SRC="$HERE/../multimodal-sae_amd/csrc"
OUT="$HERE/bin"
OBJ="$OUT/obj_$$"
mkdir -p "$OBJ"
HIPCC="${HIPCC:-/opt/rocm/bin/hipcc}"
NAME="${MSAE_DBG_NAME:-libmsae_dbg.so}"
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -munsafe-fp-atomics -Wno-unused-function ${MSAE_DBG_FLAGS--DMSAE_RESCORE_DEBUG}"
OBJS=""
for f in capi decode topk sparsify encode_f32 encode_fused train; do
  "$HIPCC" $FLAGS -c "$SRC/$f.hip" -o "$OBJ/$f.o" &
  OBJS="$OBJS $OBJ/$f.o"
done
wait
"$HIPCC" --offload-arch=gfx950 -shared -fPIC $OBJS -o "$OUT/$NAME"
rm -rf "$OBJ"
echo "built $OUT/$NAME"
