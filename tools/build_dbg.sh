#!/bin/sh
# Instrumented build of the library (tools/bin/libmsae_dbg.so, not shipped): -DMSAE_RESCORE_DEBUG makes the
# re-score kernel report (rounds << 24 | first-round rows << 12 | rows) in `status` of verified tokens.
set -e
HERE="$(cd "$(dirname "$0")" && pwd)"
SRC="$HERE/../multimodal-sae_amd/csrc"
OUT="$HERE/bin"
mkdir -p "$OUT/obj"
HIPCC="${HIPCC:-/opt/rocm/bin/hipcc}"
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -munsafe-fp-atomics -Wno-unused-function -DMSAE_RESCORE_DEBUG $MSAE_DBG_FLAGS"
OBJS=""
for f in capi decode topk sparsify encode_f32 encode_fused train; do
  "$HIPCC" $FLAGS -c "$SRC/$f.hip" -o "$OUT/obj/$f.o" &
  OBJS="$OBJS $OUT/obj/$f.o"
done
wait
"$HIPCC" --offload-arch=gfx950 -shared -fPIC $OBJS -o "$OUT/libmsae_dbg.so"
rm -rf "$OUT/obj"
echo "built $OUT/libmsae_dbg.so"
