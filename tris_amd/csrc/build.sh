#!/bin/bash
# Builds libtris_hip.so for gfx950 in-tree (tris_amd/libtris_hip.so).  hipcc cross-compiles without a GPU.
set -e
HERE="$(cd "$(dirname "$0")" && pwd)"
ROOT="$(cd "$HERE/../.." && pwd)"
HIPCC=${HIPCC:-/opt/rocm/bin/hipcc}
OUT="$HERE/../libtris_hip.so"
OBJ="$HERE/_obj"
mkdir -p "$OBJ"
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -I$ROOT/include -I$HERE"
pids=()
for f in gemm_conv conv_direct norm attn attn_mfma heads optim eval xattn xattn_fused data comm; do
  [ -f "$HERE/$f.hip" ] || continue
  stale=0
  for dep in "$HERE/$f.hip" "$HERE"/*.h "$ROOT/include/tris_hip.h"; do
    [ "$dep" -nt "$OBJ/$f.o" ] && stale=1
  done
  if [ ! -f "$OBJ/$f.o" ] || [ $stale = 1 ]; then
    $HIPCC $FLAGS -c "$HERE/$f.hip" -o "$OBJ/$f.o" &
    pids+=($!)
  fi
done
for p in "${pids[@]}"; do wait $p; done
$HIPCC --offload-arch=gfx950 -shared -fPIC -o "$OUT" "$OBJ"/*.o
echo "built $OUT"
