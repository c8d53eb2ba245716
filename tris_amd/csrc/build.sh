#!/bin/bash
# Builds libtris_hip.so for gfx950 in-tree (tris_amd/libtris_hip.so).  hipcc cross-compiles without a GPU.
# The GEMM family (gemm_inst.hip) is compiled once per operand-kind pair, all units side by side.
set -e
HERE="$(cd "$(dirname "$0")" && pwd)"
ROOT="$(cd "$HERE/../.." && pwd)"
HIPCC=${HIPCC:-/opt/rocm/bin/hipcc}
OUT="$HERE/../libtris_hip.so"
OBJ="$HERE/_obj"
mkdir -p "$OBJ"
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -I$ROOT/include -I$HERE"
pids=()
fail=0
stale() {  # stale <object> <source>: the object is missing or older than the source, any header, or this script
  local o="$1" src="$2"
  [ -f "$o" ] || return 0
  for dep in "$src" "$HERE"/*.h "$ROOT/include/tris_hip.h" "$HERE/build.sh"; do
    [ "$dep" -nt "$o" ] && return 0
  done
  return 1
}
# A kind / B kind pairs of the GEMM family (gemm_params.h): ROWK x NK, ROWK x KN, COLK x NK, COLK x KN, IM2COL x NK,
# IM2COL x KN_DGRAD, COLK x KN_IM2COL
for pair in 00 01 10 11 20 22 13; do
  o="$OBJ/gemm_inst_$pair.o"
  if stale "$o" "$HERE/gemm_inst.hip"; then
    $HIPCC $FLAGS -DTRIS_GEMM_AK=${pair:0:1} -DTRIS_GEMM_BK=${pair:1:1} -c "$HERE/gemm_inst.hip" -o "$o" &
    pids+=($!)
  fi
done
for prec in 1 3 4; do   # the direct 3x3 kernels, one unit per arithmetic (x3, h2, h2 on operand planes)
  o="$OBJ/conv_direct_$prec.o"
  if stale "$o" "$HERE/conv_direct.hip"; then
    $HIPCC $FLAGS -DTRIS_DIRECT_PREC=$prec -c "$HERE/conv_direct.hip" -o "$o" &
    pids+=($!)
  fi
done
for f in gemm_conv norm planes attn attn_mfma attn_h2 heads optim eval xattn xattn_fused xattn_px data comm; do
  [ -f "$HERE/$f.hip" ] || continue
  if stale "$OBJ/$f.o" "$HERE/$f.hip"; then
    $HIPCC $FLAGS -c "$HERE/$f.hip" -o "$OBJ/$f.o" &
    pids+=($!)
  fi
done
for p in "${pids[@]}"; do wait $p || fail=1; done
[ $fail = 0 ] || { echo "build failed" >&2; exit 1; }
rm -f "$OBJ/conv_direct.o"   # (objects of earlier layouts of this directory)
$HIPCC --offload-arch=gfx950 -shared -fPIC -o "$OUT" "$OBJ"/*.o
echo "built $OUT"
