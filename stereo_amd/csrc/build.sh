#!/bin/bash
# Builds stereo_amd/libstereo_hip.so for gfx950 (cross-compiles without a GPU): one hipcc -c per
# translation unit, in parallel, then one link.
# -ffp-contract=off: the reference is SSE2 without FMA; contraction would change bits.
# The device assembly is kept (-save-temps, under csrc/_build/) and checked for one known
# miscompile of this toolchain: s_mov_b64 with a 64-bit literal, which the encoder truncates
# to its low 32 bits (a wave-uniform +inf became 0.0 that way).
#: trws_dev.h names m0 as clobbered by its v_writelane sequence (the lane number
# travels in m0); clang warns about any reserved register on a clobber list.
set -e
HERE="$(cd "$(dirname "$0")" && pwd)"
ROOT="$(cd "$HERE/../.." && pwd)"
HIPCC="${HIPCC:-/opt/rocm/bin/hipcc}"
OUT="${STEREO_HIP_OUT:-$ROOT/stereo_amd/libstereo_hip.so}"
TMP="${STEREO_HIP_TMP:-$HERE/_build}"
mkdir -p "$TMP"
rm -f "$TMP"/*.o "$TMP"/*.log
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -Wall -Wno-unused-function -Wno-unused-command-line-argument -I$ROOT/include -save-temps=obj"
pids=()
objs=()
for src in "$HERE"/*.hip "$HERE"/*.cpp; do
  base="$(basename "${src%.*}")"
  ( cd "$TMP" && "$HIPCC" $FLAGS -c "$src" -o "$TMP/$base.o" "$@" > "$TMP/$base.log" 2>&1 ) &
  pids+=($!)
  objs+=("$TMP/$base.o")
done
fail=0
for pid in "${pids[@]}"; do wait "$pid" || fail=1; done
cat "$TMP"/*.log
if [ "$fail" != 0 ]; then echo "error: compilation failed" >&2; exit 1; fi
if grep -nE 's_mov_b64 s\[[0-9:]+\], 0x[0-9a-f]{9,}' "$TMP"/*-hip-amdgcn-amd-amdhsa-gfx950.s; then
  echo "error: s_mov_b64 with a 64-bit literal in the device code (mis-encoded by this toolchain)" >&2
  exit 1
fi
"$HIPCC" --offload-arch=gfx950 -shared -fPIC -o "$TMP/libstereo_hip.so" "${objs[@]}"
mv "$TMP/libstereo_hip.so" "$OUT"
rm -f "$TMP"/*.bc "$TMP"/*.hipi "$TMP"/*.out "$TMP"/*.txt "$TMP"/*.hipfb "$TMP"/*-host-*.s "$TMP"/*.o
echo "built $OUT"
