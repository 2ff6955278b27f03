#!/bin/bash
# Builds stereo_amd/libstereo_hip.so for gfx950 (cross-compiles without a GPU).
# -ffp-contract=off: the reference is SSE2 without FMA; contraction would change bits.
# The device assembly is kept (-save-temps, under csrc/_build/) and checked for one known
# miscompile of this toolchain: s_mov_b64 with a 64-bit literal, which the encoder truncates
# to its low 32 bits (a wave-uniform +inf became 0.0 that way).
set -e
HERE="$(cd "$(dirname "$0")" && pwd)"
ROOT="$(cd "$HERE/../.." && pwd)"
HIPCC="${HIPCC:-/opt/rocm/bin/hipcc}"
OUT="$ROOT/stereo_amd/libstereo_hip.so"
TMP="$HERE/_build"
mkdir -p "$TMP"
SRCS=$(ls "$HERE"/*.hip "$HERE"/*.cpp)
"$HIPCC" --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -ffp-contract=off \
  -Wall -Wno-unused-function -Wno-unused-command-line-argument -I"$ROOT/include" \
  -save-temps=obj -o "$TMP/libstereo_hip.so" $SRCS "$@"
if grep -nE 's_mov_b64 s\[[0-9:]+\], 0x[0-9a-f]{9,}' "$TMP"/*-hip-amdgcn-amd-amdhsa-gfx950.s; then
  echo "error: s_mov_b64 with a 64-bit literal in the device code (mis-encoded by this toolchain)" >&2
  exit 1
fi
mv "$TMP/libstereo_hip.so" "$OUT"
rm -f "$TMP"/*.bc "$TMP"/*.hipi "$TMP"/*.o "$TMP"/*.out "$TMP"/*.txt "$TMP"/*.hipfb "$TMP"/*-host-*.s
echo "built $OUT"
