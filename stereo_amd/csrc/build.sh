#!/bin/bash
# Builds stereo_amd/libstereo_hip.so for gfx950 (cross-compiles without a GPU).
# -ffp-contract=off: the reference is SSE2 without FMA; contraction would change bits.
set -e
HERE="$(cd "$(dirname "$0")" && pwd)"
ROOT="$(cd "$HERE/../.." && pwd)"
HIPCC="${HIPCC:-/opt/rocm/bin/hipcc}"
OUT="$ROOT/stereo_amd/libstereo_hip.so"
SRCS=$(ls "$HERE"/*.hip "$HERE"/*.cpp)
"$HIPCC" --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -ffp-contract=off \
  -Wall -Wno-unused-function -I"$ROOT/include" -o "$OUT" $SRCS "$@"
echo "built $OUT"
