#!/bin/bash
# Development tool (MI355X box): A/B of two builds of the library on the same box -- the in-tree one and
# stereo_amd/libstereo_hip_base.so -- over the Teddy-sized volumes, shared and per-edge positions.
out=gpurun_out; mkdir -p $out; tag=${1:-ab}
export PYTHONUNBUFFERED=1
{
for lib in stereo_amd/libstereo_hip_base.so stereo_amd/libstereo_hip.so; do
  echo "=== $lib"
  for v in noise ncc teddy; do STEREO_HIP_LIB=$PWD/$lib timeout 300 python tools/time_trws.py 1 375 450 60 8 20 0 $v 2>&1 | grep -v amdgpu; done
  STEREO_HIP_LIB=$PWD/$lib timeout 300 python tools/time_trws.py 1 375 450 60 8 10 1 noise 2>&1 | grep -v amdgpu
  STEREO_HIP_LIB=$PWD/$lib timeout 300 python tools/time_trws.py 2 375 450 60 64 10 0 noise 2>&1 | grep -v amdgpu
  STEREO_HIP_LIB=$PWD/$lib timeout 300 python tools/time_trws.py 1 375 450 16 3 10 1 noise 2>&1 | grep -v amdgpu
done
} > $out/${tag}_ab.txt 2>&1
cat $out/${tag}_ab.txt
