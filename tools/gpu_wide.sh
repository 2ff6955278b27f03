#!/bin/bash
# Development tool (MI355X box): wide-kernel parity tests and A/B timing against stereo_amd/libstereo_hip_base.so
out=gpurun_out; mkdir -p $out; tag=${1:-wide}
export PYTHONUNBUFFERED=1
{
  timeout 1500 python -m pytest tests/test_trws_wide_gpu.py tests/test_strips_gpu.py -x -q -m gpu 2>&1 | tail -4
  for lib in stereo_amd/libstereo_hip_base.so stereo_amd/libstereo_hip.so; do
    echo "=== $lib"
    STEREO_HIP_LIB=$PWD/$lib timeout 600 python tools/time_trws.py 1 1000 1500 256 8 4 0 noise 2>&1 | grep -v amdgpu
    STEREO_HIP_LIB=$PWD/$lib timeout 600 python tools/time_trws.py 1 2000 3000 256 8 3 0 noise 2>&1 | grep -v amdgpu
    STEREO_HIP_LIB=$PWD/$lib timeout 600 python tools/time_trws.py 1 1000 1500 256 8 4 0 noise 0 1 2>&1 | grep -v amdgpu
  done
} > $out/${tag}_wide.txt 2>&1
cat $out/${tag}_wide.txt
