#!/bin/bash
out=gpurun_out; mkdir -p $out; tag=${1:-qw}
export PYTHONUNBUFFERED=1
{
  timeout 1500 python -m pytest tests/test_trws_quadratic_gpu.py tests/test_trws_wide_gpu.py tests/test_strips_gpu.py -x -q -m gpu 2>&1 | tail -12
  for lib in stereo_amd/libstereo_hip_base.so stereo_amd/libstereo_hip.so; do
    echo "=== $lib"
    STEREO_HIP_LIB=$PWD/$lib timeout 600 python tools/time_trws.py 2 375 450 100 64 5 0 noise 2>&1 | grep -v amdgpu
    STEREO_HIP_LIB=$PWD/$lib timeout 600 python tools/time_trws.py 2 375 450 256 64 3 0 noise 2>&1 | grep -v amdgpu
  done
} > $out/${tag}_quadwide.txt 2>&1
cat $out/${tag}_quadwide.txt
