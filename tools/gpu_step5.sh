#!/bin/bash
out=gpurun_out; mkdir -p $out; tag=${1:-r04n}
export PYTHONUNBUFFERED=1
{
  timeout 1200 python -m pytest tests/test_certificate_gpu.py tests/test_trws_gpu.py tests/test_stress_gpu.py -x -q -m gpu 2>&1 | tail -3
  timeout 200 python tools/stress_certificate.py 60 100 $out/${tag}_stress_certificate.json 2>&1 | tail -2
  for lib in stereo_amd/libstereo_hip_base.so stereo_amd/libstereo_hip.so stereo_amd/libstereo_hip_base.so stereo_amd/libstereo_hip.so; do
    echo "=== $lib"
    for v in ncc teddy; do STEREO_HIP_LIB=$PWD/$lib timeout 300 python tools/time_trws.py 1 375 450 60 8 20 0 $v 2>&1 | grep -v amdgpu; done
  done
} > $out/${tag}_log.txt 2>&1
cat $out/${tag}_log.txt
