#!/bin/bash
# Development tool (MI355X box): the fp32 useful-cone-loop MEASUREMENT (SURVEY 8(d) configs 4 / 5 name an fp32 message
# variant): stereo_amd/libstereo_hip_fp32cones.so = the library built with -DSTEREO_WIDE_FP32_CONES
# (bash stereo_amd/csrc/build.sh -DSTEREO_WIDE_FP32_CONES into STEREO_HIP_OUT) against the product build, same volume.
out=gpurun_out; mkdir -p $out; tag=${1:-fp32}
{
for lib in stereo_amd/libstereo_hip.so stereo_amd/libstereo_hip_fp32cones.so; do
  echo "=== $lib"
  STEREO_HIP_LIB=$PWD/$lib timeout 600 python tools/time_trws.py 1 1000 1500 256 8 4 0 noise 2>&1 | grep -v amdgpu
  STEREO_HIP_LIB=$PWD/$lib timeout 600 python tools/time_trws.py 1 2000 3000 256 8 3 0 noise 2>&1 | grep -v amdgpu
done
} > $out/${tag}_fp32_cones.txt 2>&1
cat $out/${tag}_fp32_cones.txt
