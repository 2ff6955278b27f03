#!/bin/bash
out=gpurun_out; mkdir -p $out; tag=${1:-mp}
export STEREO_HIP_LIB=$PWD/stereo_amd/libstereo_hip_mprof.so STEREO_HIP_TRWS_PROF=1
{ for v in teddy ncc; do echo "== $v"; timeout 300 python tools/time_trws.py 1 375 450 60 8 10 0 $v 2>&1 | grep -v amdgpu; done; } > $out/${tag}_mprof_pipe.txt 2>&1
cat $out/${tag}_mprof_pipe.txt
