#!/bin/bash
# Development tool (MI355X box): per-phase cycles of a visit of the wide kernel (-DSTEREO_HIP_MESSAGE_PROFILE flavour,
# stereo_amd/libstereo_hip_mprof.so) for the border chain (run 0) and an interior row, forward / backward sweeps.
tag=${1:-wprof}; out=gpurun_out; mkdir -p $out
export STEREO_HIP_LIB=$PWD/stereo_amd/libstereo_hip_mprof.so STEREO_HIP_TRWS_PROF=1 PYTHONUNBUFFERED=1
{
for run in 0 300; do for dbg in 4; do
  echo "== run $run debug $dbg (4: forward sweeps only, 2: backward only)"
  STEREO_HIP_TRWS_PROF_RUN=$run STEREO_HIP_TRWS_DEBUG=$dbg timeout 600 python tools/time_trws.py 1 1000 1500 256 8 4 0 noise 2>&1 | grep -v amdgpu.ids
done; done
} > $out/${tag}_wide_profile.txt 2>&1
cat $out/${tag}_wide_profile.txt
