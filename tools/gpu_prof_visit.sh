#!/bin/bash
# Development tool (MI355X box): per-phase cycles of a visit of the pipe kernel from the
# -DSTEREO_HIP_VISIT_PROFILE flavour of the library (stereo_amd/libstereo_hip_prof.so), for the border chain
# (run 0) and an interior row, forward and backward sweeps, on a volume.   tools/gpu_prof_visit.sh <tag> [volume]
tag=${1:-prof}; vol=${2:-teddy}
out=gpurun_out; mkdir -p $out
export STEREO_HIP_LIB=$PWD/stereo_amd/libstereo_hip_prof.so STEREO_HIP_TRWS_PROF=1
{
for run in 0 200; do for dbg in 4 2; do
  echo "== run $run debug $dbg (4: forward sweeps only, 2: backward only) volume $vol"
  STEREO_HIP_TRWS_PROF_RUN=$run STEREO_HIP_TRWS_DEBUG=$dbg timeout 300 python tools/time_trws.py 1 375 450 60 8 10 0 $vol 2>&1 | grep -v amdgpu.ids
done; done
} > $out/${tag}_visit_profile.txt 2>&1
cat $out/${tag}_visit_profile.txt
