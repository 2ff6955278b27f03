#!/bin/bash
# Development tool (MI355X box): wall-clock timeline of a sweep (STEREO_HIP_TRWS_TIMELINE) on the Teddy-sized volumes.
out=gpurun_out; mkdir -p $out; tag=${1:-tl}
{
for v in teddy noise; do
  echo "== $v"
  STEREO_HIP_TRWS_TIMELINE=1 timeout 300 python tools/time_trws.py 1 375 450 60 8 5 0 $v 2>&1 | grep -v amdgpu
done
} > $out/${tag}_timeline.txt 2>&1
cat $out/${tag}_timeline.txt
