#!/bin/bash
# Development tool (MI355X box): timings + timeline of the speculative schedule only.
out=gpurun_out; mkdir -p $out; tag=${1:-spect}
export PYTHONUNBUFFERED=1 STEREO_HIP_TRWS_SPIN_SECONDS=${SPIN:-3}
{
for v in teddy noise; do
  echo "== $v"
  STEREO_HIP_TRWS_TIMELINE=1 timeout 300 python tools/time_trws.py 1 375 450 60 8 5 0 $v 2>&1 | grep -v amdgpu | cut -c1-1200
  timeout 300 python tools/time_trws.py 1 375 450 60 8 20 0 $v 2>&1 | grep -v amdgpu
done
} > $out/${tag}_spect.txt 2>&1
cat $out/${tag}_spect.txt
