#!/bin/bash
# Development tool (MI355X box): timeline of the wide kernel's speculative schedule.
out=gpurun_out; mkdir -p $out; tag=${1:-wspectl}
export PYTHONUNBUFFERED=1 STEREO_HIP_TRWS_SPIN_SECONDS=${SPIN:-5}
{
for sz in "1000 1500" "2000 3000"; do
  for spec in 1 0; do
    echo "== STEREO_HIP_TRWS_SPEC=$spec $sz x 256"
    STEREO_HIP_TRWS_SPEC=$spec STEREO_HIP_TRWS_TIMELINE=1 timeout 400 python tools/time_trws.py 1 $sz 256 8 3 0 noise 2>&1 | grep -v amdgpu | cut -c1-1800
  done
done
} > $out/${tag}.txt 2>&1
cat $out/${tag}.txt
