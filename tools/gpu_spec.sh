#!/bin/bash
# Development tool (MI355X box): the speculative schedule -- equality check, timelines, timings on / off.
out=gpurun_out; mkdir -p $out; tag=${1:-spec}
export PYTHONUNBUFFERED=1 STEREO_HIP_TRWS_SPIN_SECONDS=${SPIN:-3}
{
timeout 900 python tools/spec_check.py ${ITERS:-5} 2>&1 | grep -v amdgpu
for v in teddy noise; do
  for sp in 1 0; do
    echo "== $v spec=$sp"
    STEREO_HIP_TRWS_SPEC=$sp STEREO_HIP_TRWS_TIMELINE=1 timeout 300 python tools/time_trws.py 1 375 450 60 8 5 0 $v 2>&1 | grep -v amdgpu | cut -c1-1500
    STEREO_HIP_TRWS_SPEC=$sp timeout 300 python tools/time_trws.py 1 375 450 60 8 20 0 $v 2>&1 | grep -v amdgpu
  done
done
} > $out/${tag}_spec.txt 2>&1
tail -c 7000 $out/${tag}_spec.txt
