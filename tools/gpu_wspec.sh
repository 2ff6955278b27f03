#!/bin/bash
# Development tool (MI355X box): the wide kernel's speculative schedule -- tests, then timings spec against plain.
out=gpurun_out; mkdir -p $out; tag=${1:-wspec}
export PYTHONUNBUFFERED=1 STEREO_HIP_TRWS_SPIN_SECONDS=${SPIN:-5}
{
timeout 900 python -m pytest tests/test_wspec_gpu.py -x -q -m gpu 2>&1 | tail -15
for spec in 1 0; do
  echo "== STEREO_HIP_TRWS_SPEC=$spec 1500x1000x256"
  STEREO_HIP_TRWS_SPEC=$spec timeout 300 python tools/time_trws.py 1 1000 1500 256 8 4 0 noise 2>&1 | grep -v amdgpu | tail -3
done
} > $out/${tag}.txt 2>&1
cut -c1-1200 $out/${tag}.txt
