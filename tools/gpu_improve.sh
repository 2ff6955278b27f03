#!/bin/bash
# Development tool (MI355X box): the QPBO Improve loop with and without the confined relabelling
# (STEREO_HIP_QPBO_CONFINED) -- the QPBO / globalstereo parity tests, the randomised Improve stress
# against the reference library, and examples/example_global.py timed both ways.
#   tools/gpu_improve.sh <tag> [stress seconds]
r=${1:-imp}; secs=${2:-120}
out=gpurun_out; mkdir -p $out
export PYTHONUNBUFFERED=1
{
  timeout 1200 python -m pytest tests/test_rd_gpu.py tests/test_globalstereo_gpu.py tests/test_fusion_gpu.py tests/test_stress_gpu.py -x -q -m gpu 2>&1 | tail -5
  timeout $((secs + 200)) python tools/stress_improve.py $secs 4131 2>&1 | tail -3
  for c in 0 1; do
    echo "=== STEREO_HIP_QPBO_CONFINED=$c"
    STEREO_HIP_QPBO_CONFINED=$c timeout 600 python examples/example_global.py 2>&1 | tail -4
    STEREO_HIP_QPBO_CONFINED=$c timeout 600 python examples/example_global.py 2>&1 | tail -2
  done
  echo "=== verbose (confined)"
  STEREO_HIP_QPBO_VERBOSE=1 timeout 600 python examples/example_global.py 2>&1 | grep -v "active per round" | tail -60
} > $out/${r}_improve.txt 2>&1
cat $out/${r}_improve.txt
