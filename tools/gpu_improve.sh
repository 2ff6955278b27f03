#!/bin/bash
# Development tool (MI355X box): the QPBO Improve loop with and without the confined relabelling
# (STEREO_HIP_QPBO_CONFINED) -- the QPBO / globalstereo parity tests, the randomised Improve stress
# against the reference library (also with the -DSTEREO_HIP_QPBO_CHECK_CONFINED build if present, which
# verifies after every confined relabelling that the heights are tight), and examples/example_global.py
# timed both ways.
#   tools/gpu_improve.sh <tag> [stress seconds]
r=${1:-imp}; secs=${2:-120}
out=gpurun_out; mkdir -p $out
export PYTHONUNBUFFERED=1
{
  timeout 1200 python -m pytest tests/test_rd_gpu.py tests/test_globalstereo_gpu.py tests/test_fusion_gpu.py tests/test_stress_gpu.py -x -q -m gpu 2>&1 | tail -5
  if [ -f stereo_amd/libstereo_hip_chk.so ]; then
    echo "=== check build"
    STEREO_HIP_LIB=stereo_amd/libstereo_hip_chk.so timeout $((secs + 200)) python tools/stress_improve.py $secs 4132 > $out/${r}_chk.log 2>&1; grep -c "confined check" $out/${r}_chk.log; grep "confined check" $out/${r}_chk.log | head -3; tail -2 $out/${r}_chk.log
    STEREO_HIP_LIB=stereo_amd/libstereo_hip_chk.so timeout 600 python examples/example_global.py 2>&1 | grep -v amdgpu.ids | tail -6
  fi
  timeout $((secs + 200)) python tools/stress_improve.py $secs 4131 2>&1 | tail -3
  for c in 0 1; do
    echo "=== STEREO_HIP_QPBO_CONFINED=$c"
    STEREO_HIP_QPBO_CONFINED=$c timeout 600 python examples/example_global.py 2>&1 | tail -1
    STEREO_HIP_QPBO_CONFINED=$c timeout 600 python examples/example_global.py 2>&1 | tail -1
  done
} > $out/${r}_improve.txt 2>&1
cat $out/${r}_improve.txt
