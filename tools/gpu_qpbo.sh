#!/bin/bash
# Development tool (MI355X box): A/B of the QPBO kernel -- parity tests on the new build, then the
# timing tools (Teddy-sized move, example_global with its per-solve breakdown) on each library given.
#   tools/gpu_qpbo.sh <tag> [lib ...]     (default: libstereo_hip_base.so if present, then libstereo_hip.so)
r=${1:-qpbo}; shift
out=gpurun_out; mkdir -p $out
export PYTHONUNBUFFERED=1
libs="$@"
if [ -z "$libs" ]; then
  [ -f stereo_amd/libstereo_hip_base.so ] && libs="stereo_amd/libstereo_hip_base.so"
  libs="$libs stereo_amd/libstereo_hip.so"
fi
{
  timeout 1200 python -m pytest tests/test_rd_gpu.py tests/test_globalstereo_gpu.py tests/test_fusion_gpu.py tests/test_stress_gpu.py -x -q -m gpu 2>&1 | tail -5
  timeout 300 python tools/stress_improve.py 40 4133 2>&1 | tail -2
  timeout 300 python tools/stress_rd.py 40 4134 2>&1 | tail -2
  for lib in $libs; do
    echo "=== $lib"
    STEREO_HIP_LIB=$lib timeout 300 python tools/time_rd.py 2>&1 | tail -2
    STEREO_HIP_LIB=$lib timeout 600 python examples/example_global.py 2>&1 | tail -1
    STEREO_HIP_LIB=$lib timeout 600 python examples/example_global.py 2>&1 | tail -1
    STEREO_HIP_LIB=$lib STEREO_HIP_QPBO_VERBOSE=1 timeout 600 python examples/example_global.py 2>&1 | grep "relabels:" | tail -8
  done
} > $out/${r}_qpbo.txt 2>&1
cat $out/${r}_qpbo.txt
if [ -f stereo_amd/libstereo_hip_qprof.so ]; then
  STEREO_HIP_LIB=stereo_amd/libstereo_hip_qprof.so STEREO_HIP_QPBO_VERBOSE=1 timeout 600 python examples/example_global.py 2>&1 | grep -E "workgroup 0" | tail -3 | tee -a $out/${r}_qpbo.txt
fi
