#!/bin/bash
# Development tool (MI355X box): after a change to the wide-label kernel -- its parity tests, the strips tests, whole-solve
# stress, timings at 1500x1000x256 and 3000x2000x256.   tools/gpu_wide_check.sh <tag> [stress seconds]
tag=${1:-wc}; secs=${2:-60}
out=gpurun_out; mkdir -p $out
export PYTHONUNBUFFERED=1
{
timeout 1500 python -m pytest tests/test_trws_wide_gpu.py tests/test_strips_gpu.py tests/test_trws_quadratic_gpu.py -q -m gpu -x 2>&1 | tail -5
timeout $((secs + 120)) python tools/stress_trws.py $secs 611 2>&1 | tail -2
timeout 300 python tools/time_trws.py 1 1000 1500 256 8 4 2>&1 | grep -v amdgpu
timeout 600 python tools/time_trws.py 1 2000 3000 256 8 3 2>&1 | grep -v amdgpu
} > $out/${tag}_wide_check.txt 2>&1
cat $out/${tag}_wide_check.txt
