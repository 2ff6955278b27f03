#!/bin/bash
# Development tool (MI355X box): after a change to the K <= 64 message routine / pipe kernel -- certificate tests,
# randomised certificate and whole-solve stress, the TRW-S parity tests, the timings of the two Teddy-sized volumes.
#   tools/gpu_pipe_check.sh <tag> [stress seconds]
tag=${1:-pc}; secs=${2:-60}
out=gpurun_out; mkdir -p $out
export PYTHONUNBUFFERED=1
{
timeout 900 python -m pytest tests/test_certificate_gpu.py tests/test_trws_gpu.py tests/test_full_runs_gpu.py -q -m gpu -x 2>&1 | tail -5
timeout $((secs + 120)) python tools/stress_certificate.py $secs 7000 2>&1 | tail -2
timeout $((secs + 120)) python tools/stress_trws.py $secs 511 2>&1 | tail -2
for v in teddy noise; do timeout 300 python tools/time_trws.py 1 375 450 60 8 20 0 $v 2>&1 | grep -v amdgpu; done
} > $out/${tag}_pipe_check.txt 2>&1
cat $out/${tag}_pipe_check.txt
