#!/bin/bash
# Development tool (runs on the MI355X box through gpurun): the parity tests of the TRW-S message path,
# the timing of the three Teddy-sized volumes and a short certificate stress, into gpurun_out/<tag>_*.
#   tools/gpu_step.sh <tag> [stress seconds]
tag=${1:-step}; secs=${2:-60}
out=gpurun_out; mkdir -p $out
export PYTHONUNBUFFERED=1
{
  [ -x tools/micro_valu.bin ] && tools/micro_valu.bin
  timeout 900 python -m pytest tests/test_certificate_gpu.py tests/test_trws_gpu.py tests/test_edge_cases_gpu.py tests/test_stress_gpu.py -x -q -m gpu 2>&1 | tail -5
  for v in noise ncc teddy; do timeout 300 python tools/time_trws.py 1 375 450 60 8 20 0 $v; done
  timeout 300 python tools/time_trws.py 2 375 450 60 64 20 0 noise
  timeout $((secs + 120)) python tools/stress_certificate.py $secs 0 $out/${tag}_stress_certificate.json
} > $out/${tag}_log.txt 2>&1
tail -40 $out/${tag}_log.txt
