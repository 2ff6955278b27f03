#!/bin/bash
out=gpurun_out; mkdir -p $out; tag=${1:-r04d}
export PYTHONUNBUFFERED=1
{
  timeout 1200 python -m pytest tests/test_certificate_gpu.py tests/test_trws_gpu.py tests/test_edge_cases_gpu.py tests/test_stress_gpu.py tests/test_strips_gpu.py tests/test_simultaneous_gpu.py tests/test_trws_quadratic_gpu.py tests/test_trws_wide_gpu.py -x -q -m gpu 2>&1 | tail -5
} > $out/${tag}_tests.txt 2>&1
bash tools/gpu_ab.sh $tag > /dev/null 2>&1
tail -5 $out/${tag}_tests.txt; cat $out/${tag}_ab.txt
