#!/bin/bash
# Development tool (MI355X box): the exchange-flag fix -- flag check, the full configs[1] run spec against plain, the spec and full-run tests.
out=gpurun_out; mkdir -p $out; tag=${1:-specfix}
export PYTHONUNBUFFERED=1
{
timeout 300 python tools/spec_flag_check.py 2>&1 | grep -v amdgpu
timeout 600 python tools/spec_full_check.py 700 100000 2>&1 | grep -v amdgpu | tail -8
timeout 900 python -m pytest tests/test_spec_gpu.py tests/test_full_runs_gpu.py tests/test_trws_gpu.py -x -q -m gpu 2>&1 | tail -5
} > $out/${tag}.txt 2>&1
cat $out/${tag}.txt
