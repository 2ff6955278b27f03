#!/bin/bash
# Development tool (MI355X box): segmenter tests + timings, the gateway-strips test.
out=gpurun_out; mkdir -p $out; tag=${1:-segment}
export PYTHONUNBUFFERED=1
{
timeout 900 python -m pytest tests/test_segment_gpu.py tests/test_segment_cpu.py -x -q 2>&1 | tail -15
timeout 600 python -m pytest tests/test_strips_gpu.py -x -q -m gpu 2>&1 | tail -5
timeout 300 python tools/time_segment.py teddy 2>&1 | grep -v amdgpu
timeout 300 python tools/time_segment.py baby2 2>&1 | grep -v amdgpu
timeout 300 python examples/example_global.py --teddy 2>&1 | grep -v amdgpu
} > $out/${tag}.txt 2>&1
cat $out/${tag}.txt
