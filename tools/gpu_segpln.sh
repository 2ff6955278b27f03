#!/bin/bash
# Development tool (MI355X box): SegPln plane-fit tests + timing of the 14 maps.
out=gpurun_out; mkdir -p $out; tag=${1:-segpln}
export PYTHONUNBUFFERED=1
{
timeout 900 python -m pytest tests/test_segpln_gpu.py tests/test_segment_gpu.py -x -q -m gpu 2>&1 | tail -8
timeout 300 python tools/time_segpln.py 2>&1 | grep -v amdgpu | tail -18
} > $out/${tag}.txt 2>&1
cat $out/${tag}.txt
