#!/bin/bash
# Development tool (MI355X box): tools/spec_full_check.py, log under gpurun_out/.
out=gpurun_out; mkdir -p $out; tag=${1:-specfull}; shift
export PYTHONUNBUFFERED=1
timeout 900 python tools/spec_full_check.py "$@" > $out/${tag}.txt 2>&1
grep -v amdgpu $out/${tag}.txt | tail -40
