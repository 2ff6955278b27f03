#!/bin/bash
out=gpurun_out; mkdir -p $out
export PYTHONUNBUFFERED=1
bash tools/gpu_step.sh ${1:-r04c} 45 > /dev/null 2>&1
bash tools/gpu_prof_visit.sh ${1:-r04c} teddy > /dev/null 2>&1
tail -12 $out/${1:-r04c}_log.txt
grep -v amdgpu $out/${1:-r04c}_visit_profile.txt | grep 'run\|per wave\|wave 0\|of the message\|loader' 
