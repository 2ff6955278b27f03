#!/bin/bash
out=gpurun_out; mkdir -p $out; tag=${1:-tlw}
{ STEREO_HIP_TRWS_TIMELINE=1 timeout 600 python tools/time_trws.py 1 2000 3000 256 8 3 0 noise 2>&1 | grep -v amdgpu; } > $out/${tag}_timeline_wide.txt 2>&1
cut -c1-1500 $out/${tag}_timeline_wide.txt
