#!/bin/bash
# Development tool (MI355X box): A/B timing of library builds under stereo_amd/csrc/_variants/*.so (STEREO_HIP_LIB).
out=gpurun_out; mkdir -p $out; tag=${1:-var}
export PYTHONUNBUFFERED=1 STEREO_HIP_TRWS_SPIN_SECONDS=3
{
for lib in stereo_amd/csrc/_variants/*.so; do
  echo "== $lib"
  for v in ${VOLUMES:-teddy noise}; do
    STEREO_HIP_LIB=$PWD/$lib STEREO_HIP_TRWS_TIMELINE=1 timeout 300 python tools/time_trws.py 1 375 450 60 8 5 0 $v 2>&1 | grep "speculative: runner\|stereo_hip spec\|recurrence" | cut -c1-400
    STEREO_HIP_LIB=$PWD/$lib timeout 300 python tools/time_trws.py 1 375 450 60 8 ${ITERS:-20} 0 $v 2>&1 | grep "it/s"
    if [ -n "$PLAIN" ]; then STEREO_HIP_TRWS_SPEC=0 STEREO_HIP_LIB=$PWD/$lib timeout 300 python tools/time_trws.py 1 375 450 60 8 ${ITERS:-20} 0 $v 2>&1 | grep "it/s" | sed 's/^/plain: /'; fi
  done
done
} > $out/${tag}_variants.txt 2>&1
cat $out/${tag}_variants.txt
