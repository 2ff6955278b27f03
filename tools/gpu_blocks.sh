#!/bin/bash
# Development tool (MI355X box): workgroups per pipelined sweep launch (STEREO_HIP_TRWS_BLOCKS), spec on.
out=gpurun_out; mkdir -p $out; tag=${1:-blk}
export PYTHONUNBUFFERED=1 STEREO_HIP_TRWS_SPIN_SECONDS=3
{
for b in 0 256 255 300; do
  echo "== blocks $b"
  for v in teddy noise; do
    STEREO_HIP_TRWS_BLOCKS=$b timeout 300 python tools/time_trws.py 1 375 450 60 8 20 0 $v 2>&1 | grep "it/s"
  done
done
timeout 600 python bench.py --no-scale --no-extras 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print({k:d[k] for k in ('value','ms_per_step')}, d['roofline']['avg_launch_us'])"
} > $out/${tag}_blocks.txt 2>&1
cat $out/${tag}_blocks.txt
