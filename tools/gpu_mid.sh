#!/bin/bash
# Development tool (MI355X box): mid-round check -- timings of the headline workloads, the TRW-S / strips / spec suites,
# the globalstereo parity test with its tie accounting shown.
out=gpurun_out; mkdir -p $out; tag=${1:-mid}
export PYTHONUNBUFFERED=1 STEREO_HIP_TRWS_SPIN_SECONDS=5
{
for v in teddy noise; do python tools/time_trws.py 1 375 450 60 8 20 0 $v 2>&1 | grep "it/s"; STEREO_HIP_TRWS_SPEC=0 python tools/time_trws.py 1 375 450 60 8 20 0 $v 2>&1 | grep "it/s" | sed 's/^/plain: /'; done
python tools/time_trws.py 1 2000 3000 256 8 2 0 noise 2>&1 | grep "it/s"
timeout 1500 python -m pytest tests/test_spec_gpu.py tests/test_strips_gpu.py tests/test_trws_gpu.py tests/test_trws_wide_gpu.py tests/test_rd_gpu.py tests/test_stress_gpu.py -q -m gpu -x 2>&1 | tail -6
timeout 900 python -m pytest tests/test_globalstereo_gpu.py -q -m gpu -x -s 2>&1 | grep "globalstereo parity\|globalstereo free\|passed\|failed" | cut -c1-900
} > $out/${tag}_mid.txt 2>&1
cat $out/${tag}_mid.txt
