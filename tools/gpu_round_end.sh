#!/bin/bash
# Development tool (MI355X box): what a round ends with -- the whole -m gpu suite, the randomised stress tools on
# the final code, the default bench line and the rocprofv3 passes of the two headline workloads.
#   tools/gpu_round_end.sh <rNN> [stress seconds]
r=${1:-r05}; secs=${2:-300}
out=gpurun_out; mkdir -p $out
export PYTHONUNBUFFERED=1
{ timeout 2400 python -m pytest tests -x -q -m gpu 2>&1 | tail -6; } > $out/${r}_gputests.txt 2>&1
{
  timeout $((secs + 200)) python tools/stress_trws.py $secs 511 2>&1 | tail -3
  timeout $((secs / 2 + 200)) python tools/stress_certificate.py $((secs / 2)) 51400 2>&1 | tail -2
  timeout 400 python tools/stress_rd.py 120 512 2>&1 | tail -2
  timeout 400 python tools/stress_improve.py 120 513 2>&1 | tail -2
  timeout 400 python tools/stress_segment.py 120 514 2>&1 | tail -2
  timeout 400 python tools/stress_segpln.py 120 515 2>&1 | tail -2
  timeout 400 python tools/stress_terms.py 120 516 2>&1 | tail -2
} > $out/${r}_stress.txt 2>&1
timeout 1500 python bench.py > $out/${r}_bench_default.json 2> $out/${r}_bench_default.err
bash tools/profile_bench.sh ${r}_trws_teddy60 --steps 20 --warmup 3 > $out/${r}_profile_teddy.txt 2>&1
bash tools/profile_bench.sh ${r}_trws_wide256_3000x2000 --height 2000 --width 3000 --labels 256 --steps 3 --warmup 1 > $out/${r}_profile_wide.txt 2>&1
tail -6 $out/${r}_gputests.txt; cat $out/${r}_stress.txt; tail -c 1500 $out/${r}_bench_default.json
bash tools/profile_qpbo.sh ${r}_qpbo > $out/${r}_profile_qpbo.txt 2>&1
tail -12 $out/${r}_profile_qpbo.txt | cut -c1-300
bash tools/gpu_timeline.sh ${r}_trws_teddy60 > /dev/null 2>&1
bash tools/gpu_timeline_wide.sh ${r}_trws_wide256_3000x2000 > /dev/null 2>&1
