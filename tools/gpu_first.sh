#!/bin/bash
# Development tool (MI355X box): -m gpu suite, default bench line, Teddy profile and the two timelines.
r=${1:-r06a}
out=gpurun_out; mkdir -p $out
export PYTHONUNBUFFERED=1
{ timeout 2400 python -m pytest tests -x -q -m gpu 2>&1 | tail -6; } > $out/${r}_gputests.txt 2>&1
timeout 1500 python bench.py > $out/${r}_bench_default.json 2> $out/${r}_bench_default.err
bash tools/profile_bench.sh ${r}_trws_teddy60 --steps 20 --warmup 3 > $out/${r}_profile_teddy.txt 2>&1
bash tools/gpu_timeline.sh ${r}_trws_teddy60 > /dev/null 2>&1
bash tools/gpu_timeline_wide.sh ${r}_trws_wide256_3000x2000 > /dev/null 2>&1
tail -6 $out/${r}_gputests.txt; tail -c 3000 $out/${r}_bench_default.json
