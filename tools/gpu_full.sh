#!/bin/bash
# Development tool (MI355X box): the whole -m gpu suite and the default bench line (all legs), timed.
#   tools/gpu_full.sh <tag>
tag=${1:-full}; out=gpurun_out; mkdir -p $out
export PYTHONUNBUFFERED=1
t0=$(date +%s)
timeout 2400 python -m pytest tests -x -q -m gpu --durations=12 > $out/${tag}_gputests.txt 2>&1
t1=$(date +%s)
timeout 1500 python bench.py > $out/${tag}_bench_default.json 2> $out/${tag}_bench_default.err
t2=$(date +%s)
tail -22 $out/${tag}_gputests.txt; echo "pytest $((t1 - t0)) s, bench $((t2 - t1)) s"; tail -c 3000 $out/${tag}_bench_default.json; tail -3 $out/${tag}_bench_default.err
