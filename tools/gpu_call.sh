#!/bin/bash
# Development tool (MI355X box): runs the given test files and the default bench line without the scale leg,
# logs under gpurun_out/<tag>_*.   tools/gpu_call.sh <tag> <pytest args...>
tag=$1; shift
out=gpurun_out; mkdir -p $out
export PYTHONUNBUFFERED=1
timeout 1500 python -m pytest "$@" -q -m gpu --durations=15 > $out/${tag}_tests.txt 2>&1
timeout 900 python bench.py --no-scale > $out/${tag}_bench.json 2> $out/${tag}_bench.err
grep -E "^E  |^(FAILED|ERROR)|passed|failed|globalstereo parity" $out/${tag}_tests.txt | cut -c1-400 | head -60; tail -c 6000 $out/${tag}_bench.json; tail -5 $out/${tag}_bench.err
