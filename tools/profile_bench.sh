#!/bin/bash
# Development tool (runs on the MI355X box): rocprofv3 passes of one bench.py configuration.
#   tools/profile_bench.sh <tag> [bench.py args...]
# Pass 1: --kernel-trace --stats (per-kernel time); passes 2/3: --pmc FETCH_SIZE / WRITE_SIZE in
# their own runs (never combined with other trace domains).  Summaries land in gpurun_out/<tag>_*;
# copy what should be judged into profiles/.
set -u
tag=$1; shift
repo=$(pwd)
out=$repo/gpurun_out
mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_$tag && mkdir -p /tmp/prof_$tag
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$tag/stats -- python "$repo/bench.py" "$@" --no-cpu-baseline > "$out/${tag}_bench_under_rocprof.json" 2> "$out/${tag}_stats.log"
find /tmp/prof_$tag/stats -name '*kernel_stats.csv' -exec cp {} "$out/${tag}_kernel_stats.csv" \;
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 900 rocprofv3 --kernel-trace --pmc $c --output-format csv -d /tmp/prof_$tag/$c -- python "$repo/bench.py" "$@" --steps 3 --warmup 1 --no-cpu-baseline > /dev/null 2> "$out/${tag}_$c.log"
  find /tmp/prof_$tag/$c -name '*counter_collection.csv' -exec cp {} "$out/${tag}_$c.csv" \;
done
python "$repo/tools/summarise_pmc.py" "$out/${tag}_FETCH_SIZE.csv" "$out/${tag}_WRITE_SIZE.csv" "$out/${tag}_bench_under_rocprof.json" > "$out/${tag}_pmc_hbm.json"
tail -1 "$out/${tag}_bench_under_rocprof.json" | cut -c1-400
head -12 "$out/${tag}_kernel_stats.csv"
cat "$out/${tag}_pmc_hbm.json" | head -50
