#!/bin/bash
# Development tool (runs on the MI355X box): rocprofv3 passes of one bench.py configuration.
#   tools/profile_bench.sh <tag> [bench.py args...]
# Pass 1: --kernel-trace --stats (per-kernel time); passes 2/3: --pmc FETCH_SIZE / WRITE_SIZE in
# their own runs; passes 4-6: SQ counter groups (instruction mix, wave / wait / active cycles, LDS
# conflicts), each in its own run (never combined with trace domains other than --kernel-trace).
# Summaries land in gpurun_out/<tag>_*; copy what should be judged into profiles/.
set -u
tag=$1; shift
repo=$(pwd)
out=$repo/gpurun_out
mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_$tag && mkdir -p /tmp/prof_$tag
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$tag/stats -- python "$repo/bench.py" "$@" --no-cpu-baseline --no-scale --no-extras > "$out/${tag}_bench_under_rocprof.json" 2> "$out/${tag}_stats.log"
find /tmp/prof_$tag/stats -name '*kernel_stats.csv' -exec cp {} "$out/${tag}_kernel_stats.csv" \;
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 900 rocprofv3 --kernel-trace --pmc $c --output-format csv -d /tmp/prof_$tag/$c -- python "$repo/bench.py" "$@" --steps 3 --warmup 1 --no-cpu-baseline --no-scale --no-extras > /dev/null 2> "$out/${tag}_$c.log"
  find /tmp/prof_$tag/$c -name '*counter_collection.csv' -exec cp {} "$out/${tag}_$c.csv" \;
done
python "$repo/tools/summarise_pmc.py" "$out/${tag}_FETCH_SIZE.csv" "$out/${tag}_WRITE_SIZE.csv" "$out/${tag}_bench_under_rocprof.json" > "$out/${tag}_pmc_hbm.json"
g=0
for group in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_INSTS_SMEM" \
             "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA" \
             "SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_WAVE_CYCLES"; do
  g=$((g+1))
  timeout 900 rocprofv3 --kernel-trace --pmc $group --output-format csv -d /tmp/prof_$tag/sq$g -- python "$repo/bench.py" "$@" --steps 3 --warmup 1 --no-cpu-baseline --no-scale --no-extras > /dev/null 2> "$out/${tag}_sq$g.log"
  find /tmp/prof_$tag/sq$g -name '*counter_collection.csv' -exec cp {} "$out/${tag}_sq$g.csv" \;
done
python "$repo/tools/summarise_counters.py" "$out/${tag}_sq_counters.json" "$out/${tag}_sq1.csv" "$out/${tag}_sq2.csv" "$out/${tag}_sq3.csv" > "$out/${tag}_sq_summary.txt" 2>&1
tail -1 "$out/${tag}_bench_under_rocprof.json" | cut -c1-300
head -8 "$out/${tag}_kernel_stats.csv"
head -40 "$out/${tag}_pmc_hbm.json"
head -60 "$out/${tag}_sq_summary.txt"
