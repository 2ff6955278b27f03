#!/bin/bash
# Development tool (runs on the MI355X box): rocprofv3 passes of the QPBO fusion path.
#   tools/profile_qpbo.sh <tag>
# (1) examples/example_global.py --teddy -- example_global.m's 14 SegPln moves with Improve on the Teddy pair (round 5;
#     the synthetic stand-in before) -- and
# (2) tools/time_moves.py -- 9 device-resident NCC plane moves -- each under --kernel-trace --stats, then
# FETCH_SIZE / WRITE_SIZE PMC passes of (1) in their own runs.  Summaries land in gpurun_out/<tag>_*.
set -u
tag=$1; repo=$(pwd); out=$repo/gpurun_out; mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp
for what in global ncc; do
  rm -rf /tmp/prof_${tag}_$what && mkdir -p /tmp/prof_${tag}_$what
  if [ $what = global ]; then cmd="python $repo/examples/example_global.py --teddy"; else cmd="python $repo/tools/time_moves.py"; fi
  timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_${tag}_$what/stats -- $cmd > "$out/${tag}_${what}_stdout.txt" 2> "$out/${tag}_${what}_stats.log"
  find /tmp/prof_${tag}_$what/stats -name '*kernel_stats.csv' -exec cp {} "$out/${tag}_${what}_kernel_stats.csv" \;
done
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 900 rocprofv3 --kernel-trace --pmc $c --output-format csv -d /tmp/prof_${tag}_global/$c -- python "$repo/examples/example_global.py" --teddy > /dev/null 2> "$out/${tag}_global_$c.log"
  find /tmp/prof_${tag}_global/$c -name '*counter_collection.csv' -exec cp {} "$out/${tag}_global_$c.csv" \;
done
python - <<PY > "$out/${tag}_global_pmc_hbm.json"
import csv, json
def per_kernel(path, counter):
    acc = {}
    for row in csv.DictReader(open(path)):
        if row.get("Counter_Name") != counter: continue
        key = row["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0][:80]
        acc.setdefault(key, {}).setdefault(row.get("Dispatch_Id"), 0.0)
        acc[key][row.get("Dispatch_Id")] += float(row["Counter_Value"])
    return {k: {"launches": len(d), "total_KB": sum(d.values()), "mean_KB": sum(d.values()) / len(d)} for k, d in acc.items()}
f, w = per_kernel("$out/${tag}_global_FETCH_SIZE.csv", "FETCH_SIZE"), per_kernel("$out/${tag}_global_WRITE_SIZE.csv", "WRITE_SIZE")
res = {"command": "rocprofv3 --kernel-trace --pmc <FETCH_SIZE|WRITE_SIZE> -- python examples/example_global.py --teddy (two separate passes)",
       "correction": "MI355X_MICROARCH.md HBM section: counters in KB; on gfx950 FETCH_SIZE is doubled for the HBM byte count",
       "per_kernel": {k: {"launches": f[k]["launches"], "fetch_KB_per_launch": f[k]["mean_KB"], "write_KB_per_launch": w.get(k, {}).get("mean_KB"),
                          "hbm_bytes_corrected_total": (2 * f[k]["total_KB"] + w.get(k, {}).get("total_KB", 0.0)) * 1024} for k in f}}
print(json.dumps(res, indent=1))
PY
head -12 "$out/${tag}_global_kernel_stats.csv"; head -8 "$out/${tag}_ncc_kernel_stats.csv"; tail -5 "$out/${tag}_global_stdout.txt"; tail -3 "$out/${tag}_ncc_stdout.txt"; head -30 "$out/${tag}_global_pmc_hbm.json"
