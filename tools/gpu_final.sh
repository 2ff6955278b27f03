#!/bin/bash
# Development tool (MI355X box): the whole -m gpu suite + smoke + the default bench line + the wide-kernel profile passes.
r=${1:-r04}; out=gpurun_out; mkdir -p $out
export PYTHONUNBUFFERED=1
{ timeout 2400 python -m pytest tests -q -m gpu 2>&1 | tail -8; python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2; } > $out/${r}_final_gputests.txt 2>&1
timeout 1500 python bench.py > $out/${r}_bench_default.json 2> $out/${r}_bench_default.err
bash tools/profile_bench.sh ${r}_trws_wide256_3000x2000 --height 2000 --width 3000 --labels 256 --steps 3 --warmup 1 > $out/${r}_profile_wide.txt 2>&1
tail -8 $out/${r}_final_gputests.txt; tail -c 600 $out/${r}_bench_default.json
