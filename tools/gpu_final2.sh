#!/bin/bash
# Development tool (MI355X box): closing run of a round after QPBO work -- the whole -m gpu suite, smoke, the QPBO
# stress tools, the rocprofv3 passes of the QPBO path and the default bench line.
#   tools/gpu_final2.sh <rNN>
r=${1:-r04}
out=gpurun_out; mkdir -p $out
export PYTHONUNBUFFERED=1
{ timeout 2400 python -m pytest tests -q -m gpu 2>&1 | tail -6; python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2; } > $out/${r}_final2_gputests.txt 2>&1
{
  timeout 400 python tools/stress_rd.py 120 512 2>&1 | tail -1
  timeout 400 python tools/stress_improve.py 120 513 2>&1 | tail -1
} > $out/${r}_final2_stress.txt 2>&1
bash tools/profile_qpbo.sh ${r}_qpbo > $out/${r}_final2_profile_qpbo.txt 2>&1
timeout 1500 python bench.py > $out/${r}_bench_default.json 2> $out/${r}_bench_default.err
cat $out/${r}_final2_gputests.txt $out/${r}_final2_stress.txt; tail -c 600 $out/${r}_bench_default.json
