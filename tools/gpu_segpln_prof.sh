#!/bin/bash
# Development tool (MI355X box): kernel trace of the 14 SegPln plane-fit maps.
repo=$(pwd); out=$repo/gpurun_out; mkdir -p $out; tag=${1:-segplnprof}
cd /tmp && export TMPDIR=/tmp; rm -rf /tmp/prof_$tag
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$tag -- python $repo/tools/time_segpln.py > $out/${tag}_run.txt 2>&1
find /tmp/prof_$tag -name '*kernel_stats.csv' -exec cp {} $out/${tag}_kernel_stats.csv \;
find /tmp/prof_$tag -name '*kernel_trace.csv' -exec cp {} $out/${tag}_kernel_trace.csv \;
head -12 $out/${tag}_kernel_stats.csv | cut -c1-220
tail -6 $out/${tag}_run.txt
python - <<'P'
import csv,sys
rows=list(csv.DictReader(open(sys.argv[1] if len(sys.argv)>1 else '/root/repo/gpurun_out/segplnprof_kernel_trace.csv')))
fit=[(int(r['End_Timestamp'])-int(r['Start_Timestamp']),r['Kernel_Name'][:60],r['Workgroup_Size_X'],r['Grid_Size_X']) for r in rows if 'fit' in r['Kernel_Name']]
fit.sort(reverse=True)
for f in fit[:12]: print("%.3f ms"%(f[0]/1e6),f[1:])
P
