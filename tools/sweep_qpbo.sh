#!/bin/bash
# Development tool (MI355X box): the QPBO schedule constants on the Teddy hard moves (tools/time_hard_moves.py teddy noref).
out=gpurun_out; mkdir -p $out; tag=${1:-sq}
run() { printf "%-60s " "$*"; env "$@" timeout 300 python tools/time_hard_moves.py teddy noref 2>/dev/null | grep -E '"moves_per_s"|"energy"' | tr -d '\n'; echo; }
{
run A=0
for v in 8 16 32 64 128; do run STEREO_HIP_QPBO_IMPROVE_BLOCKS=$v; done
if [ -n "$FULL" ]; then
for v in 8 24 32 64; do run STEREO_HIP_QPBO_TILED=$v; done
for v in 4 8 32 64; do run STEREO_HIP_QPBO_SWITCH=$v; done
for v in 2 8 16; do run STEREO_HIP_QPBO_FIRST_INTERVAL=$v; done
for v in 0 50 200 500; do run STEREO_HIP_QPBO_STALL_PERMILLE=$v; done
for v in 16 64 1024; do run STEREO_HIP_QPBO_RELABEL_EVERY=$v; done
run STEREO_HIP_QPBO_ADAPTIVE=0
fi
} > $out/${tag}_sweep_qpbo.txt 2>&1
cat $out/${tag}_sweep_qpbo.txt
