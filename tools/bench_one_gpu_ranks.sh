#!/bin/bash
# Development aid: run bench.py's N > 1 path (row strips, IPC hand-over) with all ranks on ONE GPU
# (STEREO_BENCH_ONE_GPU=1: gloo rendezvous, the strips split the CUs) on a reduced scale volume, and
# compare energy / bound / label checksum with the single-GPU run.  Not a measurement.
out=${1:-gpurun_out/one_gpu_ranks}; H=${2:-400}; W=${3:-600}
mkdir -p $out
for n in 2 4; do
  STEREO_BENCH_ONE_GPU=1 timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node $n --master-addr 127.0.0.1 \
    --master-port 29533 bench.py --gpus $n --steps 3 --warmup 1 --no-cpu-baseline --scale-height $H --scale-width $W \
    > $out/bench_n$n.json 2> $out/bench_n$n.err
  echo "n=$n rc=$?"; tail -3 $out/bench_n$n.err
done
python bench.py --no-cpu-baseline --steps 3 --warmup 1 --scale-height $H --scale-width $W > $out/bench_n1.json 2> $out/bench_n1.err
python - <<EOF2
import json
for n in (1, 2, 4):
    try:
        j = json.loads(open("$out/bench_n%d.json" % n).read().strip().splitlines()[-1]); s = j["scale"]
        print(n, "value", j["value"], "| scale", s["value"], s["energy"], s["lower_bound"], s["label_sum"], s.get("stored_per_gpu"),
              "| index order", s.get("index_order_option", {}).get("energy"))
    except Exception as e:
        print(n, "ERR", e)
EOF2
