#!/bin/bash
# Development tool (MI355X box): visit profile, the new full-size / oracle-side tests, the N > 1 bench path on one GPU.
out=gpurun_out; mkdir -p $out
export PYTHONUNBUFFERED=1
bash tools/gpu_prof_visit.sh r04b teddy > /dev/null 2>&1
bash tools/gpu_prof_visit.sh r04b_noise noise > /dev/null 2>&1
{
  timeout 1500 python -m pytest tests/test_globalstereo_gpu.py tests/test_fusion_gpu.py -x -q -m gpu 2>&1 | tail -15
  timeout 1500 python -m pytest "tests/test_trws_wide_gpu.py::test_wide_kernel_matches_oracle_at_750x500x256" "tests/test_trws_wide_gpu.py::test_full_size_3000x2000x256_properties" -x -q -m gpu --durations=5 2>&1 | tail -15
} > $out/r04b_tests.txt 2>&1
bash tools/bench_one_gpu_ranks.sh $out/r04b_one_gpu_ranks 400 600 > $out/r04b_one_gpu_ranks.txt 2>&1
cat $out/r04b_visit_profile.txt | grep -v '^$' | tail -60
tail -30 $out/r04b_tests.txt
tail -12 $out/r04b_one_gpu_ranks.txt
