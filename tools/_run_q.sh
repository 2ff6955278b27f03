bash tools/gpu_qpbo.sh r04s
echo "=== check build"
for seed in 13 4135; do
STEREO_HIP_LIB=stereo_amd/libstereo_hip_chk.so timeout 300 python tools/stress_improve.py 40 $seed > gpurun_out/r04s_chk.log 2>&1; grep -c "confined check" gpurun_out/r04s_chk.log; grep "confined check" gpurun_out/r04s_chk.log | head -4; tail -1 gpurun_out/r04s_chk.log
done
STEREO_HIP_LIB=stereo_amd/libstereo_hip_chk.so timeout 600 python examples/example_global.py 2>&1 | grep -E "confined check|moves/s" | tail -3
