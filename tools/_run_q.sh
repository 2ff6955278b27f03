bash tools/gpu_qpbo.sh r04w
echo "=== check build"
STEREO_HIP_LIB=stereo_amd/libstereo_hip_chk.so timeout 300 python tools/stress_improve.py 40 13 > gpurun_out/r04w_chk.log 2>&1; grep -c "confined check" gpurun_out/r04w_chk.log; tail -1 gpurun_out/r04w_chk.log
