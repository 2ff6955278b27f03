bash tools/gpu_qpbo.sh r04u
echo "=== no table"
STEREO_HIP_QPBO_NO_TABLE=1 timeout 600 python examples/example_global.py 2>&1 | tail -1
STEREO_HIP_QPBO_NO_TABLE=1 timeout 600 python examples/example_global.py 2>&1 | tail -1
echo "=== check build"
STEREO_HIP_LIB=stereo_amd/libstereo_hip_chk.so timeout 300 python tools/stress_improve.py 40 13 > gpurun_out/r04u_chk.log 2>&1; grep -c "confined check" gpurun_out/r04u_chk.log; tail -1 gpurun_out/r04u_chk.log
