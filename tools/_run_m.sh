for lib in stereo_amd/libstereo_hip_base.so stereo_amd/libstereo_hip.so; do
  echo "=== $lib"
  STEREO_HIP_LIB=$lib python tools/time_moves.py 2>&1 | tail -1
  STEREO_HIP_LIB=$lib python tools/time_moves.py 2>&1 | tail -1
  STEREO_HIP_LIB=$lib python tools/time_fusion.py 2>&1 | tail -3
  STEREO_HIP_LIB=$lib python examples/example_global.py 2>&1 | tail -1
done
timeout 1200 python -m pytest tests/test_rd_gpu.py tests/test_globalstereo_gpu.py tests/test_fusion_gpu.py tests/test_pipeline_gpu.py tests/test_simultaneous_gpu.py -x -q -m gpu 2>&1 | tail -3
