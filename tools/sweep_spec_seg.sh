export PYTHONUNBUFFERED=1 STEREO_HIP_TRWS_SPIN_SECONDS=5
for seg in 8 12 16 24 32 48; do
  echo "== SEG=$seg"
  STEREO_HIP_TRWS_SPEC_SEG=$seg timeout 300 python tools/time_trws.py 1 375 450 60 8 20 0 teddy 2>&1 | grep "ms/iter" | cut -c1-90
  STEREO_HIP_TRWS_SPEC_SEG=$seg timeout 400 python tools/time_trws.py 1 2000 3000 256 8 3 0 noise 2>&1 | grep "ms/iter" | cut -c1-90
done
