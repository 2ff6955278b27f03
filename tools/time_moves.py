"""Development tool: per-move breakdown of device-resident NCC binary fusion moves with plane proposals
(STEREO_HIP_QPBO_VERBOSE / STEREO_HIP_FUSION_VERBOSE print the phases)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import stereo_amd
from bench import synthetic_pair
H, W, K = 375, 450, 60
im0, im1 = synthetic_pair(H, W, K)
dm = stereo_amd.dispmap_ncc([im0, im1], np.arange(K, dtype=np.float64), 1, 40.0, 8.0)
pps = [stereo_amd.PlaneProposal([0.0, 0.0, 1.0, -float(d)]) for d in np.linspace(2, K - 3, 9)]
dm.binary_fusion(pps[0])
t = time.perf_counter()
for p in pps[1:]:
    dm.binary_fusion(p)
print("%.3f ms per move" % ((time.perf_counter() - t) / 8 * 1e3))
