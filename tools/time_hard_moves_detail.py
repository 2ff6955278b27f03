#!/usr/bin/env python
"""Development tool (MI355X box): wall time of every hard move of the Teddy example next to the solver's own report."""
import ctypes, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import stereo_amd
gold = os.path.join(ROOT, "tests", "golden")
g = np.load(os.path.join(gold, "teddy_pair.npz")); sg = np.load(os.path.join(gold, "teddy_segments.npz"))
im0, im1 = g["im0"].astype(np.float64), g["im1"].astype(np.float64)
P34 = np.tile(np.hstack([np.eye(3), np.zeros((3, 1))])[:, :, None], (1, 1, 2)); P34[0, 3, 1] = -0.25
gs = stereo_amd.dispmap_globalstereo([im0, im1], P34, (0, 59), 4, segment=sg["segment"], rng=np.random.default_rng(0))
props = gs.segpln([sg["segments"][:, :, b] for b in range(14)], seed=0)
libc = ctypes.CDLL(None)
for k, pl in enumerate(props):
    libc.srand(1000 + k)
    t = time.perf_counter()
    e, lb, nu = gs.binary_fusion(pl)
    print("MOVE %2d: %.2f ms wall, unlabelled %d, energy %.4f" % (k, (time.perf_counter() - t) * 1e3, nu, e), file=sys.stderr, flush=True)
