"""Development tool: time end-to-end binary fusion moves (dispmap_ncc.binary_fusion: pairwise terms,
two unaries, QPBO, scatter of the winning planes) on a synthetic Teddy-sized pair."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import stereo_amd
from oracle import terms as ot

H, W, D = 375, 450, 60
rng = np.random.default_rng(0)
tex = rng.uniform(0, 255, size=(H, W + D, 3))
for _ in range(3):
    tex = (tex + np.roll(tex, 1, 0) + np.roll(tex, 1, 1)) / 3
disp = (10 + 30 * (np.arange(W)[None, :] / W) + 5 * np.sin(np.arange(H)[:, None] / 40.0)).astype(int)
im0 = tex[:, :W]
im1 = np.zeros_like(im0)
cols = np.arange(W)[None, :] + disp
for c in range(3):
    im1[:, :, c] = np.take_along_axis(tex[:, :, c], np.clip(cols, 0, W + D - 1), 1)
t = time.time()
dm = stereo_amd.dispmap_ncc([im0, im1], np.arange(D, dtype=np.float64), 1, 40.0, 8.0)
print("construct (NCC volume + init) %.3f s" % (time.time() - t))
N = H * W
props = [ot.fronto_parallel(float(d), N) for d in (5, 15, 25, 35, 45)]
dm.binary_fusion(props[0])
t = time.time()
for p in props[1:]:
    e, lb, nu = dm.binary_fusion(p)
dt = (time.time() - t) / (len(props) - 1)
print("binary_fusion %.4f s per move (%.1f moves/s), energy %.3f" % (dt, 1 / dt, dm.energy()))
import cProfile, pstats
pr = cProfile.Profile(); pr.enable()
for p in props[:3]:
    dm.binary_fusion(p)
pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(14)

# simultaneous fusion (dispmap_super.m:153-198) of 7 proposals + current, 20 TRW-S iterations
dm.maxiter, dm.max_relgap = 20, 0.0
sp = [ot.fronto_parallel(float(d), N) for d in (4, 12, 20, 28, 36, 44, 52)]
dm.simultaneous_fusion(sp)
t = time.time()
e, lb, it = dm.simultaneous_fusion(sp)
dt = time.time() - t
print("simultaneous_fusion K=%d: %.3f s for %d iterations (%.1f it/s incl. unary/positions/scatter), energy %.3f lb %.3f" % (
    len(sp) + 1, dt, it, it / dt, e, lb))
