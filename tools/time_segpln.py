import os, sys, time, numpy as np
ROOT = "/root/repo"; sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import stereo_amd
from stereo_amd import terms as T
gold = os.path.join(ROOT, "tests", "golden")
g = np.load(os.path.join(gold, "teddy_pair.npz")); sg = np.load(os.path.join(gold, "teddy_segments.npz"))
im0, im1 = g["im0"].astype(np.float64), g["im1"].astype(np.float64)
P34 = np.tile(np.hstack([np.eye(3), np.zeros((3, 1))])[:, :, None], (1, 1, 2)); P34[0, 3, 1] = -0.25
gs = stereo_amd.dispmap_globalstereo([im0, im1], P34, (0, 59), 4, segment=sg["segment"], rng=np.random.default_rng(0))
t=time.perf_counter(); wta = gs.segpln_wta(); print("wta %.1f ms" % ((time.perf_counter()-t)*1e3))
for rep in range(2):
    tot=0
    for b in range(14):
        seg = sg["segments"][:, :, b]
        t=time.perf_counter(); out = T.segpln_planes(wta, seg, seed=b); dt=(time.perf_counter()-t)*1e3; tot+=dt
        print("map %2d: S=%d largest=%d  %.1f ms" % (b, int(seg.max()), int(np.bincount(seg.ravel().astype(np.int64))[1:].max()), dt))
    print("total %.1f ms" % tot)
maps = sg["segments"]
t = time.perf_counter()
for b in range(14):
    T.segpln_planes(wta, maps[:, :, b], seed=b, want_proposal=False)
print("planes only (no 4 x N array back to the host): total %.1f ms" % ((time.perf_counter() - t) * 1e3))
maps = np.asfortranarray(maps.astype(np.int32))   # (as dispmap_globalstereo.segpln_segments holds them: a map contiguous, column major)
for want in (True, False):
    for rep in range(3):
        t = time.perf_counter()
        T.segpln_planes_batch(wta, maps, list(range(14)), want_proposal=want)
        dt = (time.perf_counter() - t) * 1e3
    print("all 14 maps in one call (stereo_segpln_planes_batch)%s: %.1f ms" % ("" if want else ", planes only", dt))
