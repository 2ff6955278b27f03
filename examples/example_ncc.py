#!/usr/bin/env python
"""example_ncc.m on the GPU: NCC cost volume, plane proposals fitted to the winner-takes-all
disparities on a lattice plus a few fronto-parallel ones, iterative binary fusion (QPBO), then --
from the initial solution again -- one simultaneous fusion of all proposals (TRW-S).

    python examples/example_ncc.py [im_left.png im_right.png] [--disparities 60]

Without images a synthetic textured pair of the Teddy size is used.  Figures are not mirrored.
"""
import argparse
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("images", nargs="*")
    ap.add_argument("--disparities", type=int, default=51)   # example_ncc.m:13 ships 0:1:50
    ap.add_argument("--maxiter", type=int, default=100)
    ap.add_argument("--host-proposals", action="store_true")
    args = ap.parse_args()
    import stereo_amd
    if len(args.images) == 2:
        from PIL import Image
        images = [np.asarray(Image.open(f).convert("RGB"), dtype=np.float64) for f in args.images]
    else:
        from bench import synthetic_pair
        images = list(synthetic_pair(375, 450, args.disparities))
    disparities = np.arange(args.disparities, dtype=np.float64)
    tol = 8 * (disparities[1] - disparities[0])
    t0 = time.time()
    dm = stereo_amd.dispmap_ncc(images, disparities, 1, 40.0, tol)
    print("NCC volume + winner-takes-all start: %.2f s, energy %.6f" % (time.time() - t0, dm.energy()))
    H, W = dm.sz
    # plane fits and proposals on the device (--host-proposals: the reference-shaped 4 x N arrays instead)
    if args.host_proposals:
        proposals = [dm.generate_new_plane_RANSAC(x, y, 5) for x in range(10, W + 1, 50) for y in range(10, H + 1, 50)]
        for d in range(0, int(disparities.max()) + 1, 10):
            p = np.zeros_like(dm.assignment)
            p[2], p[3] = 1, -d
            proposals.append(p)
    else:
        t0 = time.time()
        proposals = dm.generate_plane_lattice(radius=5)      # the whole 10:50:W x 10:50:H lattice in one launch
        print("plane lattice: %d local fits in %.1f ms (one launch)" % (len(proposals), (time.time() - t0) * 1e3))
        proposals += [stereo_amd.PlaneProposal([0.0, 0.0, 1.0, -float(d)]) for d in range(0, int(disparities.max()) + 1, 10)]
    t0 = time.time()
    for p in proposals:
        dm.binary_fusion(p)
    dt = time.time() - t0
    single = dm.energy()
    print("iterative binary fusion: %d moves in %.3f s (%.1f moves/s), energy %.6f" % (
        len(proposals), dt, len(proposals) / dt, single))
    # the same proposals revisited until none changes the energy (dispmap_super.m:85-152): one native call
    dm.restart()
    t0 = time.time()
    n = dm.binary_fuse_until_convergence(proposals, rng=np.random.default_rng(1))
    dt = time.time() - t0
    print("binary_fuse_until_convergence: %d moves in %.3f s (%.1f moves/s, schedule on the resident state), energy %.6f" % (
        n - 1, dt, (n - 1) / dt, dm.energy()))
    dm.restart()
    dm.maxiter = args.maxiter
    t0 = time.time()
    e, lb, it = dm.simultaneous_fusion(proposals)
    print("simultaneous fusion of %d proposals: %d TRW-S iterations in %.3f s, energy %.6f (bound %.6f)" % (
        len(proposals), it, time.time() - t0, dm.energy(), lb))


if __name__ == "__main__":
    main()
