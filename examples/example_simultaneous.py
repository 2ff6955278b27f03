#!/usr/bin/env python
"""example_simultaneous.m on the GPU: a dispmap_globalstereo object with the l1 smoothness kernel,
14 piecewise-planar proposals fused (a) iteratively by binary fusion until no proposal changes the
energy and (b) at once by simultaneous fusion (TRW-S over the 14 proposals and the current
solution, maxiter 3000, max_relgap 1e-5 -- example_simultaneous.m:50-51).

    python examples/example_simultaneous.py [im_left.png im_right.png] [--size H W] [--baby2]

--baby2: the example's own input -- the Baby2 pair with the edge weights of the reference's mean-shift segmentation
and the 14 SegPln proposals on the reference's 14 segmentation maps (committed fixtures, tests/golden/baby2_pair.npz /
baby2_segments.npz; the segmenters themselves stay on the host), planes fitted on the device.
Otherwise deterministic stand-ins are used as in example_global.py, on the given images or a synthetic pair.
"""
import argparse
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "examples"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("images", nargs="*")
    ap.add_argument("--size", type=int, nargs=2, default=[370, 413])      # Baby2 at third resolution
    ap.add_argument("--baby2", action="store_true", help="example_simultaneous.m's own input from the committed fixtures")
    args = ap.parse_args()
    import stereo_amd
    from stereo_amd import terms as T
    from example_global import piecewise_planar
    sg = None
    if args.baby2:
        gold = os.path.join(ROOT, "tests", "golden")
        g = np.load(os.path.join(gold, "baby2_pair.npz")); sg = np.load(os.path.join(gold, "baby2_segments.npz"))
        images = [g["im0"].astype(np.float64), g["im1"].astype(np.float64)]
    elif len(args.images) == 2:
        from PIL import Image
        images = [np.asarray(Image.open(f).convert("RGB"), dtype=np.float64) for f in args.images]
    else:
        from bench import synthetic_pair
        images = list(synthetic_pair(args.size[0], args.size[1], 85))
    H, W = images[0].shape[:2]
    P = np.tile(np.hstack([np.eye(3), np.zeros((3, 1))])[:, :, None], (1, 1, 2))   # example_simultaneous.m:15-16
    P[0, 3, 1] = -0.25
    disp_range, disparity_factor = (0, 85), 3                                       # :17-18
    conn = T.construct_neighborhood(H, W)
    img = images[0].transpose(1, 0, 2).reshape(H * W, -1)
    same = np.abs(img[conn[0]] - img[conn[1]]).sum(axis=1) < 30.0
    weights = np.where(same, 108.0, 9.0) * 2.0
    rng = np.random.default_rng(0)
    if sg is not None:
        dm = stereo_amd.dispmap_globalstereo(images, P, disp_range, disparity_factor, segment=sg["segment"], rng=rng)
        print("start energy %.6f" % dm.energy())
        t0 = time.time()
        proposals = dm.segpln([sg["segments"][:, :, b] for b in range(14)], seed=0)         # example_simultaneous.m:30
        print("SegPln: window matching + 14 maps, %.2f s" % (time.time() - t0))
    else:
        dm = stereo_amd.dispmap_globalstereo(images, P, disp_range, disparity_factor, smooth_weights=weights, rng=rng)
        print("start energy %.6f" % dm.energy())
        d_lo, d_hi = dm.d_min, dm.d_min + dm.d_step
        proposals = [piecewise_planar(H, W, cell, rng, d_lo, d_hi) for cell in (8, 12, 16, 24, 32, 48, 64) for _ in range(2)]

    t0 = time.time()
    moves = dm.binary_fuse_until_convergence(proposals, rng=np.random.default_rng(1))
    dt = time.time() - t0
    e_iter = dm.energy()
    print("iterative binary fusion: %d moves in %.3f s (%.1f moves/s), energy %.6f" % (moves, dt, moves / dt, e_iter))

    dm.restart()
    dm.maxiter = 3000
    dm.max_relgap = 1e-5
    t0 = time.time()
    e, lb, iters = dm.simultaneous_fusion(proposals)
    dt = time.time() - t0
    print("simultaneous fusion of %d proposals: %d TRW-S iterations in %.3f s (%.1f it/s), energy %.6f, bound %.6f, "
          "relative gap %.2e" % (len(proposals), iters, dt, iters / dt, e, lb, (e - lb) / abs(e)))
    print("simultaneous / iterative energy: %.6f" % (e / e_iter))


if __name__ == "__main__":
    main()
