#!/usr/bin/env python
"""example_simultaneous.m on the GPU: a dispmap_globalstereo object with the l1 smoothness kernel,
14 piecewise-planar proposals fused (a) iteratively by binary fusion until no proposal changes the
energy and (b) at once by simultaneous fusion (TRW-S over the 14 proposals and the current
solution, maxiter 3000, max_relgap 1e-5 -- example_simultaneous.m:50-51).

    python examples/example_simultaneous.py [im_left.png im_right.png] [--size H W]

As in example_global.py the SegPln proposals and the segmentation behind the edge weights are out
of scope; deterministic stand-ins are used, and a synthetic pair without images.
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
    args = ap.parse_args()
    import stereo_amd
    from stereo_amd import terms as T
    from example_global import piecewise_planar
    if len(args.images) == 2:
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
