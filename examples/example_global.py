#!/usr/bin/env python
"""example_global.m on the GPU: dispmap_globalstereo (photo-consistency unary, truncated linear
smoothness with segment-dependent weights, QPBO with Improve) and a sweep of binary fusions over
piecewise-planar proposals.

    python examples/example_global.py [im_left.png im_right.png] [--size H W] [--teddy]

--teddy: the example's own input, from the two images of the Teddy pair alone (tests/golden/teddy_pair.npz):
the object segments the reference image (mean shift, stereo_amd/segment.py: filter on the device, region
graph on the host) for its edge weights, segpln() makes the 14 segmentation maps (mean shift and the
graph-based segmenter at seven scales each) and fits a plane per segment on the device -- the maps equal
the reference's own (tests/golden/teddy_segments.npz, checked here when the file is present).  Otherwise
the script uses deterministic stand-ins -- block-wise random planes at several block sizes, and "same
segment" = small colour difference across the edge -- on the given images or, without images, on a
synthetic textured pair.
"""
import argparse
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def piecewise_planar(H, W, cell, rng, d_lo, d_hi):
    """One proposal, 4 x N (pixel id = col * H + row): a random plane [a b 1 -d] per cell x cell block."""
    rows, cols = np.meshgrid(np.arange(H), np.arange(W), indexing="ij")
    by, bx = rows // cell, cols // cell
    nb = (by.max() + 1, bx.max() + 1)
    a = rng.normal(0, 0.3, nb)[by, bx]
    b = rng.normal(0, 0.3, nb)[by, bx]
    d0 = rng.uniform(d_lo, d_hi, nb)[by, bx]
    x, y = cols + 1.0, rows + 1.0
    cx, cy = (bx + 0.5) * cell, (by + 0.5) * cell
    # disparity d0 at the block centre: -(a x + b y + p4) = d0 + a (cx - x) + b (cy - y)
    p4 = -(d0 + a * cx + b * cy)
    planes = np.stack([a, b, np.ones_like(a), p4])                     # 4 x H x W
    return np.asfortranarray(planes.transpose(0, 2, 1).reshape(4, H * W))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("images", nargs="*")
    ap.add_argument("--size", type=int, nargs=2, default=[375, 450])
    ap.add_argument("--segpln", action="store_true",
                    help="proposals = the class's SegPln proposals (dispmap_globalstereo.m:60-201: window matching, LO-RANSAC "
                         "plane per segment, on the device) over 14 colour / block segmentations standing in for the "
                         "mean-shift / Felzenszwalb maps of :122-137, instead of the random piecewise-planar stand-ins")
    ap.add_argument("--teddy", action="store_true", help="example_global.m's own input from the committed fixtures")
    args = ap.parse_args()
    import stereo_amd
    from stereo_amd import terms as T
    if args.teddy:
        gold = os.path.join(ROOT, "tests", "golden")
        g = np.load(os.path.join(gold, "teddy_pair.npz"))
        images = [g["im0"].astype(np.float64), g["im1"].astype(np.float64)]
        P = np.tile(np.hstack([np.eye(3), np.zeros((3, 1))])[:, :, None], (1, 1, 2)); P[0, 3, 1] = -0.25
        t0 = time.time()
        dm = stereo_amd.dispmap_globalstereo(images, P, (0, 59), 4, rng=np.random.default_rng(0))
        print("object (mean-shift segmentation: %d segments) + random start: %.2f s, energy %.6f" % (
            int(dm.segment.max()), time.time() - t0, dm.energy()))
        t1 = time.time()
        maps = dm.segpln_segments()
        t2 = time.time()
        proposals = dm.segpln(seed=0)
        print("SegPln: 14 segmentation maps %.2f s (%s segments), window matching + plane fits %.2f s" % (
            t2 - t1, [int(maps[:, :, b].max()) for b in range(14)], time.time() - t2))
        ref = os.path.join(gold, "teddy_segments.npz")
        if os.path.exists(ref):
            sg = np.load(ref)
            print("maps equal to the reference's own segmenters' (tests/golden/teddy_segments.npz): edge-weight map %s, the 14 maps %s" % (
                np.array_equal(dm.segment, sg["segment"]), np.array_equal(maps, sg["segments"])))
        t0 = time.time()
        for p in proposals:
            e, lb, unl = dm.binary_fusion(p)
        dt = time.time() - t0
        print("binary fusion of %d SegPln proposals: %.3f s (%.1f moves/s), energy %.6f, last move: %d unlabelled" % (
            len(proposals), dt, len(proposals) / dt, dm.energy(), int(unl)))
        return
    if len(args.images) == 2:
        from PIL import Image
        images = [np.asarray(Image.open(f).convert("RGB"), dtype=np.float64) for f in args.images]
    else:
        from bench import synthetic_pair
        images = list(synthetic_pair(args.size[0], args.size[1], 60))
    H, W = images[0].shape[:2]
    P = np.tile(np.hstack([np.eye(3), np.zeros((3, 1))])[:, :, None], (1, 1, 2))   # example_global.m:17-18
    P[0, 3, 1] = -0.25
    disp_range, disparity_factor = (0, 59), 4
    # edge weights: lambda_h inside a "segment" (small colour difference), lambda_l across
    conn = T.construct_neighborhood(H, W)
    img = images[0].transpose(1, 0, 2).reshape(H * W, -1)
    same = np.abs(img[conn[0]] - img[conn[1]]).sum(axis=1) < 30.0
    weights = np.where(same, 108.0, 9.0) * 2.0                                      # dispmap_globalstereo.m:400-403
    rng = np.random.default_rng(0)
    t0 = time.time()
    dm = stereo_amd.dispmap_globalstereo(images, P, disp_range, disparity_factor, smooth_weights=weights, rng=rng)
    print("object + random start: %.2f s, energy %.6f" % (time.time() - t0, dm.energy()))
    d_lo, d_hi = dm.d_min, dm.d_min + dm.d_step
    if args.segpln:
        # 14 segmentation maps at the scales of `mults` (:122): blocks of growing size, split by coarse colour
        t1 = time.time()
        maps = []
        for m in (1, 2, 3, 4, 5, 6, 7, 3, 5, 8, 12, 24, 50, 100):
            cell = 4 + 2 * min(m, 30)
            q = (images[0] // (256 // max(2, 8 - min(m, 6)))).astype(np.int64)
            lab = ((np.arange(H)[:, None] // cell) * 4096 + (np.arange(W)[None, :] // cell)) * 512 + q[:, :, 0] * 64 + q[:, :, 1] * 8 + q[:, :, 2]
            maps.append(np.unique(lab, return_inverse=True)[1].reshape(H, W) + 1)
        proposals = dm.segpln(maps, seed=0)
        print("SegPln: window matching + %d maps (%d .. %d segments), %.2f s" % (
            len(maps), min(int(m.max()) for m in maps), max(int(m.max()) for m in maps), time.time() - t1))
    else:
        proposals = [piecewise_planar(H, W, cell, rng, d_lo, d_hi) for cell in (8, 12, 16, 24, 32, 48, 64) for _ in range(2)]
    t0 = time.time()
    for k, p in enumerate(proposals):
        e, lb, unl = dm.binary_fusion(p)
    dt = time.time() - t0
    print("binary fusion of %d piecewise-planar proposals: %.3f s (%.1f moves/s), energy %.6f, last move: %d unlabelled" % (
        len(proposals), dt, len(proposals) / dt, dm.energy(), int(unl)))


if __name__ == "__main__":
    main()
