"""Development tool: randomised parity stress of the SegPln plane fits (stereo_segpln_planes / _batch) against the
oracle (oracle/terms.py:segpln_planes): random disparity maps (planes + noise + outliers, zeros included: points at
infinity), random label maps (blocks, noise, a few huge segments; label 0 = no segment; empty labels), random seeds and
inlier thresholds.  Proposals, planes and inlier counts must be equal bit for bit (NaN == NaN).
    python tools/stress_segpln.py [seconds] [seed]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import stereo_amd
from stereo_amd import terms as T
from oracle import terms as ot

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)


def labels(kind, H, W):
    if kind == 0:      # blocks
        bh, bw = int(rng.integers(1, 12)), int(rng.integers(1, 12))
        gh, gw = (H + bh - 1) // bh, (W + bw - 1) // bw
        ids = rng.permutation(gh * gw).reshape(gh, gw) + 1
        seg = np.repeat(np.repeat(ids, bh, axis=0), bw, axis=1)[:H, :W]
    elif kind == 1:    # scattered labels (segments of non-contiguous pixels)
        seg = rng.integers(1, int(rng.integers(2, 40)), (H, W))
    elif kind == 2:    # a few huge segments (the 512-thread kernel above 4096 points)
        seg = 1 + (np.add.outer(np.arange(H) * int(rng.integers(0, 3)), np.arange(W)) * int(rng.integers(1, 4)) // max(W, 1)) % int(rng.integers(1, 4))
    else:              # medium segments (384 .. 4096 points)
        seg = 1 + (np.arange(H)[:, None] // max(H // int(rng.integers(2, 6)), 1)) * 7 + (np.arange(W)[None, :] // max(W // int(rng.integers(2, 6)), 1))
    seg = np.asarray(seg, np.int64)
    if rng.integers(0, 2):
        seg[rng.uniform(size=seg.shape) < rng.uniform(0, 0.2)] = 0      # pixels of no segment
    if rng.integers(0, 3) == 0:
        seg[seg > 0] += int(rng.integers(1, 4))                          # labels nobody carries
    return seg.astype(np.int32)


def disparities(seg, H, W):
    y, x = np.mgrid[1:H + 1, 1:W + 1]
    d = np.zeros((H, W))
    for a in np.unique(seg):
        c = rng.uniform(-0.2, 0.2, 2)
        plane = 20 + c[0] * x + c[1] * y + rng.uniform(-5, 5)
        d[seg == a] = plane[seg == a]
    mode = int(rng.integers(0, 4))
    if mode == 0:
        d = np.round(d)                                   # what winner-takes-all gives: few distinct values, exact ties
    elif mode == 1:
        d = d + rng.normal(0, 0.3, d.shape)
    elif mode == 2:
        d = np.round(d + rng.normal(0, 1.0, d.shape))
    out = rng.uniform(size=d.shape) < rng.uniform(0, 0.3)
    d[out] = rng.integers(0, 60, int(out.sum()))
    if rng.integers(0, 2):
        d[rng.uniform(size=d.shape) < 0.02] = 0.0         # 1 / 0: a point at infinity (kept: WC(:,3) ~= 0)
    return np.maximum(d, 0.0)


t0, n, bad, first, kernels = time.time(), 0, 0, None, [0, 0, 0]
while time.time() - t0 < budget:
    kind = int(rng.integers(0, 4))
    H, W = (int(rng.integers(60, 130)), int(rng.integers(70, 150))) if kind >= 2 else (int(rng.integers(3, 70)), int(rng.integers(3, 70)))
    M = int(rng.integers(1, 4))
    segs = [labels(kind if m == 0 else int(rng.integers(0, 2)), H, W) for m in range(M)]
    wta = disparities(segs[0], H, W)
    seeds = [int(rng.integers(0, 1 << 40)) for _ in range(M)]
    rt = float(rng.choice([0.1, 0.02, 0.5]))
    got = T.segpln_planes_batch(wta, segs, seeds, rt=rt) if M > 1 or rng.integers(0, 2) else [T.segpln_planes(wta, segs[0], seed=seeds[0], rt=rt)]
    for m in range(M):
        want = ot.segpln_planes(wta, segs[m], seed=seeds[m], rt=rt)
        sizes = np.bincount(segs[m].ravel())[1:]
        kernels[0] += int((sizes > 4096).sum()); kernels[1] += int(((sizes > 384) & (sizes <= 4096)).sum()); kernels[2] += int(((sizes > 0) & (sizes <= 384)).sum())
        ok = (np.array_equal(got[m][0].view(np.uint64), want[0].view(np.uint64)) and np.array_equal(got[m][1].view(np.uint64), np.ascontiguousarray(want[1]).view(np.uint64))
              and np.array_equal(got[m][2], want[2]))
        n += 1
        if not ok:
            bad += 1
            if first is None:
                first = (kind, H, W, M, m, rt, seeds[m])
                np.savez(os.path.join(ROOT, "gpurun_out", "stress_segpln_first.npz") if os.path.isdir(os.path.join(ROOT, "gpurun_out")) else "/tmp/stress_segpln_first.npz",
                         wta=wta, seg=segs[m], seed=seeds[m], rt=rt)
print("stress segpln: %d maps (%d large, %d medium, %d small segments), %d mismatches, %.0f s%s" % (n, kernels[0], kernels[1], kernels[2], bad, time.time() - t0,
                                                                                                  "" if first is None else ", first: %s" % (first,)))
sys.exit(1 if bad else 0)
