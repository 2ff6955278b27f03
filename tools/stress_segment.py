"""Development tool: randomised parity stress of the two segmenters (stereo_segment_ms / stereo_segment_gb, f3) against
the reference's own, compiled where they lie (oracle/_ref/libref_segment*.so, where they travelled): random images of
random sizes and kinds (noise, smoothed noise, flat blocks, ramps, few colours, photographs' crops), random parameters.
Labels must be equal pixel for pixel.   python tools/stress_segment.py [seconds] [seed]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import stereo_amd
from stereo_amd import segment as S
from oracle import pyoracle as po

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
if not po.have_ref_segment() or po.ref_segment_gb_lib() is None:
    print("stress segment: oracle/_ref did not travel"); sys.exit(0)
photo = np.load(os.path.join(ROOT, "tests", "golden", "teddy_pair.npz"))["im0"]


def smooth(a, r):
    for ax in (0, 1):
        acc = np.zeros_like(a, dtype=np.float64)
        for d in range(-r, r + 1):
            acc += np.roll(a, d, axis=ax)
        a = acc / (2 * r + 1)
    return a


def image(kind, H, W):
    if kind == 0:      # noise
        im = rng.integers(0, 256, (H, W, 3))
    elif kind == 1:    # smoothed noise, stretched
        a = smooth(rng.uniform(0, 1, (H, W, 3)), int(rng.integers(1, 4)))
        im = 255 * (a - a.min()) / max(a.max() - a.min(), 1e-9)
    elif kind == 2:    # flat blocks (where the reference's scan-order shortcuts fire)
        bh, bw = int(rng.integers(1, 9)), int(rng.integers(1, 9))
        small = rng.integers(0, 256, ((H + bh - 1) // bh, (W + bw - 1) // bw, 3))
        im = np.repeat(np.repeat(small, bh, axis=0), bw, axis=1)[:H, :W]
    elif kind == 3:    # ramps + a little noise
        y, x = np.mgrid[0:H, 0:W]
        c = rng.uniform(-3, 3, (3, 2))
        im = np.stack([128 + c[k, 0] * x + c[k, 1] * y for k in range(3)], axis=2) + rng.integers(0, int(rng.integers(1, 4)), (H, W, 3))
    elif kind == 4:    # few colours
        pal = rng.integers(0, 256, (int(rng.integers(2, 6)), 3))
        im = pal[rng.integers(0, len(pal), (H, W))]
    else:              # a crop of the Teddy photograph
        y0, x0 = int(rng.integers(0, photo.shape[0] - H + 1)), int(rng.integers(0, photo.shape[1] - W + 1))
        im = photo[y0:y0 + H, x0:x0 + W]
    return np.clip(np.asarray(im, np.float64), 0, 255).astype(np.uint8)


t0, n, bad, first = time.time(), [0, 0], [0, 0], None
while time.time() - t0 < budget:
    H, W = int(rng.integers(3, 100)), int(rng.integers(3, 130))
    kind = int(rng.integers(0, 6))
    im = image(kind, H, W)
    if rng.integers(0, 2):
        h_s = int(rng.integers(1, 9)); h_r = float(rng.choice([1.0, 1.5, 2.5, 5.0, 6.5, 10.0, 20.0])); mn = int(rng.choice([0, 1, 4, 20, 100]))
        if mn >= H * W:
            continue   # (one region below minRegion covering the image: the reference reads a null neighbour list while pruning)
        got, want = S.vgg_segment_ms(im, h_s, h_r, mn), po.ref_segment_ms(im, h_s, h_r, mn)
        which, what = 0, ("ms", kind, H, W, h_s, h_r, mn)
    else:
        sigma = float(rng.choice([0, 0.3, 0.5, 0.8, 1.2, 2.0])); k = float(rng.choice([1.0, 50.0, 120.0, 300.0, 1500.0])); mn = int(rng.choice([0, 2, 10, 50]))
        compress = int(rng.integers(0, 2))
        got, want = S.vgg_segment_gb(im, sigma, k, mn, compress), po.ref_segment_gb(im, sigma, k, mn, compress)
        which, what = 1, ("gb", kind, H, W, sigma, k, mn, compress)
    n[which] += 1
    if not np.array_equal(got, want):
        bad[which] += 1
        if first is None:
            first = (what, int((np.asarray(got) != np.asarray(want)).sum()))
            np.savez(os.path.join(ROOT, "gpurun_out", "stress_segment_first.npz") if os.path.isdir(os.path.join(ROOT, "gpurun_out")) else "/tmp/stress_segment_first.npz",
                     im=im, got=got, want=want, what=np.array(str(what)))
print("stress segment: %d mean-shift + %d graph-based images, %d + %d mismatches, %.0f s%s" % (n[0], n[1], bad[0], bad[1], time.time() - t0,
                                                                                            "" if first is None else ", first: %s (%d pixels)" % first))
sys.exit(1 if bad[0] + bad[1] else 0)
