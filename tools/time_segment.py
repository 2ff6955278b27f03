"""Development tool (GPU): the segmenters on the example pairs, stage by stage (device filter, host finish, host regions)
next to the reference's own segmenters (oracle/_ref) where they travelled.  usage: time_segment.py [teddy|baby2]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from stereo_amd import segment as S
name = sys.argv[1] if len(sys.argv) > 1 else "teddy"
im = np.load(os.path.join(ROOT, "tests", "golden", "%s_pair.npz" % name))["im0"]
H, W, _ = im.shape
S.vgg_segment_ms(im[:32, :32].copy(), 2, 3, 0)   # warm-up
def stages(h_s, h_r, mn):
    t0 = time.perf_counter(); own, ev = S.ms_own(im, h_s, h_r)
    t1 = time.perf_counter(); f, walked = S.ms_finish(im, h_s, h_r, own, ev)
    t2 = time.perf_counter(); seg = S.ms_regions(f, H, W, h_r, mn)
    t3 = time.perf_counter()
    return seg, (t1 - t0, t2 - t1, t3 - t2, walked)
seg, (a, b, c, walked) = stages(4, 5.0, 0)
print("%s %dx%d vgg_segment_ms(4, 5, 0): device filter (incl. LUV, lattice, copies) %.1f ms, host finish %.1f ms (%d pixels walked), regions %.1f ms -> %d segments" % (
    name, W, H, a * 1e3, b * 1e3, walked, c * 1e3, int(seg.max())))
t = time.perf_counter(); maps1 = S.segpln_segments(im, workers=1); dt1 = time.perf_counter() - t
for rep in range(2):
    t = time.perf_counter(); maps = S.segpln_segments(im); dt = time.perf_counter() - t
print("the 14 SegPln maps: %.1f ms one after the other, %.1f ms side by side (%d host cores; equal: %s)" % (dt1 * 1e3, dt * 1e3, os.cpu_count() or 1, np.array_equal(maps, maps1)))
for m in range(1, 8):
    _, (a, b, c, walked) = stages(m, 1.5 * m, 10 * m)
    print("   mean shift scale %d: filter %.1f ms, finish %.1f ms, regions %.1f ms" % (m, a * 1e3, b * 1e3, c * 1e3))
t = time.perf_counter(); w = S.gb_weights(im, 0); t1 = time.perf_counter(); S.gb_regions(w, H, W, 300.0, 30, 1); t2 = time.perf_counter()
print("   graph based: weights %.1f ms, sort + union-find %.1f ms" % ((t1 - t) * 1e3, (t2 - t1) * 1e3))
try:
    from oracle import pyoracle
    if pyoracle.have_ref_segment():
        t = time.perf_counter(); r = pyoracle.ref_segment_ms(im, 4, 5.0, 0); d1 = time.perf_counter() - t
        t = time.perf_counter(); rm = pyoracle.ref_segpln_segments(im); d2 = time.perf_counter() - t
        print("reference segmenters on this host: vgg_segment_ms %.0f ms (equal: %s), the 14 maps %.0f ms (equal: %s)" % (
            d1 * 1e3, np.array_equal(r, seg), d2 * 1e3, np.array_equal(rm, maps)))
except Exception as e:
    print("reference not available:", e)
