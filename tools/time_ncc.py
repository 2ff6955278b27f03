#!/usr/bin/env python
"""Development tool: NCC cost volume of a Teddy-sized pair, both layouts (run under
rocprofv3 --kernel-trace --stats for the kernel times)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from bench import synthetic_pair
from stereo_amd import terms as T
H, W, D = (int(a) for a in sys.argv[1:4]) if len(sys.argv) > 3 else (375, 450, 60)
im0, im1 = synthetic_pair(H, W, D)
d = np.arange(D, dtype=np.float64)
for layout in (0, 1):
    T.ncc_volume(im0, im1, d, 2, layout)
    t = time.time()
    for _ in range(3):
        v = T.ncc_volume(im0, im1, d, 2, layout)
    print("layout %d: %.1f ms per call incl. H2D / D2H of %.0f MB" % (layout, (time.time() - t) / 3 * 1e3, v.nbytes / 1e6))
