#!/usr/bin/env python
"""Plane per segment of the 14 SegPln proposals (dispmap_globalstereo.m:60-201) on the reference's
own segmentation maps (tests/golden/*_segments.npz `segments`), computed by the oracle's NumPy
restatement (oracle/terms.py: segpln_wta, segpln_planes -- window matching, LO-RANSAC, least
squares; seed b for map b).  Run from the repo root:

    python tests/golden/make_golden_segpln.py

Outputs baby2_segpln_planes.npz / teddy_segpln_planes.npz: `planes_b` (S_b, 4) rows [N1 N2 1 N3]
([0 0 1 0] where no plane was fitted, :157-161) and `wta` (H, W), data only."""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from oracle import terms as ot  # noqa: E402
from example_inputs import example_P  # noqa: E402

for name, disp_range, factor in (("baby2", [0, 85], 3), ("teddy", [0, 59], 4)):
    g = np.load(os.path.join(HERE, "%s_pair.npz" % name))
    sg = np.load(os.path.join(HERE, "%s_segments.npz" % name))
    ims = [g["im0"].astype(np.float64), g["im1"].astype(np.float64)]
    lo, hi = disp_range[0] * factor, disp_range[1] * factor
    disps = (lo + np.arange(np.floor(hi - lo + 1e-10) + 1))[::-1]                    # :48-49
    wta = ot.segpln_wta(ims, example_P(), disps)
    out = {"wta": wta}
    for b in range(14):
        seg = sg["segments"][:, :, b]
        prop, planes, ninl = ot.segpln_planes(wta, seg, seed=b)
        # plane row per segment as the proposal holds it (after the NaN / Inf rule of :193-196)
        s = seg.T.reshape(-1).astype(np.int64)
        rows = np.zeros((int(s.max()), 4)); rows[:, 2] = 1.0
        first = np.unique(s, return_index=True)
        rows[first[0] - 1] = prop[:, first[1]].T
        assert np.array_equal(rows[s - 1].T, prop)
        out["planes_%d" % b] = rows
    path = os.path.join(HERE, "%s_segpln_planes.npz" % name)
    np.savez_compressed(path, **out)
    print(path, os.path.getsize(path), "bytes")
