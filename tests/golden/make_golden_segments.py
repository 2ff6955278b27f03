#!/usr/bin/env python
"""Generates the segmentation fixtures behind the globalstereo edge weights
(dispmap_globalstereo.m:391-403): the reference's OWN mean-shift segmenter (EDISON, vendored under
imrender/vgg/seg_ms, compiled from /root/reference by oracle/Makefile into oracle/_ref/
libref_segment_ms.so) run on the reference image of each committed pair with the defaults of
ojw_default_options.m:68 (seg_params = [4 5 0]).  Needs the build container.  Run from the repo root:

    python tests/golden/make_golden_segments.py

Outputs: teddy_segments.npz, baby2_segments.npz -- `segment` H x W uint32 (ids from 1, as
vgg_segment_ms returns them) and `segments` H x W x 14 uint32, the maps segpln() builds its 14
piecewise-planar proposals on (:121-134: mean shift at seven scales, then the reference's graph-based
segmenter imrender/vgg/seg_gb at seven scales); data only.  With these, configs[2] / configs[4] get the edge weights a
MATLAB user gets: lambda_h inside a segment, lambda_l across a boundary."""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
from oracle import pyoracle  # noqa: E402

for name in ("teddy", "baby2"):
    g = np.load(os.path.join(HERE, "%s_pair.npz" % name))
    seg = pyoracle.ref_segment_ms(g["im0"], 4, 5.0, 0)
    again = pyoracle.ref_segment_ms(g["im0"], 4, 5.0, 0)
    assert np.array_equal(seg, again), "the segmenter is not deterministic"
    out = os.path.join(HERE, "%s_segments.npz" % name)
    maps = pyoracle.ref_segpln_segments(g["im0"])
    assert np.array_equal(maps, pyoracle.ref_segpln_segments(g["im0"]))
    np.savez_compressed(out, segment=seg, segments=maps)
    print(out, seg.shape, "segments:", int(seg.max()), "segpln maps:", [int(maps[:, :, b].max()) for b in range(14)],
          os.path.getsize(out), "bytes")
