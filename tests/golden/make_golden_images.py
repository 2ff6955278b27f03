#!/usr/bin/env python
"""Generates the image-pair fixtures from the DATA files the reference's examples read
(example_ncc.m:8-9, example_global.m:10-11: data/teddy/im2.png, im6.png; example_simultaneous.m:9-10:
data/baby2/im2.png, im6.png -- Middlebury stereo pairs, pixels only, no source).  Needs the build
container (/root/reference) and PIL.  Run from the repo root:

    python tests/golden/make_golden_images.py

Outputs: teddy_pair.npz (375 x 450 x 3 uint8, im0 = im2.png = left / reference view, im1 = im6.png)
and baby2_pair.npz (370 x 413 x 3 uint8), the inputs of BASELINE.json configs[0..2] and configs[4]."""
import os

import numpy as np
from PIL import Image

HERE = os.path.dirname(os.path.abspath(__file__))
REF = "/root/reference/data"

for name in ("teddy", "baby2"):
    im0 = np.asarray(Image.open(os.path.join(REF, name, "im2.png")).convert("RGB"), dtype=np.uint8)
    im1 = np.asarray(Image.open(os.path.join(REF, name, "im6.png")).convert("RGB"), dtype=np.uint8)
    assert im0.shape == im1.shape and im0.ndim == 3
    out = os.path.join(HERE, "%s_pair.npz" % name)
    np.savez_compressed(out, im0=im0, im1=im1)
    print(out, im0.shape, os.path.getsize(out), "bytes")
