#!/usr/bin/env python
"""Development tool (MI355X box): the bench's hard-move sequences against the reference library, move by move --
where do accepted planes differ, and are those pixels exact / near unary ties?"""
import ctypes
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "examples"))
import stereo_amd
from stereo_amd import terms as T
from oracle import pyoracle, terms as ot
from bench import synthetic_pair
from example_global import piecewise_planar

H, W = 375, 450
P34 = np.tile(np.hstack([np.eye(3), np.zeros((3, 1))])[:, :, None], (1, 1, 2)); P34[0, 3, 1] = -0.25
grng = np.random.default_rng(0)
im0, im1 = synthetic_pair(H, W, 60)
gconn = T.construct_neighborhood(H, W)
gimg = im0.transpose(1, 0, 2).reshape(H * W, -1)
same = np.abs(gimg[gconn[0]] - gimg[gconn[1]]).sum(axis=1) < 30.0
gs = stereo_amd.dispmap_globalstereo([im0, im1], P34, (0, 59), 4, smooth_weights=np.where(same, 108.0, 9.0) * 2.0, rng=grng)
props = [piecewise_planar(H, W, cell, grng, gs.d_min, gs.d_min + gs.d_step) for cell in (8, 12, 16, 24, 32, 48, 64) for _ in range(2)]
N = H * W
i1, i2 = ot.construct_neighborhood(H, W); conn = np.stack([i1, i2], 1); pts = ot.get_points(H, W)
disp = lambda a, p: ot.globalstereo_rescale(ot.disparity_from_assignment(a, p), gs.d_min, gs.d_step)
un = lambda a: T.globalstereo_unary(im0, im1, gs.P2, gs.d_min, gs.d_step, gs.options["col_thresh"], np.asfortranarray(a))
a = np.array(gs.assignment)
libc = ctypes.CDLL(None)
for k, pl in enumerate(props):
    U0, U1 = un(a), un(pl)
    E = ot.all_pairwise_costs(1, np.asarray(gs.smooth_weights), gs.tol, a, pl, i1, i2, pts, disp_fn=disp)
    strong = pyoracle.ref_rd(U0, U1, *E, conn, stage=1)[0]
    weak = pyoracle.ref_rd(U0, U1, *E, conn)[0]
    lab, e_r, lb_r, nu_r = pyoracle.ref_rd(U0, U1, *E, conn, improve=True, seed=1000 + k)
    libc.srand(1000 + k)
    glab, ge, glb, gnu = stereo_amd.rd(U0, U1, *E, conn.T + 1, dict(improve=True))
    gstrong = stereo_amd.rd(U0, U1, *E, conn.T + 1, {})[0]
    d = glab != lab
    ds = (gstrong == 1) != (weak == 1)
    out = d & (U0 != U1)
    print("move %2d: unlabelled %5d (gpu %5d) strong-unlabelled %5d | final labels differ %4d, outside exact ties %4d, max |U0-U1| there %.3g | "
          "no-improve labels (==1) differ %4d outside ties %4d | differing pixels strongly labelled by ref: %d" % (
              k, nu_r, gnu, int((strong < 0).sum()), int(d.sum()), int(out.sum()), float(np.abs(U0 - U1)[out].max()) if out.any() else 0.0,
              int(ds.sum()), int((ds & (U0 != U1)).sum()), int((d & (strong >= 0)).sum())), flush=True)
    a[:, lab == 1] = pl[:, lab == 1]
