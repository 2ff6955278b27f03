"""Inputs of the reference's examples on its own image pairs, assembled on the CPU from committed data
and the oracle's NumPy restatement only (test infrastructure).

baby2_problem(): BASELINE.json configs[4] -- example_simultaneous.m on the Baby2 pair -- so that the
golden run (tests/golden/make_golden_full.py, build container) and the -m gpu test (GPU box) feed the
solver the same bits:

  images      tests/golden/baby2_pair.npz (data/baby2/im2.png, im6.png)
  weights     lambda_h / lambda_l by the reference's own mean-shift segmentation
              (tests/golden/baby2_segments.npz `segment`, dispmap_globalstereo.m:391-403)
  start       rand(sz) * d_step + d_min (:56) from a seeded NumPy generator (MATLAB's stream is
              not reproducible: the start is an input of the parity test)
  proposals   the 14 SegPln proposals (:60-201) on the reference's own 14 segmentation maps
              (`segments`), plane per segment from tests/golden/baby2_segpln_planes.npz (made by
              oracle/terms.py: window matching, LO-RANSAC, least squares; the device versions of
              both steps are checked bit for bit against it in tests/test_segpln_gpu.py)
  unary       ephoto of the colour difference (:355-375,405), QUANTISED to multiples of 2^-20:
              exp / log differ in the last bit between libm builds, SIMD paths and the device;
              on a 2^-20 grid every host computes the same doubles (and the volume gains exact
              ties between labels, which is the harder case for the message envelope)
  q, qprim    K x E positions of trws.m:33 (dispmap_super.m:177-183), disparities rescaled to
              the search range (:336-345); general planes: q != qprim.

teddy_global(): BASELINE.json configs[2] -- example_global.m on the Teddy pair: the constructor's
fields from the raw arguments, the 14 SegPln proposals, a seeded start, and a generator of the
fusion moves' QPBO inputs (U0, U1, E00 .. E11) with NumPy unaries, for the CPU-side experiments on
what the reference's labels depend on (tests/test_oracle_qpbo.py).
"""
import os

import numpy as np

from oracle import terms as ot

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def example_P():
    P = np.tile(np.hstack([np.eye(3), np.zeros((3, 1))])[:, :, None], (1, 1, 2))     # example_simultaneous.m:15-16
    P[0, 3, 1] = -0.25
    return P


def proposals_from_planes(segments, planes):
    """segments (H, W, B) ids from 1, planes: list of (S_b, 4) rows [N1 N2 1 N3] (or [0 0 1 0]) ->
    B proposals 4 x N, pixel id = col * H + row."""
    out = []
    for b in range(segments.shape[2]):
        s = segments[:, :, b].T.reshape(-1).astype(np.int64)
        out.append(np.asfortranarray(planes[b][s - 1].T))
    return out


def baby2_problem(seed=13):
    g = np.load(os.path.join(GOLD, "baby2_pair.npz"))
    sg = np.load(os.path.join(GOLD, "baby2_segments.npz"))
    pl = np.load(os.path.join(GOLD, "baby2_segpln_planes.npz"))
    im0, im1 = g["im0"].astype(np.float64), g["im1"].astype(np.float64)
    H, W = im0.shape[:2]
    P = example_P()
    su = ot.globalstereo_setup(P, [0, 85], 3, sg["segment"], 2, 1)                   # example_simultaneous.m:17-18
    rng = np.random.default_rng(seed)
    start = rng.random((H, W)) * su["d_step"] + su["d_min"]
    a = np.zeros((4, H * W)); a[2] = 1.0; a[3] = -start.T.reshape(-1)
    props = proposals_from_planes(sg["segments"], [pl["planes_%d" % b] for b in range(14)]) + [a]   # dispmap_super.m:158
    i1, i2 = ot.construct_neighborhood(H, W)
    pts = ot.get_points(H, W)
    disp = lambda asg, p: ot.globalstereo_rescale(ot.disparity_from_assignment(asg, p), su["d_min"], su["d_step"])
    unary = np.stack([ot.globalstereo_unary_cost(im0, im1, su["P2"], su["d_min"], su["d_step"], su["col_thresh"], p, pts)
                      for p in props], 1)                                              # N x K
    unary = np.round(unary * 2.0 ** 20) / 2.0 ** 20
    q, qp = ot.trws_positions(props, i1, i2, pts, disp_fn=disp)
    return dict(kernel=1, unary=np.ascontiguousarray(unary), conn=np.stack([i1, i2], 1), q=q, qprim=qp,
                alphas=np.asarray(su["weights"], np.float64), tol=su["tol"], props=props, H=H, W=W)


class TeddyGlobal:
    """example_global.m:10-31 from the raw arguments (oracle/terms.py), Teddy pair, real segmentation."""

    def __init__(self, seed=5):
        g = np.load(os.path.join(GOLD, "teddy_pair.npz"))
        sg = np.load(os.path.join(GOLD, "teddy_segments.npz"))
        pl = np.load(os.path.join(GOLD, "teddy_segpln_planes.npz"))
        self.im0, self.im1 = g["im0"].astype(np.float64), g["im1"].astype(np.float64)
        self.H, self.W = self.im0.shape[:2]
        self.N = self.H * self.W
        self.P = example_P()                                                           # example_global.m:17-18
        self.segment = sg["segment"]
        self.su = ot.globalstereo_setup(self.P, [0, 59], 4, self.segment, 2, 1)        # :19-20
        self.i1, self.i2 = ot.construct_neighborhood(self.H, self.W)
        self.conn = np.stack([self.i1, self.i2], 1)
        self.pts = ot.get_points(self.H, self.W)
        self.proposals = proposals_from_planes(sg["segments"], [pl["planes_%d" % b] for b in range(14)])
        rng = np.random.default_rng(seed)
        self.start = rng.random((self.H, self.W)) * self.su["d_step"] + self.su["d_min"]   # dispmap_globalstereo.m:56
        self.a = np.zeros((4, self.N)); self.a[2] = 1.0; self.a[3] = -self.start.T.reshape(-1)

    def disp(self, asg, p):
        return ot.globalstereo_rescale(ot.disparity_from_assignment(asg, p), self.su["d_min"], self.su["d_step"])

    def unary(self, asg):
        su = self.su
        return ot.globalstereo_unary_cost(self.im0, self.im1, su["P2"], su["d_min"], su["d_step"], su["col_thresh"], asg, self.pts)

    def move_terms(self, prop, unary=None):
        """-> U0, U1, (E00, E01, E10, E11) of fusing `prop` into the current assignment (dispmap_super.m:61-84)."""
        un = unary or self.unary
        E = ot.all_pairwise_costs(1, self.su["weights"], self.su["tol"], self.a, prop, self.i1, self.i2, self.pts, disp_fn=self.disp)
        return un(self.a), un(prop), E

    def accept(self, prop, labelling):
        m = np.asarray(labelling) == 1                                                 # dispmap_super.m:83
        self.a[:, m] = prop[:, m]
