"""-m gpu end-to-end tests on a 64x96 crop of the reference's Teddy pair
(tests/golden/teddy_crop.npz, data only): the mirrored dispmap classes on the GPU vs the
same pipeline assembled from the oracle (NumPy terms, reference QPBO where built, restated
TRW-S).  The NCC volume is checked on its own (1e-12, test_terms_gpu.py) and then shared:
everything downstream of it -- initialisation, unaries, pairwise terms, both solvers, the
scatter of the winning planes -- must agree bit for bit."""
import os

import numpy as np
import pytest

from oracle import terms as ot

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _crop():
    g = np.load(os.path.join(GOLD, "teddy_crop.npz"))
    return g["im0"].astype(np.float64), g["im1"].astype(np.float64)


class OraclePipeline:
    """dispmap_ncc assembled from the oracle pieces."""

    def __init__(self, oracle, im0, im1, disparities, kernel, unary_weight, tol, ncc=None):
        self.o, self.kernel, self.uw, self.tol = oracle, kernel, unary_weight, tol
        self.H, self.W = im0.shape[:2]
        self.d = np.asarray(disparities, np.float64)
        self.ncc = ot.compute_ncc(im0, im1, self.d) if ncc is None else np.array(ncc)
        self.i1, self.i2 = ot.construct_neighborhood(self.H, self.W)
        self.conn = np.stack([self.i1, self.i2], 1)
        self.pts = ot.get_points(self.H, self.W)
        self.w = np.ones(len(self.i1))
        best = ot.best_disp_from_ncc(self.ncc, self.d)
        self.a = np.zeros((4, self.H * self.W)); self.a[2] = 1; self.a[3] = -best.T.reshape(-1)

    def unary(self, a):
        return ot.ncc_unary_cost(self.ncc, self.d, self.uw, a, self.pts)

    def energy(self):
        E00 = ot.pairwise_cost(self.kernel, self.w,
                               ot.disparity_from_assignment(self.a[:, self.i2], self.pts[:, self.i2]),
                               ot.disparity_from_assignment(self.a[:, self.i1], self.pts[:, self.i2]), self.tol)
        return float(np.sum(self.unary(self.a)) + np.sum(E00))

    def binary_fusion(self, prop):
        E = ot.all_pairwise_costs(self.kernel, self.w, self.tol, self.a, prop, self.i1, self.i2, self.pts)
        lab, e, lb, nu = self.o.ref_rd(self.unary(self.a), self.unary(prop), *E, self.conn)
        self.a[:, lab == 1] = prop[:, lab == 1]
        return e, lb, nu

    def simultaneous_fusion(self, props, maxiter, relgap):
        props = list(props) + [self.a.copy()]
        unary = np.stack([self.unary(p) for p in props], 1)
        q, qp = ot.trws_positions(props, self.i1, self.i2, self.pts)
        L, e, lb, it = self.o.trws(self.kernel, unary, self.conn, q, qp, self.w, self.tol, maxiter, relgap, mode=1)
        a = np.zeros_like(self.a)
        for k, p in enumerate(props):
            a[:, L == k + 1] = p[:, L == k + 1]
        self.a = a
        return e, lb, it


def _proposals(N, disps):
    return [ot.fronto_parallel(d, N) for d in disps] + [
        np.stack([np.full(N, s), np.zeros(N), np.ones(N), np.full(N, -d0)]) for s, d0 in ((0.05, 8.0), (-0.04, 20.0))]


def test_ncc_binary_and_simultaneous_fusion(hip, oracle):
    if not oracle.have_ref_qpbo():
        pytest.skip("oracle/_ref/libref_qpbo.so not present")
    im0, im1 = _crop()
    disps = np.arange(0, 24.0)
    dm = hip.dispmap_ncc([im0, im1], disps, 1, 40.0, 8.0)
    assert np.max(np.abs(dm.ncc - ot.compute_ncc(im0, im1, disps))) < 1e-12
    ref = OraclePipeline(oracle, im0, im1, disps, 1, 40.0, 8.0, ncc=dm.ncc)
    assert np.array_equal(dm.assignment, ref.a)            # WTA + parabola initialisation
    assert abs(dm.energy() - ref.energy()) <= 1e-9 * abs(ref.energy())
    N = im0.shape[0] * im0.shape[1]
    props = _proposals(N, [0, 6, 12, 18])
    for P in props:                                        # example_ncc.m:44-49
        e, lb, nu = dm.binary_fusion(P)
        e_r, lb_r, nu_r = ref.binary_fusion(P)
        assert np.array_equal(dm.assignment, ref.a)
        assert nu == nu_r and abs(e - e_r) <= 1e-9 * abs(e_r) and abs(lb - lb_r) <= 1e-9 * abs(lb_r)
        assert abs(dm.energy() - e) <= 1e-9 * abs(e)      # the solver's energy is the model's energy
    single = dm.energy()
    dm.restart(); ref.__init__(oracle, im0, im1, disps, 1, 40.0, 8.0, ncc=dm.ncc)
    dm.maxiter, dm.max_relgap = 30, 1e-4
    e, lb, it = dm.simultaneous_fusion(props)              # example_ncc.m:58-60
    e_r, lb_r, it_r = ref.simultaneous_fusion(props, 30, 1e-4)
    assert it == it_r
    assert np.array_equal(dm.assignment, ref.a)
    assert abs(e - e_r) <= 1e-9 * abs(e_r) and abs(lb - lb_r) <= 1e-9 * abs(lb_r)
    assert dm.energy() <= single * 1.05                    # simultaneous fusion is at least comparable


def test_fuse_until_convergence_and_globalstereo(hip, oracle):
    im0, im1 = _crop()
    H, W = im0.shape[:2]
    N = H * W
    dm = hip.dispmap_ncc([im0, im1], np.arange(0, 24.0), 1, 40.0, 8.0)
    e0 = dm.energy()
    dm.maxiter = 12
    n = dm.binary_fuse_until_convergence(_proposals(N, [0, 8, 16]), rng=np.random.default_rng(1))
    assert n >= 2 and dm.energy() <= e0
    # globalstereo unary + segment-dependent weights (example_global.m:17-20 constants)
    P = np.zeros((3, 4, 2)); P[:, :3, 0] = np.eye(3); P[:, :3, 1] = np.eye(3); P[0, 3, 1] = -0.25
    seg = (np.arange(H)[:, None] // 16) * 10 + (np.arange(W)[None, :] // 24)
    gs = hip.dispmap_globalstereo([im0, im1], P, [0, 10], 4, segment=seg, rng=np.random.default_rng(3))
    assert (gs.d_min, gs.d_step) == (0.0, 40.0) and gs.improve
    assert set(np.unique(gs.smooth_weights)) == {18.0, 216.0}      # (9 | 108) * 2
    e_start = gs.energy()
    want = ot.globalstereo_unary_cost(im0, im1, gs.P2, 0.0, 40.0, 30.0, gs.assignment, ot.get_points(H, W))
    assert np.max(np.abs(gs.unary_cost(gs.assignment) - want)) < 1e-11
    for d in (4.0, 12.0, 24.0, 36.0):
        gs.binary_fusion(ot.fronto_parallel(d, N))
    assert gs.energy() < e_start
    with pytest.raises(hip.StereoHipError, match="wrong size"):
        gs.binary_fusion(np.zeros((4, 3)))


def test_example_ncc_on_the_teddy_pair_full_size(hip, oracle):
    """BASELINE.json configs[0] = example_ncc.m at full size on the reference's own Teddy pair
    (tests/golden/teddy_pair.npz = data/teddy/im2.png, im6.png, 375 x 450): 60 disparities (BASELINE's
    label count; the script ships 0:1:50), tol 8, unary weight 40, kernel 1 (:13-16); proposals = the
    local plane fits on the 10:50:W x 10:50:H lattice with radius 5 (:24-32, 72 of them, fitted by the
    product and handed to both sides: proposals are inputs) followed by the fronto-parallel planes
    d = 0:10:max(disparities) (:35-41); one binary_fusion per proposal in that order (:45-49).  After
    EVERY move the whole 4 x N assignment is bit-equal to the pipeline driven by the reference's QPBO
    library, num_unlabelled equal, energy and bound within 1e-9."""
    if not oracle.have_ref_qpbo():
        pytest.skip("oracle/_ref/libref_qpbo.so not present")
    g = np.load(os.path.join(GOLD, "teddy_pair.npz"))
    im0, im1 = g["im0"].astype(np.float64), g["im1"].astype(np.float64)
    H, W = im0.shape[:2]
    N = H * W
    disps = np.arange(0, 60.0)
    dm = hip.dispmap_ncc([im0, im1], disps, 1, 40.0, 8.0)
    ref = OraclePipeline(oracle, im0, im1, disps, 1, 40.0, 8.0, ncc=dm.ncc)
    assert np.array_equal(dm.assignment, ref.a)
    lattice = dm.generate_plane_lattice(radius=5, first=10, step=50)
    assert len(lattice) == 9 * 8
    props = [np.asarray(p.expand(N) if hasattr(p, "expand") else p) for p in lattice]
    props += [ot.fronto_parallel(d, N) for d in np.arange(0, disps.max() + 1e-9, 10.0)]
    assert len(props) == 78
    accepted = 0
    for k, (pl, P) in enumerate(zip(list(lattice) + props[72:], props)):
        before = ref.a.copy()
        e, lb, nu = dm.binary_fusion(pl)
        e_r, lb_r, nu_r = ref.binary_fusion(P)
        assert nu == nu_r, (k, nu, nu_r)
        assert np.array_equal(dm.assignment, ref.a), "move %d: %d pixels took another plane" % (
            k, int((dm.assignment != ref.a).any(0).sum()))
        assert abs(e - e_r) <= 1e-9 * abs(e_r) and abs(lb - lb_r) <= 1e-9 * abs(lb_r), (k, e, e_r, lb, lb_r)
        accepted += int((ref.a != before).any(0).sum())
    assert abs(dm.energy() - ref.energy()) <= 1e-9 * abs(ref.energy())
    assert accepted > N // 2          # the sweep actually rewrites the disparity map
