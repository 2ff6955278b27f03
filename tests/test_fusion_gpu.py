"""-m gpu tests of the device-resident fusion context (stereo_fusion_*): a move through the
context must equal the same move assembled from the stateless entry points (pairwise terms,
unaries, stereo_rd) bit for bit in the planes it accepts, and to 1e-12 relative in the energies
(the context sums on the device with a fixed tree, the host path with numpy)."""
import os

import numpy as np
import pytest

from oracle import terms as ot

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _crop():
    g = np.load(os.path.join(GOLD, "teddy_crop.npz"))
    return g["im0"].astype(np.float64), g["im1"].astype(np.float64)


def _stateless_move(hip, dm, a, prop, improve=False):
    from stereo_amd import terms as T
    E = T.pairwise_terms(dm.smoothness_kernel, dm.neighborhood, dm.points, a, prop, dm.smooth_weights, dm.tol,
                         dm.d_min, dm.d_step)
    U0, U1 = dm.unary_cost(a), dm.unary_cost(prop)
    lab, e, lb, nu = hip.rd(U0, U1, *E, dm.neighborhood + 1, {"improve": improve})
    out = a.copy(order="F")
    out[:, lab == 1] = prop[:, lab == 1]
    U = dm.unary_cost(out)
    P = T.pairwise_terms(dm.smoothness_kernel, dm.neighborhood, dm.points, out, None, dm.smooth_weights, dm.tol,
                         dm.d_min, dm.d_step)
    return out, float(np.sum(U) + np.sum(P)), e, lb, nu


@pytest.mark.parametrize("kernel", [1, 2])
def test_context_moves_equal_stateless_moves(kernel, hip):
    im0, im1 = _crop()
    disps = np.arange(0, 24.0)
    dm = hip.dispmap_ncc([im0, im1], disps, kernel, 40.0, 8.0 if kernel == 1 else 30.0)
    N = im0.shape[0] * im0.shape[1]
    a = dm.assignment.copy(order="F")
    props = [ot.fronto_parallel(d, N) for d in (3.0, 9.0, 15.0)] + [
        np.stack([np.full(N, 0.05), np.full(N, -0.02), np.ones(N), np.full(N, -8.0)])]
    for prop in props:
        a, en, e, lb, nu = _stateless_move(hip, dm, a, prop)
        e2, lb2, nu2 = dm.binary_fusion(prop)
        assert np.array_equal(dm.assignment, a)
        assert abs(dm.energy() - en) <= 1e-12 * abs(en)
        assert abs(e2 - e) <= 1e-9 * abs(e) and abs(lb2 - lb) <= 1e-9 * abs(lb) and nu2 == nu
    # the energy of a move that changes nothing is reproduced bit for bit (binary_fuse_until_convergence
    # compares energies with ~=, dispmap_super.m:137)
    e_before = dm.energy()
    dm.binary_fusion(dm.assignment.copy(order="F"))
    assert dm.energy() == e_before


def test_context_api_and_errors(hip):
    from stereo_amd.fusion import FusionContext
    from stereo_amd import terms as T
    H, W = 6, 7
    conn = T.construct_neighborhood(H, W)
    ctx = FusionContext(H, W, 1, 2.0, conn, np.ones(conn.shape[1]))
    a = ot.fronto_parallel(2.0, H * W)
    with pytest.raises(hip.StereoHipError, match="Overload unary_cost"):
        ctx.set_assignment(a)                     # no unary source yet (dispmap_super.m:200-203)
    rng = np.random.default_rng(0)
    ncc = np.asfortranarray(rng.uniform(-1, 1, size=(H, W, 5)))
    ctx.unary_ncc(ncc, np.arange(5.0), 3.0)
    with pytest.raises(hip.StereoHipError, match="no assignment"):
        ctx.binary(a)
    e = ctx.set_assignment(a)
    got, e2 = ctx.get_assignment()
    assert np.array_equal(got, a) and e2 == e
    U = T.ncc_unary(ncc, np.arange(5.0), 3.0, a)
    P = T.pairwise_terms(1, conn, T.get_points(H, W), a, None, np.ones(conn.shape[1]), 2.0)
    assert abs(e - (U.sum() + P.sum())) <= 1e-12 * abs(e)
    bad = a.copy()
    bad[2, 3] = 0
    with pytest.raises(hip.StereoHipError, match="Infinite disparity"):
        ctx.binary(bad)
    with pytest.raises(hip.StereoHipError, match="Unkown kernel type"):
        FusionContext(H, W, 3, 2.0, conn, np.ones(conn.shape[1]))


@pytest.mark.parametrize("kernel", [1, 2])
def test_context_simultaneous_equals_stateless(kernel, hip):
    """stereo_fusion_simultaneous (unary K x N, q / qprim, TRW-S and the scatter on the device) vs
    the same fusion assembled from the stateless entry points: identical planes, TRW-S energy,
    bound and iteration count; stored energy to 1e-12."""
    from stereo_amd import terms as T
    im0, im1 = _crop()
    disps = np.arange(0, 24.0)
    tol = 8.0 if kernel == 1 else 30.0
    dm = hip.dispmap_ncc([im0, im1], disps, kernel, 40.0, tol)
    dm.maxiter, dm.max_relgap = 12, 1e-6
    N = im0.shape[0] * im0.shape[1]
    props = [ot.fronto_parallel(d, N) for d in (2.0, 7.0, 12.0, 18.0)] + [
        np.stack([np.full(N, 0.05), np.full(N, -0.02), np.ones(N), np.full(N, -8.0)])]
    a0 = dm.assignment.copy(order="F")
    allp = props + [a0]
    unary = np.stack([dm.unary_cost(p) for p in allp], axis=0)
    q, qp = T.trws_positions(dm.neighborhood, dm.points, allp)
    L, e, lb, it = hip.trws(np.int32(kernel), unary, dm.neighborhood + 1, q, qp, dm.smooth_weights, tol,
                            {"maxiter": 12, "max_relgap": 1e-6})
    want = np.zeros_like(a0)
    for k, p in enumerate(allp):
        want[:, L == k + 1] = p[:, L == k + 1]
    e2, lb2, it2 = dm.simultaneous_fusion(props)
    assert np.array_equal(dm.assignment, want)
    assert e2 == e and lb2 == lb and it2 == it
    U = dm.unary_cost(want)
    P = T.pairwise_terms(kernel, dm.neighborhood, dm.points, want, None, dm.smooth_weights, tol)
    en = float(U.sum() + P.sum())
    assert abs(dm.energy() - en) <= 1e-12 * abs(en)
    # a binary move after it still works on the resident state
    dm.binary_fusion(props[0])
    assert np.isfinite(dm.energy())


def test_proposals_built_on_the_device(hip, oracle):
    """plane-table proposals (one plane for all pixels / one per segment) built in HBM give
    the very move the 4 x N array gives; the device plane fit (dispmap_ncc.m:48-92) agrees with the
    NumPy mirror of the reference's IRLS / SVD to rounding."""
    import os
    from stereo_amd import PlaneProposal
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "teddy_crop.npz"))
    im0, im1 = g["im0"].astype(np.float64), g["im1"].astype(np.float64)
    H, W = im0.shape[:2]
    N = H * W
    for kernel in (1, 2):
        a = hip.dispmap_ncc([im0, im1], np.arange(0, 24.0), kernel, 40.0, 8.0)
        b = hip.dispmap_ncc([im0, im1], np.arange(0, 24.0), kernel, 40.0, 8.0)
        # device plane fits against the host mirror
        for (x, y) in ((10, 10), (40, 30), (90, 60), (3, 2)):
            host = a.generate_new_plane_RANSAC(x, y, 5)
            dev = a.generate_new_plane_RANSAC(x, y, 5, on_device=True)
            assert isinstance(dev, PlaneProposal)
            assert np.allclose(dev.planes[:, 0], host[:, 0], rtol=1e-6, atol=1e-6), (kernel, x, y, dev.planes[:, 0], host[:, 0])
        # the same sequence of moves from plane tables and from 4 x N arrays
        seg = ((np.arange(H)[:, None] // 20) * 8 + (np.arange(W)[None, :] // 30)).T.reshape(-1)
        rng = np.random.default_rng(kernel)
        S = int(seg.max()) + 1
        props = [PlaneProposal([0.0, 0.0, 1.0, -7.0]), PlaneProposal([0.03, -0.02, 1.0, -12.0]),
                 PlaneProposal(np.stack([rng.normal(0, 0.05, S), rng.normal(0, 0.05, S), np.ones(S), -rng.uniform(0, 23, S)]), seg)]
        for P in props:
            ra = a.binary_fusion(P)
            rb = b.binary_fusion(P.expand(N))
            assert ra == rb and a.energy() == b.energy()
            assert np.array_equal(a.assignment, b.assignment)
        a.restart(); b.restart()
        a.maxiter = b.maxiter = 8
        single = props[:2] + [PlaneProposal([0.0, 0.0, 1.0, -15.0])]
        ra = a.simultaneous_fusion(single)
        rb = b.simultaneous_fusion([P.expand(N) for P in single])
        assert ra == rb and np.array_equal(a.assignment, b.assignment)


def test_plane_lattice_in_one_launch(hip):
    """the proposal lattice of example_ncc.m:24-32 -- `for x = 10:50:W, for y = 10:50:H`,
    a local plane fit of the winner-takes-all disparities at every point (dispmap_ncc.m:48-92) -- as ONE
    launch (stereo_fusion_fit_planes), against the NumPy mirror of the reference's SVD / IRLS, same order."""
    import os
    from stereo_amd import PlaneProposal
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "teddy_pair.npz"))
    im0, im1 = g["im0"][100:260, 150:330].astype(np.float64), g["im1"][100:260, 150:330].astype(np.float64)   # 160 x 180
    from oracle import terms as ot
    for kernel in (1, 2):
        dm = hip.dispmap_ncc([im0, im1], np.arange(0, 32.0), kernel, 40.0, 8.0)
        dev = dm.generate_plane_lattice(radius=5)
        host = dm.generate_plane_lattice(radius=5, on_device=False)
        # the ORACLE's restatement of dispmap_ncc.m:48-92 (oracle/terms.py: plane_lattice) on the oracle's own
        # winner-takes-all disparities (best_disp_from_ncc of the NumPy NCC volume): nothing of the product in it
        ncc_o = ot.compute_ncc(im0, im1, np.arange(0, 32.0), 2)
        want = ot.plane_lattice(ot.best_disp_from_ncc(ncc_o, np.arange(0, 32.0)), kernel, radius=5)
        assert len(dev) == len(host) == len(want) == 4 * 4    # x = 10, 60, 110, 160; y = 10, 60, 110, 160
        for k, (d, h, o) in enumerate(zip(dev, host, want)):
            assert isinstance(d, PlaneProposal)
            assert np.allclose(d.planes[:, 0], o, rtol=1e-6, atol=1e-6), (kernel, k, d.planes[:, 0], o)
            assert np.allclose(h[:, 0], o, rtol=1e-9, atol=1e-9), (kernel, k, h[:, 0], o)
        # one call, many centres == many calls, one centre (bit for bit)
        ctx = dm._context()
        one = [ctx.fit_plane(x, y, 5)[0] for x in (10, 60) for y in (10, 110)]
        many, cnt = ctx.fit_planes([10, 10, 60, 60], [10, 110, 10, 110], 5)
        assert np.array_equal(np.stack(one, 1), many) and cnt.min() > 50


class _OracleSchedule:
    """binary_fuse_until_convergence of the mirrored class (the reference's loop, stereo_amd/dispmap.py)
    driving the ORACLE's moves: dispmap_super's schedule code with OraclePipeline.binary_fusion."""

    def __init__(self, ref):
        from stereo_amd.dispmap import dispmap_super
        self.ref = ref
        self._loop = dispmap_super.binary_fuse_until_convergence
        self.maxiter = 1000
        self.sz = (ref.H, ref.W)
        self._improve = False

    def _context(self):
        return None

    def energy(self):
        return self.ref.energy()

    def binary_fusion(self, P):
        return self.ref.binary_fusion(np.asarray(P))

    def run(self, props, ids):
        return self._loop(self, props, rng=ids, device_loop=False)


def test_fuse_until_convergence_on_the_device_matches_the_loop_and_the_oracle(hip, oracle):
    """dispmap_super.m:85-152 as one native call on the resident state
    (stereo_fusion_fuse_until_convergence) == the same schedule as a Python loop of binary_fusion calls
    (energies bit for bit, planes bit for bit, 4 x N arrays and plane tables alike) == the schedule run
    over the oracle's moves (reference QPBO library): planes bit for bit, energies to 1e-9."""
    if not oracle.have_ref_qpbo():
        pytest.skip("oracle/_ref/libref_qpbo.so not present")
    import os
    from test_pipeline_gpu import OraclePipeline, _crop
    from stereo_amd import PlaneProposal
    im0, im1 = _crop()
    H, W = im0.shape[:2]
    N = H * W
    disps = np.arange(0, 24.0)
    planes = [[0.0, 0.0, 1.0, -3.0], [0.0, 0.0, 1.0, -9.0], [0.04, -0.02, 1.0, -14.0], [0.0, 0.0, 1.0, -19.0], [-0.03, 0.05, 1.0, -8.0]]
    table = [PlaneProposal(p) for p in planes]
    arrays = [t.expand(N) for t in table]
    ids = list(np.random.default_rng(5).integers(1, len(planes) + 1, 60))
    a = hip.dispmap_ncc([im0, im1], disps, 1, 40.0, 8.0)      # native schedule, 4 x N arrays
    b = hip.dispmap_ncc([im0, im1], disps, 1, 40.0, 8.0)      # native schedule, plane table
    c = hip.dispmap_ncc([im0, im1], disps, 1, 40.0, 8.0)      # Python loop
    for dm in (a, b, c):
        dm.maxiter = 40
    na = a.binary_fuse_until_convergence(arrays, rng=ids)
    nb = b.binary_fuse_until_convergence(table, rng=ids)
    nc = c.binary_fuse_until_convergence(arrays, rng=ids, device_loop=False)
    assert na == nb == nc and na >= 6
    assert a.fusion_energies == b.fusion_energies == c.fusion_energies
    assert np.array_equal(a.assignment, c.assignment) and np.array_equal(b.assignment, c.assignment)
    assert a.energy() == c.energy() == a.fusion_energies[-1]
    ref = OraclePipeline(oracle, im0, im1, disps, 1, 40.0, 8.0, ncc=a.ncc)
    sched = _OracleSchedule(ref)
    sched.maxiter = 40
    nr = sched.run(arrays, ids)
    assert nr == na
    assert np.array_equal(a.assignment, ref.a)
    assert all(abs(x - y) <= 1e-9 * abs(y) for x, y in zip(a.fusion_energies, sched.fusion_energies))
