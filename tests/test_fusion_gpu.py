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
