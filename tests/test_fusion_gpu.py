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
