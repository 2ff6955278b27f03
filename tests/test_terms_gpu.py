"""-m gpu tests of the term-builder / cost-volume kernels against the NumPy restatement
(oracle/terms.py; no executable MATLAB reference exists -> "parity unpinned" for these rows).

Tolerances: plane->disparity, pairwise terms, q/qprim, NCC sampling use only + - * / in the
same association as the oracle: bit exact.  NCC volume: 1e-12 absolute on values in [-1,1]
(complex division in the oracle vs real arithmetic on the device).  globalstereo unary:
1e-12 relative (exp/log implementations differ by an ulp)."""
import numpy as np
import pytest

from oracle import terms as ot

pytestmark = pytest.mark.gpu


def _images(seed, H, W):
    rng = np.random.default_rng(seed)
    base = rng.uniform(0, 255, size=(H, W + 12, 3))
    for _ in range(2):   # smooth a little so that NCC has structure
        base = (base + np.roll(base, 1, 0) + np.roll(base, 1, 1)) / 3
    im0 = np.round(base[:, 6:6 + W])
    im1 = np.round(base[:, 9:9 + W])   # shifted by 3
    return im0, im1


def _planes(rng, N, slant=0.05, spread=8.0):
    P = np.zeros((4, N))
    P[0] = rng.normal() * slant + rng.normal(size=N) * 0.01
    P[1] = rng.normal() * slant
    P[2] = rng.choice([1.0, 2.0, -1.0], size=N)
    P[3] = -rng.uniform(0, spread, N)
    return P


@pytest.mark.parametrize("kernel", [1, 2])
def test_pairwise_terms_and_positions_bit_exact(kernel, hip):
    from stereo_amd import terms as st
    H, W = 13, 17
    N = H * W
    rng = np.random.default_rng(3)
    conn = st.construct_neighborhood(H, W)
    i1, i2 = ot.construct_neighborhood(H, W)
    assert np.array_equal(conn[0], i1) and np.array_equal(conn[1], i2)
    pts = st.get_points(H, W)
    assert np.array_equal(pts, ot.get_points(H, W))
    cur, prop = _planes(rng, N), _planes(rng, N)
    w = rng.uniform(0.5, 3, conn.shape[1])
    tol = 2.5
    for d_min, d_step in ((0.0, 0.0), (0.0, 236.0), (4.0, 100.0)):
        fn = ot.disparity_from_assignment if d_step == 0 else (
            lambda a, p: ot.globalstereo_rescale(ot.disparity_from_assignment(a, p), d_min, d_step))
        want = ot.all_pairwise_costs(kernel, w, tol, cur, prop, i1, i2, pts, disp_fn=fn)
        got = st.pairwise_terms(kernel, conn, pts, cur, prop, w, tol, d_min, d_step)
        for a, b in zip(want, got):
            assert np.array_equal(a, b)
        only00 = st.pairwise_terms(kernel, conn, pts, cur, None, w, tol, d_min, d_step)
        assert np.array_equal(only00, want[0])
        props = [_planes(rng, N) for _ in range(5)]
        q, qp = st.trws_positions(conn, pts, props, d_min, d_step)
        wq, wqp = ot.trws_positions(props, i1, i2, pts, disp_fn=fn)
        assert np.array_equal(q.T, wq) and np.array_equal(qp.T, wqp)


def test_infinite_disparity_is_an_error(hip):
    from stereo_amd import terms as st
    H, W = 4, 5
    conn = st.construct_neighborhood(H, W)
    P = np.zeros((4, H * W))     # c == 0 everywhere
    with pytest.raises(hip.StereoHipError, match="Infinite disparity"):
        st.pairwise_terms(1, conn, st.get_points(H, W), P, None, np.ones(conn.shape[1]), 1.0)


@pytest.mark.parametrize("disps", [np.arange(0, 9.0), np.array([0.0, 1.5, 2.5, 4.0, 7.25])])
def test_ncc_volume_sampler_and_wta(disps, hip):
    from stereo_amd import terms as st
    H, W = 21, 34
    im0, im1 = _images(5, H, W)
    want = ot.compute_ncc(im0, im1, disps)
    got = st.ncc_volume(im0, im1, disps)
    assert got.shape == want.shape
    assert np.max(np.abs(got - want)) < 1e-12
    assert np.max(np.abs(want)) > 0.5            # the volume is not trivially zero
    lf = st.ncc_volume(im0, im1, disps, layout=1)
    assert np.array_equal(lf, got.transpose(2, 1, 0).reshape(len(disps), H * W, order="C").reshape(len(disps), -1)
                          if False else got.reshape(H * W, len(disps), order="F").T)
    # sampler + unary on the SAME volume: bit exact
    rng = np.random.default_rng(9)
    pts = ot.get_points(H, W)
    for trial in range(3):
        P = np.zeros((4, H * W)); P[2] = 1.0
        P[0] = rng.normal() * 0.05
        P[3] = -rng.uniform(-1, disps.max() + 1, H * W)
        if trial == 0:
            P[0] = 0; P[3] = -np.round(rng.uniform(0, disps.max(), H * W))   # exactly on samples / ties
        want_u = ot.ncc_unary_cost(want, disps, 40.0, P, pts)
        got_u = st.ncc_unary(np.asfortranarray(want), disps, 40.0, P)
        assert np.array_equal(want_u, got_u)
        got_u2 = st.ncc_unary(np.asfortranarray(want.reshape(H * W, len(disps), order="F").T), disps, 40.0, P,
                              layout=1, shape=(H, W))
        assert np.array_equal(want_u, got_u2)
    wb = ot.best_disp_from_ncc(want, disps)
    gb = st.ncc_best_disp(np.asfortranarray(want), disps)
    assert np.array_equal(np.nan_to_num(wb, nan=-7.0), np.nan_to_num(gb, nan=-7.0))


def test_globalstereo_unary(hip):
    from stereo_amd import terms as st
    H, W = 19, 27
    im0, im1 = _images(7, H, W)
    P2 = np.zeros((4, 3)); P2[0, 0] = P2[1, 1] = P2[2, 2] = 1.0; P2[3, 0] = -0.25   # example_global.m:17-18
    rng = np.random.default_rng(2)
    pts = ot.get_points(H, W)
    d_min, d_step = 0.0, 40.0
    for trial in range(3):
        A = np.zeros((4, H * W)); A[2] = 1.0
        A[3] = -rng.uniform(0, 40, H * W)
        A[0] = rng.normal() * 0.1
        if trial == 2:
            A[3] = -(pts[0] - 1) * 4.0     # lands exactly on / beyond the image border: edge branches
            A[0] = 0
        want = ot.globalstereo_unary_cost(im0, im1, P2, d_min, d_step, 30.0, A, pts)
        got = st.globalstereo_unary(im0, im1, P2, d_min, d_step, 30.0, A)
        assert np.max(np.abs(got - want) / np.maximum(1e-300, np.abs(want) + 1e-3)) < 1e-12
    gray = st.globalstereo_unary(im0[:, :, 0], im1[:, :, 0], P2, d_min, d_step, 30.0, A)
    want_g = ot.globalstereo_unary_cost(im0[:, :, :1], im1[:, :, :1], P2, d_min, d_step, 30.0, A, pts)
    assert np.max(np.abs(gray - want_g)) < 1e-12


@pytest.mark.parametrize("shape", [(133, 75, 37), (61, 33, 17), (60, 32, 16), (7, 5, 3)])
def test_staged_ncc_volume_matches_numpy_and_per_tap_kernel(shape, hip, monkeypatch):
    """Integer disparities take the staged kernels (statistics pre-pass + separable cross term, several
    row bands / column segments / disparity groups, ragged at every border); the NumPy restatement
    and the per-tap kernel (forced by STEREO_HIP_NCC_NAIVE) must agree to 1e-12, both layouts."""
    from stereo_amd import terms as st
    H, W, D = shape
    im0, im1 = _images(11, H, W)
    disps = np.arange(D, dtype=np.float64)
    want = ot.compute_ncc(im0, im1, disps)
    got = st.ncc_volume(im0, im1, disps)
    lf = st.ncc_volume(im0, im1, disps, layout=1)
    monkeypatch.setenv("STEREO_HIP_NCC_NAIVE", "1")
    naive = st.ncc_volume(im0, im1, disps)
    assert np.max(np.abs(got - want)) < 1e-12
    assert np.max(np.abs(got - naive)) < 1e-12
    assert np.array_equal(lf, got.reshape(H * W, D, order="F").T)
    assert np.max(np.abs(want)) > 0.5


def test_sampler_on_unsorted_disparities_scans(hip):
    """The bisection needs strictly ascending samples; any other order keeps the reference's scan."""
    from stereo_amd import terms as st
    H, W = 9, 11
    rng = np.random.default_rng(4)
    disps = np.array([3.0, 0.0, 5.0, 1.0, 1.0, 8.0])
    ncc = rng.uniform(-1, 1, size=(H, W, len(disps)))
    pts = ot.get_points(H, W)
    P = np.zeros((4, H * W)); P[2] = 1.0
    P[3] = -rng.uniform(-1, 9, H * W)
    P[3, :20] = -np.round(rng.uniform(0, 8, 20))
    want = ot.ncc_unary_cost(ncc, disps, 40.0, P, pts)
    got = st.ncc_unary(np.asfortranarray(ncc), disps, 40.0, P)
    assert np.array_equal(want, got, equal_nan=True)   # (the duplicated sample divides by zero on both sides)
