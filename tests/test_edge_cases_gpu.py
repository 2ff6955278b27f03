"""-m gpu edge cases the reference accepts (or rejects) at the solver boundary: empty edge
sets, single nodes, one label, all-zero weights, duplicated proposals, parallel and
antiparallel multi-edges, K just below / above the wave width."""
import numpy as np
import pytest

from helpers import grid_conn, trws_problem

pytestmark = pytest.mark.gpu


def _cmp_trws(hip, oracle, kernel, unary, conn, q, qp, al, tol, maxiter=4, relgap=0.0):
    ref = oracle.trws(kernel, unary, conn, q, qp, al, tol, maxiter, relgap, mode=1)
    got = hip.trws(kernel, unary.T, conn.T + 1, q.T, qp.T, al, tol, dict(maxiter=maxiter, max_relgap=relgap))
    assert np.array_equal(got[0], ref[0])
    assert got[1:] == ref[1:]
    return got


def test_trws_single_edge_and_two_nodes(hip, oracle):
    rng = np.random.default_rng(0)
    conn = np.array([[0, 1]])
    _cmp_trws(hip, oracle, 1, rng.uniform(0, 5, (2, 4)), conn, rng.normal(size=(1, 4)), rng.normal(size=(1, 4)),
              np.array([1.5]), 2.0)


def test_trws_one_label(hip, oracle):
    p = trws_problem(31, 4, 5, 1)
    lab, en, lb, it = _cmp_trws(hip, oracle, 1, p["unary"], p["conn"], p["q"], p["qprim"], p["alphas"], 1.0)
    assert np.all(lab == 1)


def test_trws_zero_weights_decouples_nodes(hip, oracle):
    p = trws_problem(32, 6, 7, 5)
    al = np.zeros_like(p["alphas"])
    lab, en, lb, it = _cmp_trws(hip, oracle, 2, p["unary"], p["conn"], p["q"], p["qprim"], al, 1.0, maxiter=3)
    assert np.array_equal(lab, np.argmin(p["unary"], 1) + 1)
    assert abs(en - p["unary"].min(1).sum()) < 1e-9


def test_trws_duplicated_proposals(hip, oracle):
    """simultaneous_fusion appends the current assignment, which usually equals one of the
    proposals at most pixels: identical labels (equal positions and equal unaries)."""
    p = trws_problem(33, 8, 9, 6)
    unary = np.concatenate([p["unary"], p["unary"][:, :2]], 1)
    q = np.concatenate([p["q"], p["q"][:, :2]], 1)
    qp = np.concatenate([p["qprim"], p["qprim"][:, :2]], 1)
    lab, _, _, _ = _cmp_trws(hip, oracle, 1, unary, p["conn"], q, qp, p["alphas"], 2.0, maxiter=6)
    assert lab.max() <= 6      # ties resolve to the FIRST minimum (typeStereoLinear.h:242-249)


@pytest.mark.parametrize("K", [63, 64, 65])
def test_trws_around_wave_width(K, hip, oracle):
    p = trws_problem(40 + K, 5, 6, K, kind="fronto")
    _cmp_trws(hip, oracle, 1, p["unary"], p["conn"], p["q"], p["qprim"], p["alphas"], 4.0, maxiter=3)


def test_trws_irregular_graph(hip, oracle):
    """Not a grid: random multigraph (parallel edges, high degrees) -> generic kernels."""
    rng = np.random.default_rng(5)
    N, K = 30, 5
    a = rng.integers(0, N, 90); b = rng.integers(0, N, 90)
    m = a != b
    conn = np.stack([a[m], b[m]], 1)
    E = conn.shape[0]
    _cmp_trws(hip, oracle, 1, rng.uniform(0, 9, (N, K)), conn, rng.normal(size=(E, K)) * 3, rng.normal(size=(E, K)) * 3,
              rng.uniform(0.5, 2, E), 2.0, maxiter=5)


def test_trws_negative_energy_stops_at_once(hip, oracle):
    """rel_gap divides by the energy (minimize.cpp:105): with a negative energy the test fires
    in iteration 1 -- replicated, not fixed."""
    p = trws_problem(34, 5, 5, 4)
    unary = p["unary"] - 100.0
    got = _cmp_trws(hip, oracle, 1, unary, p["conn"], p["q"], p["qprim"], p["alphas"], 2.0, maxiter=50, relgap=1e-4)
    assert got[3] == 1


def test_rd_without_edges_and_with_multi_edges(hip, oracle):
    rng = np.random.default_rng(1)
    N = 7
    U0, U1 = rng.uniform(0, 3, N), rng.uniform(0, 3, N)
    z = np.zeros(0)
    lab, en, lb, nu = hip.rd(U0, U1, z, z, z, z, np.zeros((2, 0), np.int64), {})
    assert np.array_equal(lab, (U1 < U0).astype(float)) and nu == 0
    assert abs(en - np.minimum(U0, U1).sum()) < 1e-12 and abs(lb - en) < 1e-12
    # three terms on the same pair, both directions
    conn = np.array([[0, 1], [1, 0], [0, 1], [2, 3]])
    E = conn.shape[0]
    T = [rng.uniform(0, 4, E) for _ in range(4)]
    got = hip.rd(U0, U1, *T, conn.T + 1, {})
    if oracle.have_ref_qpbo():
        ref = oracle.ref_rd(U0, U1, *T, conn)
    else:
        ref = oracle.rd(U0, U1, *T, conn)
    assert np.array_equal(got[0], ref[0])
    assert abs(got[1] - ref[1]) < 1e-9 and abs(got[2] - ref[2]) < 1e-9


def test_argument_errors(hip):
    p = trws_problem(35, 3, 3, 3)
    bad = p["conn"].T + 1
    bad = bad.copy(); bad[0, 0] = 99
    with pytest.raises(hip.StereoHipError, match="out of range"):
        hip.trws(1, p["unary"].T, bad, p["q"].T, p["qprim"].T, p["alphas"], 1.0, {})
    with pytest.raises(hip.StereoHipError, match="K must be in"):
        hip.trws(1, np.zeros((600, 2)), np.array([[1], [2]]), np.zeros((600, 1)), np.zeros((600, 1)), np.ones(1), 1.0, {})
