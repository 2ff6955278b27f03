"""CPU tests of the TRW-S oracle (restatement) against what pins it:

* the reference's own TRW-S type classes (oracle/_ref, where built) and the
  committed golden vectors generated from them (tests/golden/make_golden.py);
* data recorded from the real reference in SURVEY.md Appendix B (node order of
  SetAutomaticOrdering on a 6x8 grid, forward/backward list shape, the 375x450
  level statistics).
"""
import os

import numpy as np
import pytest

from helpers import grid_conn, trws_problem
from make_golden import RUNS  # tests/golden on sys.path via conftest

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")

# ranks from the reference's SetAutomaticOrdering, 6x8 grid
SURVEY_RANKS_6x8 = np.array([
    [0, 23, 22, 21, 20, 19, 18, 17],
    [1, 29, 28, 27, 26, 25, 24, 16],
    [2, 35, 34, 33, 32, 31, 30, 15],
    [3, 47, 44, 42, 40, 38, 36, 14],
    [4, 46, 45, 43, 41, 39, 37, 13],
    [5, 6, 7, 8, 9, 10, 11, 12]])


def test_ordering_matches_reference_probe(oracle):
    H, W = 6, 8
    s = oracle.trws_structure(H * W, grid_conn(H, W))
    assert np.array_equal(s["rank"].reshape(W, H).T, SURVEY_RANKS_6x8)
    nf = np.diff(s["fwd_ptr"]).reshape(W, H).T
    nb = np.diff(s["bwd_ptr"]).reshape(W, H).T
    # corners 4/0 .. 2/2, border 4/2, interior 4/4, first interior
    # column 2/6, node (H-3, 1) 0/8
    assert (nf[0, 0], nb[0, 0]) == (4, 0)
    assert (nf[H - 3, 1], nb[H - 3, 1]) == (0, 8)
    assert (nf[1, 1], nb[1, 1]) == (2, 6)
    assert (nf[2, 3], nb[2, 3]) == (4, 4)
    assert (nf[2, 0], nb[2, 0]) == (4, 2)


def test_interior_list_order(oracle):
    """forward = [down(sw), left(sw), left(unsw), down(unsw)],
    backward = [up(sw), right(sw), right(unsw), up(unsw)]."""
    H, W = 12, 14
    conn = grid_conn(H, W)
    s = oracle.trws_structure(H * W, conn)
    r, c = 4, 6
    i = c * H + r

    def desc(e):
        a, b = conn[e]
        other = b if a == i else a
        d = {(1, 0): "down", (-1, 0): "up", (0, 1): "right", (0, -1): "left"}[
            (int(other % H) - r, int(other // H) - c)]
        return d + ("(sw)" if s["dir"][e] else "(unsw)")

    fwd = [desc(e) for e in s["fwd_idx"][s["fwd_ptr"][i]:s["fwd_ptr"][i + 1]]]
    bwd = [desc(e) for e in s["bwd_idx"][s["bwd_ptr"][i]:s["bwd_ptr"][i + 1]]]
    assert fwd == ["down(sw)", "left(sw)", "left(unsw)", "down(unsw)"]
    assert bwd == ["up(sw)", "right(sw)", "right(unsw)", "up(unsw)"]


def test_golden_messages(oracle):
    """Envelope restatement == reference type classes on the committed vectors (bit exact)."""
    g = np.load(os.path.join(GOLD, "trws_messages.npz"))
    n = len(g["K"])
    assert n >= 400
    for i in range(n):
        K = int(g["K"][i])
        args = (int(g["kernel"][i]), g["Di"][i][:K], float(g["gamma"][i]), g["msg"][i][:K],
                g["q"][i][:K], g["qprim"][i][:K], float(g["alpha"][i]), float(g["lam"][i]),
                int(g["dir"][i]), int(g["mdir"][i]))
        out, vmin = oracle.update_message(*args, impl="envelope")
        assert np.array_equal(out, g["out"][i][:K]), i
        assert vmin == g["vmin"][i], i
        col = oracle.add_column(int(g["kernel"][i]), g["q"][i][:K], g["qprim"][i][:K],
                                float(g["alpha"][i]), float(g["lam"][i]), int(g["ksrc"][i]),
                                g["col_in"][i][:K], int(g["dir"][i]), int(g["mdir"][i]))
        assert np.array_equal(col, g["col_out"][i][:K]), i


def test_bruteforce_differs_only_on_ties(oracle):
    """The plain min-plus message equals the reference's envelope on tie-free vectors; on
    integer (tie-prone) vectors the reference's envelope drops tangent cones
    (typeStereoLinear.h:444-449) and may differ -- which is why the product defaults to the
    reference semantics."""
    g = np.load(os.path.join(GOLD, "trws_messages.npz"))
    diff_tiefree = 0
    for i in range(len(g["K"])):
        K = int(g["K"][i])
        ties = np.all(g["Di"][i][:K] == np.round(g["Di"][i][:K]))
        out, _ = oracle.update_message(int(g["kernel"][i]), g["Di"][i][:K], float(g["gamma"][i]),
                                       g["msg"][i][:K], g["q"][i][:K], g["qprim"][i][:K],
                                       float(g["alpha"][i]), float(g["lam"][i]), int(g["dir"][i]),
                                       int(g["mdir"][i]), impl="brute")
        if not ties and not np.array_equal(out, g["out"][i][:K]):
            diff_tiefree += 1
    assert diff_tiefree == 0


@pytest.mark.parametrize("run", RUNS, ids=[r[0] for r in RUNS])
def test_golden_runs(run, oracle):
    """Full runs: restated core + restated envelope == restated core + REFERENCE messages."""
    name, seed, H, W, K, kernel, kind, integer, tol, maxiter, relgap = run
    g = np.load(os.path.join(GOLD, "trws_runs.npz"))
    p = trws_problem(seed, H, W, K, kind=kind, integer=integer)
    lab, en, lb, it = oracle.trws(kernel, p["unary"], p["conn"], p["q"], p["qprim"], p["alphas"],
                                  tol, maxiter, relgap, mode=1)
    assert np.array_equal(lab.astype(np.int32), g[name + "_labels"])
    assert np.array_equal(np.array([en, lb, it]), g[name + "_scalars"])


def test_against_live_reference_types(oracle):
    """Where oracle/_ref exists: fresh random messages through the reference classes."""
    if not oracle.have_ref_types():
        pytest.skip("oracle/_ref/libref_trws_types.so not built (needs /root/reference)")
    rng = np.random.default_rng(7)
    for trial in range(1500):
        K = int(rng.integers(2, 66))
        kernel = 1 + trial % 2
        if trial % 3 == 0:
            q = np.arange(K, dtype=float); qp = q.copy()
        elif trial % 3 == 1:
            q = rng.integers(0, 12, K).astype(float); qp = rng.integers(0, 12, K).astype(float)
        else:
            q = rng.normal(size=K) * 10; qp = rng.normal(size=K) * 10
        Di = rng.normal(size=K) * 20; msg = rng.normal(size=K) * 5
        if trial % 5 == 0:
            Di = np.round(Di); msg = np.round(msg)
        args = (kernel, Di, 0.25, msg, q, qp, float(rng.choice([0.0, 1.0, 2.5, 18.0])),
                float(rng.choice([8.0, 0.5, 64.0])), int(rng.integers(0, 2)), int(rng.integers(0, 2)))
        a, va = oracle.update_message(*args, impl="envelope")
        b, vb = oracle.update_message(*args, impl="ref")
        assert np.array_equal(a, b) and va == vb, trial
    # Vector::ComputeMin: first minimum wins
    v = np.array([3.0, 1.0, 1.0, 2.0])
    import ctypes as C
    km = C.c_int()
    r = oracle.ref_types().ref_vec_min(C.c_int(4), v.ctypes.data_as(C.POINTER(C.c_double)), C.byref(km))
    assert (r, km.value) == (1.0, 1)


def test_unsupported_kernel_and_modes(oracle):
    p = trws_problem(3, 4, 5, 3)
    with pytest.raises(RuntimeError):
        oracle.trws(3, p["unary"], p["conn"], p["q"], p["qprim"], p["alphas"], 1.0, 2, 0.0)
    a = oracle.trws(1, p["unary"], p["conn"], p["q"], p["qprim"], p["alphas"], 1.0, 3, 0.0, mode=0)
    b = oracle.trws(1, p["unary"], p["conn"], p["q"], p["qprim"], p["alphas"], 1.0, 3, 0.0, mode=1)
    assert np.array_equal(a[0], b[0]) and a[1:] == b[1:]  # tie-free data: both message forms agree


def test_restated_envelope_equals_reference_classes_on_adversarial_messages(oracle):
    """tests/adversarial.py (near-tangent cones, ramps, vTrunc ties, near-duplicate positions): the
    oracle's envelope restatement against the reference's own type classes, bit for bit."""
    if not oracle.have_ref_types():
        pytest.skip("oracle/_ref/libref_trws_types.so not present")
    import adversarial
    n = 0
    for seed in range(16):
        kernel = 1 + seed % 2
        K = (7, 33, 60, 64)[(seed // 2) % 4]
        b = adversarial.batch(seed, kernel, K, 150)
        for m in range(150):
            args = (kernel, b["Di"][m], float(b["gamma"][m]), b["msg"][m], b["qd"][m], b["qs"][m],
                    float(b["alpha"][m]), b["lam"], 0, 0)
            a, va = oracle.update_message(*args, impl="envelope")
            r, vr = oracle.update_message(*args, impl="ref")
            assert np.array_equal(a, r) and va == vr, (seed, m)
            n += 1
    assert n == 2400
