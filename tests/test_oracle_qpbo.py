"""CPU tests of the restated roof-duality oracle (oracle/qpbo_oracle.c) against the
REFERENCE QPBO library (oracle/_ref, where built) and the committed golden vectors that
were generated from it (tests/golden/rd_runs.npz)."""
import os

import numpy as np
import pytest

from helpers import fusion_problem, glass_problem
from make_golden_rd import RD_RUNS, make_problem

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _rel(a, b):
    return abs(a - b) / max(1.0, abs(a), abs(b))


def _args(p):
    return (p["U0"], p["U1"], p["E00"], p["E01"], p["E10"], p["E11"], p["conn"])


@pytest.mark.parametrize("run", RD_RUNS, ids=[r[0] for r in RD_RUNS])
def test_golden_rd_runs(run, oracle):
    name, kind, seed, H, W, params = run
    g = np.load(os.path.join(GOLD, "rd_runs.npz"))
    p = make_problem(kind, seed, H, W, params)
    strong, _, _, _ = oracle.rd(*_args(p), stage=1)
    assert np.array_equal(strong.astype(np.int8), g[name + "_strong"])
    weak, en, lb, nu = oracle.rd(*_args(p))
    en_r, lb_r, nu_r, en_i = g[name + "_scalars"]
    det = g[name + "_strong"] >= 0
    assert np.array_equal(weak[det].astype(np.int8), g[name + "_weak"][det])
    if np.array_equal(weak.astype(np.int8), g[name + "_weak"]):
        assert nu == nu_r and _rel(en, en_r) < 1e-9
    assert _rel(lb, lb_r) < 1e-9
    imp, en2, _, _ = oracle.rd(*_args(p), improve=True, seed=seed)
    assert np.array_equal(imp.astype(np.int8), g[name + "_improved"])
    assert _rel(en2, en_i) < 1e-9


def test_against_live_reference(oracle):
    if not oracle.have_ref_qpbo():
        pytest.skip("oracle/_ref/libref_qpbo.so not built (needs /root/reference)")
    probs = [fusion_problem(s, 12, 14, kernel=1 + s % 2, tol=8.0, nonsub_boost=3.0 * (s % 3)) for s in range(8)]
    probs += [fusion_problem(20 + s, 20, 25, integer=True) for s in range(3)]
    probs += [glass_problem(s, 8, 9, 3.0, s % 2 == 0) for s in range(8)]
    for k, p in enumerate(probs):
        a = _args(p)
        rs, os_ = oracle.ref_rd(*a, stage=1), oracle.rd(*a, stage=1)
        assert np.array_equal(rs[0], os_[0]), k                   # strong persistency: flow invariant
        rw, ow = oracle.ref_rd(*a), oracle.rd(*a)
        det = rs[0] >= 0
        assert np.array_equal(rw[0][det], ow[0][det]), k
        assert np.all(ow[0][rw[0] < 0] < 0), k
        assert _rel(rw[2], ow[2]) < 1e-9, k
        if np.array_equal(rw[0], ow[0]):
            assert rw[3] == ow[3] and _rel(rw[1], ow[1]) < 1e-9, k
        ri, oi = oracle.ref_rd(*a, improve=True, seed=k), oracle.rd(*a, improve=True, seed=k)
        assert np.array_equal(ri[0], oi[0]), k
        assert _rel(ri[1], oi[1]) < 1e-9, k


def test_energy_is_the_model_energy_and_bounds_hold(oracle):
    p = fusion_problem(77, 15, 18, nonsub_boost=4.0)
    lab, en, lb, nu = oracle.rd(*_args(p))
    x = (lab == 1)
    e = np.where(x, p["U1"], p["U0"]).sum()
    xi, xj = x[p["conn"][:, 0]], x[p["conn"][:, 1]]
    e += np.where(xi, np.where(xj, p["E11"], p["E10"]), np.where(xj, p["E01"], p["E00"])).sum()
    assert abs(e - en) <= 1e-9 * abs(e)
    assert lb <= en + 1e-9 * abs(en)
    # tiny exhaustive check: the bound is below the true optimum, strong labels are optimal
    q = glass_problem(5, 3, 3, 3.0, True)
    lab, en, lb, nu = oracle.rd(*_args(q), stage=1)
    N = 9
    best, arg = np.inf, None
    for m in range(1 << N):
        y = np.array([(m >> i) & 1 for i in range(N)], bool)
        v = np.where(y, q["U1"], q["U0"]).sum()
        a, b = y[q["conn"][:, 0]], y[q["conn"][:, 1]]
        v += np.where(a, np.where(b, q["E11"], q["E10"]), np.where(b, q["E01"], q["E00"])).sum()
        if v < best - 1e-12:
            best, arg = v, [y]
        elif abs(v - best) <= 1e-12:
            arg.append(y)
    assert lb <= best + 1e-9
    for i in range(N):
        if lab[i] >= 0:      # strong persistency: every optimum agrees with the label
            assert all(bool(y[i]) == bool(lab[i]) for y in arg)


NEAR_TIE = 1e-14      # |U0 - U1| at or below this: both planes' photo-consistency saturates at log 2 (to the last bits)


def _reformulations(E):
    """Edge orders under which the energy function is the same function: reversed, and seeded shuffles."""
    yield "reversed", np.arange(E)[::-1]
    yield "shuffled", np.random.default_rng(100).permutation(E)


def test_tie_labels_of_the_reference_depend_on_its_own_edge_order(oracle):
    """SURVEY a18 / VERDICT r4 'close a18 with evidence', the CPU half.  dispmap_globalstereo on the Teddy
    pair (example_global.m's constants, real mean-shift edge weights) fusing block-wise random planes --
    the moves of tests/test_globalstereo_gpu.py, where ~23 % of the pixels have EXACTLY equal unaries
    for both planes (both leave the right image, the photo-consistency saturates,
    dispmap_globalstereo.m:371,405; the example's own SegPln proposals follow the scene and have 5-14 %
    such pixels and an order of magnitude fewer order-dependent labels).  Established here, inside the
    reference's own QPBO library:
      (1) handing it the SAME energy function with the edges in another order changes its strong labels
          at dozens of pixels per move -- and only at pixels with |U0 - U1| <= 1e-14;
      (2) on the same moves with every input rounded to a 2^-30 grid (a change of < 5e-10 per term; all
          capacities, sums and augmentations are then exact in double precision) the labels do NOT depend
          on the edge order any more, although there are even more exact ties -- so (1) is rounding in the
          library's capacities and flows (QPBO.h:127-128 warns about it), not a property of the energy;
      (3) on that grid this repo's restatement (oracle/qpbo_oracle.c: another construction order, another
          max-flow) gives the reference's labels at EVERY pixel, Improve included.
    The -m gpu half (tests/test_globalstereo_gpu.py) shows (3) for the HIP path and that on the
    unrounded inputs its differences from the reference lie in the same near-tie set."""
    if not oracle.have_ref_qpbo():
        pytest.skip("oracle/_ref/libref_qpbo.so not built (needs /root/reference)")
    from example_inputs import TeddyGlobal
    from helpers import piecewise_planar
    T = TeddyGlobal(seed=5)
    rng = np.random.default_rng(5)
    d_lo, d_hi = T.su["d_min"], T.su["d_min"] + T.su["d_step"]
    Q = 2.0 ** 30
    grid = lambda x: np.round(np.asarray(x) * Q) / Q
    flipped_total = 0
    for k, cell in enumerate((8, 12, 16)):
        prop = piecewise_planar(T.H, T.W, cell, rng, d_lo, d_hi)
        U0, U1, E = T.move_terms(prop)
        ne = T.conn.shape[0]
        base = oracle.ref_rd(U0, U1, *E, T.conn, stage=1)[0]
        near = np.abs(U0 - U1) <= NEAR_TIE
        assert (U0 == U1).sum() > 0.15 * T.N                       # the exact ties this is about
        U0g, U1g, Eg = grid(U0), grid(U1), [grid(x) for x in E]
        base_g = oracle.ref_rd(U0g, U1g, *Eg, T.conn, stage=1)[0]
        F = np.zeros(T.N, bool)
        for name, pr in _reformulations(ne):
            r = oracle.ref_rd(U0, U1, *[x[pr].copy() for x in E], T.conn[pr].copy(), stage=1)[0]
            d = r != base
            assert not (d & ~near).any(), "move %d, %s: the reference's labels moved outside the near ties" % (k, name)
            F |= d
            rg = oracle.ref_rd(U0g, U1g, *[x[pr].copy() for x in Eg], T.conn[pr].copy(), stage=1)[0]
            assert np.array_equal(rg, base_g), "move %d, %s: order dependence on the exact grid" % (k, name)   # (2)
        flipped_total += int(F.sum())
        if k > 0:
            assert F.any(), "move %d: no pixel of the reference depends on its edge order" % k                # (1)
        # (3) restatement == reference on the grid, strong labels and the full gateway behaviour with Improve
        assert np.array_equal(oracle.rd(U0g, U1g, *Eg, T.conn, stage=1)[0], base_g)
        rf = oracle.ref_rd(U0g, U1g, *Eg, T.conn, improve=True, seed=1000 + k)
        of = oracle.rd(U0g, U1g, *Eg, T.conn, improve=True, seed=1000 + k)
        assert np.array_equal(rf[0], of[0]) and rf[3] == of[3] and _rel(rf[1], of[1]) < 1e-12
        T.accept(prop, oracle.ref_rd(U0, U1, *E, T.conn, improve=True, seed=1000 + k)[0])
    assert flipped_total >= 40, flipped_total
