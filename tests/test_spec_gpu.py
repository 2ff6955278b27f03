"""-m gpu: the speculative schedule of the border chain (stereo_trws_plan_spec_stats).

A runner computes the chain's min-plus recurrence ahead, segments of the chain recompute every visit with the certified
routine side by side and commit in order after comparing what they started from with what the segment in front
produced.  Whatever the runner hands over, the results must be the plain schedule's -- which the other tests pin to
the oracle -- bit for bit after every iteration: labels, energy, bound.  The development switches 16384 / 32768 make the
runner's rows / labels WRONG at every third / fifth cut: those segments must be walked a second time, and still nothing
may change.  (65536: rows wrong by a lot.  Round 6 found what a second walk must not inherit from the first: the helper
waves' exchange flags show schedule positions, the second walk visits the same positions, and a flag left standing let
the finishing wave merge the FIRST walk's partial minima -- bound differences of 1e-16 relative from iteration 368 of
configs[1] on, tests/test_full_runs_gpu.py.)
"""
import numpy as np
import pytest

from helpers import grid_conn

pytestmark = pytest.mark.gpu

ENV = ("STEREO_HIP_TRWS_SPEC", "STEREO_HIP_TRWS_DEBUG", "STEREO_HIP_TRWS_SPEC_SEG")


def _solve(monkeypatch, env, kernel, unary, conn, tol, iters, positions, alphas):
    from stereo_amd.trws import TrwsPlan
    for k in ENV:
        monkeypatch.delenv(k, raising=False)
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    N, K = unary.shape
    plan = TrwsPlan(kernel, K, N, conn.T)
    plan.upload(unary.T, alphas, tol, positions=positions)
    out = []
    for _ in range(iters):
        plan.iterate(1, max_relgap=-1e300)
        lab, en, lb, _ = plan.result()
        out.append((lab.copy(), en, lb))
    st = plan.spec_stats()
    plan.close()
    return out, st


def _same(a, b):
    return all(np.array_equal(x[0], y[0]) and x[1] == y[1] and x[2] == y[2] for x, y in zip(a, b))


def _problem(seed, H, W, K, weights):
    rng = np.random.default_rng(seed)
    conn = grid_conn(H, W)
    E = conn.shape[0]
    unary = rng.uniform(0, 40, size=(H * W, K))
    if weights == "unit":
        alphas = np.ones(E)
    elif weights == "random":      # the two directed edges of a pair of pixels get DIFFERENT weights: two messages per hand-over
        alphas = rng.uniform(0.5, 2.0, size=E)
    elif weights == "pairs":       # equal weights within a pair, varying along the chain
        alphas = rng.uniform(0.5, 2.0, size=E)
        key = {}
        for e, (a, b) in enumerate(conn):
            k = (min(a, b), max(a, b))
            alphas[e] = key.setdefault(k, alphas[e])
    else:                          # some zero weights (typeStereoLinear.h:390-396: constant messages)
        alphas = rng.uniform(0.5, 2.0, size=E)
        alphas[rng.random(E) < 0.1] = 0.0
    return unary, conn, alphas


CASES = [
    # seed, H, W, K, tol, step, weights, segment length
    (1, 30, 40, 16, 4.0, 1.0, "unit", None),
    (2, 36, 52, 60, 8.0, 1.0, "random", None),
    (3, 41, 33, 7, 2.0, 1.0, "zeros", None),
    (4, 30, 40, 16, 3.0, 1.0, "pairs", "8"),       # window of three entries: one group of four; short segments
    (5, 64, 64, 33, 7.5, 1.0, "unit", "24"),       # window of seven; long segments
    (6, 35, 45, 64, 3.0, 0.5, "pairs", None),      # K = 64: no idle lane
]


@pytest.mark.parametrize("case", CASES, ids=[str(c[0]) for c in CASES])
def test_speculative_schedule_equals_the_plain_one(case, hip, monkeypatch):
    seed, H, W, K, tol, step, weights, seg = case
    unary, conn, alphas = _problem(seed, H, W, K, weights)
    pos = np.arange(K, dtype=np.float64) * step
    base = {"STEREO_HIP_TRWS_SPEC_SEG": seg} if seg else {}
    plain, st0 = _solve(monkeypatch, dict(base, STEREO_HIP_TRWS_SPEC="0"), 1, unary, conn, tol, 4, pos, alphas)
    assert not st0["active"]
    spec, st = _solve(monkeypatch, base, 1, unary, conn, tol, 4, pos, alphas)
    assert st["active"] and st["commits"] > 0 and st["runner_visits"] > 0
    assert _same(plain, spec)
    rows, st1 = _solve(monkeypatch, dict(base, STEREO_HIP_TRWS_DEBUG="16384"), 1, unary, conn, tol, 4, pos, alphas)
    assert st1["active"] and st1["second_walks"] > 0, "a wrong row of the runner's must cost a second walk"
    assert _same(plain, rows)
    both, st2 = _solve(monkeypatch, dict(base, STEREO_HIP_TRWS_DEBUG="49152"), 1, unary, conn, tol, 4, pos, alphas)
    assert st2["second_walks"] > st1["second_walks"], "a wrong label of the runner's must cost a second walk"
    assert _same(plain, both)
    # a row that is wrong by a lot (every second entry + 0.375): the first walk's messages -- and what its helper waves
    # left in their exchange areas for the twins' shared loops -- are far from the second walk's
    far, st3 = _solve(monkeypatch, dict(base, STEREO_HIP_TRWS_DEBUG="65536"), 1, unary, conn, tol, 4, pos, alphas)
    assert st3["second_walks"] > 0
    assert _same(plain, far)


def test_speculative_schedule_against_the_oracle(hip, oracle, monkeypatch):
    """... and once directly: through the trws.m boundary (the gateway finds the shared positions itself) vs the CPU oracle."""
    for k in ENV:
        monkeypatch.delenv(k, raising=False)
    H, W, K = 31, 38, 12
    unary, conn, alphas = _problem(9, H, W, K, "pairs")
    q = np.tile(np.arange(K, dtype=np.float64), (conn.shape[0], 1))
    lab_o, en_o, lb_o, it_o = oracle.trws(1, unary, conn, q, q, alphas, 3.0, 6, 0.0, mode=1)
    lab, en, lb, it = hip.trws(1, unary.T, conn.T + 1, q.T, q.T, alphas, 3.0, dict(maxiter=6, max_relgap=0.0))
    assert it == it_o and np.array_equal(lab, lab_o) and en == en_o and lb == lb_o


def test_where_the_speculative_schedule_stays_off(hip, oracle, monkeypatch):
    """Positions that are not uniformly spaced, windows of more than eight entries and the quadratic kernel keep the
    plain schedule (and its results)."""
    H, W, K = 30, 40, 10
    unary, conn, alphas = _problem(11, H, W, K, "unit")
    uneven = np.cumsum(np.random.default_rng(3).uniform(0.5, 1.5, K))
    got, st = _solve(monkeypatch, {}, 1, unary, conn, 3.0, 3, uneven, alphas)
    assert not st["active"]
    q = np.tile(uneven, (conn.shape[0], 1))
    lab_o, en_o, lb_o, _ = oracle.trws(1, unary, conn, q, q, alphas, 3.0, 3, -1e300, mode=1)
    assert np.array_equal(got[-1][0], lab_o) and got[-1][1] == en_o and got[-1][2] == lb_o
    wide, stw = _solve(monkeypatch, {}, 1, unary, conn, 9.5, 3, np.arange(K, dtype=np.float64), alphas)
    assert not stw["active"]
    qw = np.tile(np.arange(K, dtype=np.float64), (conn.shape[0], 1))
    lab_o, en_o, lb_o, _ = oracle.trws(1, unary, conn, qw, qw, alphas, 9.5, 3, -1e300, mode=1)
    assert np.array_equal(wide[-1][0], lab_o) and wide[-1][1] == en_o and wide[-1][2] == lb_o
    got2, st2 = _solve(monkeypatch, {}, 2, unary, conn, 9.0, 3, np.arange(K, dtype=np.float64), alphas)
    assert not st2["active"]
    q2 = np.tile(np.arange(K, dtype=np.float64), (conn.shape[0], 1))
    lab_o, en_o, lb_o, _ = oracle.trws(2, unary, conn, q2, q2, alphas, 9.0, 3, -1e300, mode=1)
    assert np.array_equal(got2[-1][0], lab_o) and got2[-1][1] == en_o and got2[-1][2] == lb_o
