"""-m gpu: the speculative schedule of the border chain in trws_wide_kernel (64 < K <= 256, even K; trws_wspec.h).

As tests/test_spec_gpu.py for the narrow kernel: whatever the runner hands over, labels, energy and bound must be the plain
schedule's -- which test_trws_wide_gpu.py pins to the oracle -- bit for bit after every iteration, also with the runner's
rows / labels made wrong on purpose (STEREO_HIP_TRWS_DEBUG 16384 / 32768 / 65536: those segments must be walked a second
time)."""
import numpy as np
import pytest

from test_spec_gpu import _problem, _same, _solve

pytestmark = pytest.mark.gpu

CASES = [
    # seed, H, W, K, tol, step, weights, segment length
    (21, 30, 40, 128, 6.0, 1.0, "unit", None),
    (22, 28, 44, 200, 8.0, 1.0, "random", "8"),     # two messages per hand-over, short segments
    (23, 36, 34, 256, 3.5, 0.5, "pairs", None),     # window of seven, every lane busy
    (24, 33, 41, 66, 2.0, 1.0, "zeros", "8"),       # just above the narrow kernel's range; constant messages
]


@pytest.mark.parametrize("case", CASES, ids=[str(c[0]) for c in CASES])
def test_speculative_schedule_of_the_wide_kernel_equals_the_plain_one(case, hip, monkeypatch):
    seed, H, W, K, tol, step, weights, seg = case
    unary, conn, alphas = _problem(seed, H, W, K, weights)
    pos = np.arange(K, dtype=np.float64) * step
    base = {"STEREO_HIP_TRWS_SPEC_SEG": seg} if seg else {}
    plain, st0 = _solve(monkeypatch, dict(base, STEREO_HIP_TRWS_SPEC="0"), 1, unary, conn, tol, 4, pos, alphas)
    assert not st0["active"]
    spec, st = _solve(monkeypatch, base, 1, unary, conn, tol, 4, pos, alphas)
    assert st["active"] and st["commits"] > 0 and st["runner_visits"] > 0
    assert _same(plain, spec)
    for dbg in ("16384", "49152", "65536"):
        got, st1 = _solve(monkeypatch, dict(base, STEREO_HIP_TRWS_DEBUG=dbg), 1, unary, conn, tol, 4, pos, alphas)
        assert st1["active"] and st1["second_walks"] > 0, "a wrong row / label of the runner's must cost a second walk"
        assert _same(plain, got), dbg


def test_the_wide_speculative_schedule_against_the_oracle(hip, oracle, monkeypatch):
    from test_spec_gpu import ENV
    for k in ENV:
        monkeypatch.delenv(k, raising=False)
    H, W, K = 30, 38, 96
    unary, conn, alphas = _problem(29, H, W, K, "pairs")
    q = np.tile(np.arange(K, dtype=np.float64), (conn.shape[0], 1))
    lab_o, en_o, lb_o, it_o = oracle.trws(1, unary, conn, q, q, alphas, 5.0, 4, 0.0, mode=1)
    lab, en, lb, it = hip.trws(1, unary.T, conn.T + 1, q.T, q.T, alphas, 5.0, dict(maxiter=4, max_relgap=0.0))
    assert it == it_o and np.array_equal(lab, lab_o) and en == en_o and lb == lb_o


def test_odd_label_counts_keep_the_plain_schedule(hip, monkeypatch):
    unary, conn, alphas = _problem(31, 30, 40, 99, "unit")
    got, st = _solve(monkeypatch, {}, 1, unary, conn, 4.0, 2, np.arange(99, dtype=np.float64), alphas)
    assert not st["active"]
