"""-m gpu parity tests: HIP TRW-S (through the C ABI) vs the CPU oracle.

Bar: labels bit exact, energy / lower bound / iteration count bit exact (the
device computes every term with the reference's association and the host sums
them in the reference's order, so there is no tolerance to state).
"""
import numpy as np
import pytest

from helpers import trws_problem

pytestmark = pytest.mark.gpu

CASES = [
    # seed, H, W, K, kernel, kind, integer, tol, maxiter, relgap
    (1, 5, 6, 4, 1, "general", False, 2.0, 4, 0.0),
    (2, 7, 9, 6, 2, "general", False, 2.0, 4, 0.0),
    (3, 12, 14, 8, 1, "general", False, 1.5, 5, 1e-4),
    (4, 8, 7, 5, 1, "fronto", False, 2.0, 6, 0.0),
    (5, 8, 7, 5, 2, "fronto", False, 4.0, 6, 0.0),
    (6, 20, 30, 16, 1, "general", False, 8.0, 5, 0.0),
    (7, 20, 30, 16, 2, "general", False, 3.0, 5, 0.0),
    (8, 9, 11, 7, 1, "general", True, 2.0, 6, 0.0),     # integer costs: exact ties
    (9, 9, 11, 7, 2, "general", True, 4.0, 6, 0.0),
    (10, 10, 12, 12, 1, "fronto", True, 3.0, 8, 0.0),   # ties on the uniform grid
    (11, 6, 5, 70, 1, "fronto", False, 8.0, 3, 0.0),    # K > 64: two label chunks per lane
    (12, 6, 5, 130, 2, "general", False, 30.0, 3, 0.0),
    (13, 1, 9, 5, 1, "general", False, 2.0, 4, 0.0),    # a chain
    (14, 2, 2, 3, 1, "general", False, 2.0, 3, 0.0),
    (15, 40, 50, 16, 1, "general", False, 2.0, 30, 1e-3),  # stops on the relative gap
]


@pytest.mark.parametrize("case", CASES, ids=[str(c[0]) for c in CASES])
def test_trws_matches_oracle(case, hip, oracle):
    seed, H, W, K, kernel, kind, integer, tol, maxiter, relgap = case
    p = trws_problem(seed, H, W, K, kind=kind, integer=integer)
    lab_o, en_o, lb_o, it_o = oracle.trws(kernel, p["unary"], p["conn"], p["q"], p["qprim"],
                                          p["alphas"], tol, maxiter, relgap, mode=1)
    lab, en, lb, it = hip.trws(kernel, p["unary"].T, p["conn"].T + 1, p["q"].T, p["qprim"].T,
                               p["alphas"], tol, dict(maxiter=maxiter, max_relgap=relgap))
    assert it == it_o
    assert np.array_equal(lab, lab_o), "labels differ at %d nodes" % int((lab != lab_o).sum())
    assert en == en_o
    assert lb == lb_o


def test_shared_positions_equal_materialised(hip, oracle):
    """The fronto-parallel fast path (one positions vector) must equal the K x E form."""
    from stereo_amd.trws import TrwsPlan
    p = trws_problem(21, 14, 17, 20, kind="fronto")
    lab_o, en_o, lb_o, it_o = oracle.trws(1, p["unary"], p["conn"], p["q"], p["qprim"],
                                          p["alphas"], 3.0, 7, 0.0, mode=1)
    plan = TrwsPlan(1, 20, 14 * 17, p["conn"].T)
    plan.upload(p["unary"].T, p["alphas"], 3.0, positions=np.arange(20.0))
    plan.iterate(7)
    lab, en, lb, it = plan.result()
    assert np.array_equal(lab, lab_o) and en == en_o and lb == lb_o and it == it_o
    # and a reset reproduces the run
    plan.reset()
    plan.iterate(7)
    lab2, en2, lb2, _ = plan.result()
    assert np.array_equal(lab2, lab) and en2 == en and lb2 == lb


def test_minplus_mode_matches_bruteforce_oracle(hip, oracle):
    from stereo_amd.trws import TrwsPlan, MESSAGES_MINPLUS
    p = trws_problem(22, 10, 9, 9, kind="general")
    lab_o, en_o, lb_o, _ = oracle.trws(2, p["unary"], p["conn"], p["q"], p["qprim"],
                                       p["alphas"], 2.5, 5, 0.0, mode=0)
    plan = TrwsPlan(2, 9, 90, p["conn"].T, message_mode=MESSAGES_MINPLUS)
    plan.upload(p["unary"].T, p["alphas"], 2.5, q=p["q"].T, qprim=p["qprim"].T)
    plan.iterate(5)
    lab, en, lb, _ = plan.result()
    assert np.array_equal(lab, lab_o) and en == en_o and lb == lb_o


def test_unsupported_kernel_fails(hip):
    p = trws_problem(23, 3, 3, 3)
    with pytest.raises(hip.StereoHipError, match="Unsupported kernel"):
        hip.trws(3, p["unary"].T, p["conn"].T + 1, p["q"].T, p["qprim"].T, p["alphas"], 1.0, {})


def test_golden_runs(hip):
    """Committed golden runs (tests/golden/trws_runs.npz: restated core driving the
    REFERENCE message functions): no oracle library needed on the box."""
    import os
    from make_golden import RUNS
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "trws_runs.npz"))
    for name, seed, H, W, K, kernel, kind, integer, tol, maxiter, relgap in RUNS:
        p = trws_problem(seed, H, W, K, kind=kind, integer=integer)
        lab, en, lb, it = hip.trws(kernel, p["unary"].T, p["conn"].T + 1, p["q"].T, p["qprim"].T,
                                   p["alphas"], tol, dict(maxiter=maxiter, max_relgap=relgap))
        assert np.array_equal(lab.astype(np.int32), g[name + "_labels"]), name
        assert np.array_equal(np.array([en, lb, it]), g[name + "_scalars"]), name


def test_teddy_size_properties(hip):
    """Full BASELINE size (450x375x60): size-independent properties instead of the oracle --
    the lower bound never decreases, energy >= bound, a reset reproduces the run bit for bit,
    and the serial-envelope path gives the same bits as the certified fast path."""
    import os
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from bench import synthetic_volume
    from helpers import grid_conn
    from stereo_amd.trws import TrwsPlan
    H, W, K = 375, 450, 60
    conn = grid_conn(H, W)
    unary = synthetic_volume(H, W, K, 3)
    plan = TrwsPlan(1, K, H * W, conn.T)
    plan.upload(unary.T, np.ones(conn.shape[0]), 8.0, positions=np.arange(K, dtype=np.float64))
    lbs, ens = [], []
    for _ in range(4):
        plan.iterate(1, max_relgap=-1e300)
        _, en, lb, _ = plan.result(want_labels=False)
        lbs.append(lb); ens.append(en)
    assert all(b2 >= b1 for b1, b2 in zip(lbs, lbs[1:]))
    assert all(e >= b for e, b in zip(ens, lbs))
    lab1, en1, lb1, it1 = plan.result()
    assert it1 == 4 and lab1.min() >= 1 and lab1.max() <= K
    plan.reset()
    plan.iterate(4, max_relgap=-1e300)
    lab2, en2, lb2, _ = plan.result()
    assert np.array_equal(lab1, lab2) and en1 == en2 and lb1 == lb2


@pytest.mark.parametrize("kernel,K,general", [(1, 60, False), (2, 16, True)])
def test_teddy_size_matches_oracle(kernel, K, general, hip, oracle):
    """Full 450x375 grid, two iterations, bit for bit against the CPU oracle: more grid rows than
    resident workgroups, the chain schedule with its two mutually dependent interleaved rows,
    the border chain of 1646 serial visits."""
    import os
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from bench import synthetic_volume
    from helpers import grid_conn
    from stereo_amd.trws import TrwsPlan
    H, W = 375, 450
    conn = grid_conn(H, W)
    E = conn.shape[0]
    unary = synthetic_volume(H, W, K, 7)
    rng = np.random.default_rng(17)
    alphas = rng.uniform(0.5, 2.0, size=E)
    tol = 8.0 if kernel == 1 else 9.0
    plan = TrwsPlan(kernel, K, H * W, conn.T)
    if general:
        q = np.arange(K, dtype=np.float64)[None, :] + 0.25 * rng.random((E, K))
        qp = np.arange(K, dtype=np.float64)[None, :] + 0.25 * rng.random((E, K))
        plan.upload(unary.T, alphas, tol, q=q.T, qprim=qp.T)
    else:
        pos = np.arange(K, dtype=np.float64)
        q = qp = np.tile(pos, (E, 1))
        plan.upload(unary.T, alphas, tol, positions=pos)
    ref = oracle.trws(kernel, unary, conn, q, qp, alphas, tol, 2, -1e300, mode=1)
    plan.iterate(2, max_relgap=-1e300)
    lab, en, lb, it = plan.result()
    assert plan.path() == 2
    assert np.array_equal(lab, ref[0]), "labels differ at %d nodes" % int((lab != ref[0]).sum())
    assert en == ref[1] and lb == ref[2] and it == ref[3]


@pytest.mark.parametrize("source", ["teddy", "synthetic"])
def test_named_workload_ncc_volume_full_size_matches_oracle(source, hip, oracle):
    """BASELINE.json configs[1] / the bench's workload at full size: the NCC cost volume (5x5xRGB,
    unary = 40 (1 - ncc), dispmap_ncc.m:107-115) of the reference's Teddy pair (tests/golden/
    teddy_pair.npz = data/teddy/im2.png, im6.png) and of the bench's synthetic pair, 450 x 375 x 60
    fronto-parallel labels, kernel 1, tol 8: five TRW-S iterations bit for bit against the oracle.
    The masked columns of the volume are flat (dispmap_ncc.m:190-191), so a few per cent of the
    messages fail the certificate and take the reference's serial envelope construction under the
    sweep kernel's real scheduling -- asserted, this is what the noise-volume test above never does."""
    import os
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from helpers import grid_conn
    from stereo_amd import terms as T
    from stereo_amd.trws import TrwsPlan
    H, W, K = 375, 450, 60
    if source == "teddy":
        g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "teddy_pair.npz"))
        im0, im1 = g["im0"].astype(np.float64), g["im1"].astype(np.float64)
    else:
        from bench import synthetic_pair
        im0, im1 = synthetic_pair(H, W, K, seed=0)
    assert im0.shape == (H, W, 3)
    ncc = T.ncc_volume(im0, im1, np.arange(K, dtype=np.float64), 2, layout=1)       # label-fastest: K x N
    unary = np.ascontiguousarray(40.0 * (1.0 - ncc.T))                               # N x K
    conn = grid_conn(H, W)
    E = conn.shape[0]
    pos = np.arange(K, dtype=np.float64)
    plan = TrwsPlan(1, K, H * W, conn.T)
    plan.upload(unary.T, np.ones(E), 8.0, positions=pos)
    plan.serial_messages(reset=True)
    iters = 5
    plan.iterate(iters, max_relgap=-1e300)
    lab, en, lb, it = plan.result()
    serial = plan.serial_messages()
    q = np.tile(pos, (E, 1))
    ref = oracle.trws(1, unary, conn, q, q, np.ones(E), 8.0, iters, -1e300, mode=1)
    assert plan.path() == 2
    assert serial > 0.0005 * 2 * E * iters, "only %d messages took the serial construction" % serial   # (1 % of them before round 5's exact-tie rule, 0.17 % since -- counted once per pair of twin messages)
    assert np.array_equal(lab, ref[0]), "labels differ at %d nodes" % int((lab != ref[0]).sum())
    assert en == ref[1] and lb == ref[2] and it == ref[3]


INDEX_ORDER = [
    # seed, H, W, K, kernel, kind, integer, tol, iters
    (401, 12, 14, 8, 1, "general", False, 1.5, 5),
    (402, 20, 30, 16, 2, "general", False, 3.0, 4),
    (403, 10, 12, 12, 1, "fronto", True, 3.0, 6),
    (404, 9, 8, 100, 1, "general", False, 3.0, 3),
    (405, 30, 26, 200, 1, "fronto", False, 8.0, 3),
]


@pytest.mark.parametrize("case", INDEX_ORDER, ids=[str(c[0]) for c in INDEX_ORDER])
def test_index_order_option_matches_oracle_in_that_order(case, hip, oracle):
    """STEREO_TRWS_ORDER_INDEX: MRFEnergy's node order when SetAutomaticOrdering is not called
    (H + W - 1 dependency levels on a grid).  Not the gateway's results -- an explicit option -- but
    bit-identical to the oracle run in the same order; and different from the gateway order's."""
    from stereo_amd.trws import TrwsPlan, ORDER_INDEX
    seed, H, W, K, kernel, kind, integer, tol, iters = case
    p = trws_problem(seed, H, W, K, kind=kind, integer=integer)
    lab_o, en_o, lb_o, it_o = oracle.trws(kernel, p["unary"], p["conn"], p["q"], p["qprim"], p["alphas"], tol,
                                          iters, -1e300, mode=1, ordering=1)
    plan = TrwsPlan(kernel, K, H * W, p["conn"].T, message_mode=ORDER_INDEX)
    if kind == "fronto":
        plan.upload(p["unary"].T, p["alphas"], tol, positions=np.arange(K, dtype=np.float64))
    else:
        plan.upload(p["unary"].T, p["alphas"], tol, q=p["q"].T, qprim=p["qprim"].T)
    assert plan.info()["levels"] == H + W - 1
    plan.iterate(iters, max_relgap=-1e300)
    lab, en, lb, it = plan.result()
    assert it == it_o and np.array_equal(lab, lab_o) and en == en_o and lb == lb_o
    ref = oracle.trws(kernel, p["unary"], p["conn"], p["q"], p["qprim"], p["alphas"], tol, iters, -1e300, mode=1)
    assert ref[2] != lb_o     # (the gateway's order gives another bound after the same iterations)


def test_gateway_keeps_the_plan_and_finds_shared_positions(hip, oracle, monkeypatch):
    """stereo_trws (what trws_mex reaches) keeps the plan of the last problem and uploads one
    vector when every column of q / qprim is that vector: first call, cached call, a call with
    other inputs on the same plan, a different problem (cache miss) and the uncached path
    (STEREO_HIP_TRWS_CACHE=0) all give the oracle's bits -- like test_rd_gpu's cached plan test."""
    from stereo_amd import _lib
    _lib.lib().stereo_trws_cache_clear()
    fr = trws_problem(31, 14, 17, 20, kind="fronto")            # K x E columns all 0..19
    fr2 = trws_problem(32, 14, 17, 20, kind="fronto")           # same graph, other unaries / weights
    ge = trws_problem(33, 14, 17, 20, kind="general")           # same graph and K, general planes
    other = trws_problem(34, 9, 11, 7, kind="general")          # another graph
    # a fronto problem whose LAST edge alone differs: must not be taken for shared positions
    odd = {k: np.array(v) for k, v in fr.items()}
    odd["q"][-1, 3] += 0.25

    def both(p, kernel=1, tol=3.0, it=6):
        want = oracle.trws(kernel, p["unary"], p["conn"], p["q"], p["qprim"], p["alphas"], tol, it, 0.0, mode=1)
        got = hip.trws(kernel, p["unary"].T, p["conn"].T + 1, p["q"].T, p["qprim"].T, p["alphas"], tol,
                       dict(maxiter=it, max_relgap=0))
        assert np.array_equal(got[0], want[0]) and got[1:] == want[1:], (got[1:], want[1:])
        return got

    a = both(fr)            # miss: plan built, shared positions
    b = both(fr)            # hit
    assert np.array_equal(a[0], b[0]) and a[1:] == b[1:]
    both(fr2)               # hit, new inputs
    both(ge)                # hit, K x E positions on the plan that ran shared positions before
    both(fr)                # ... and back
    both(odd)
    both(fr, kernel=2, tol=9.0)   # miss: the kernel is part of the key
    both(other)             # miss
    both(fr)                # miss again
    monkeypatch.setenv("STEREO_HIP_TRWS_CACHE", "0")
    c = both(fr)
    assert np.array_equal(a[0], c[0]) and a[1:] == c[1:]
    monkeypatch.delenv("STEREO_HIP_TRWS_CACHE")
    _lib.lib().stereo_trws_cache_clear()
    both(ge)
    _lib.lib().stereo_trws_cache_clear()
