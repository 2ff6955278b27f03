"""-m gpu parity tests: HIP roof-duality fusion (through the C ABI) vs the REFERENCE QPBO
library (oracle/_ref/libref_qpbo.so, built from /root/reference and shipped prebuilt).

Bar: the mask `labelling == 1` (all MATLAB consumes, dispmap_super.m:83) and the full
{-1,0,1} vector bit exact wherever the reference's answer is determined by the energy
(strong persistency, and weak persistency between comparable components); energy and lower
bound within 1e-9 relative (the reference evaluates them from its final reparameterisation,
i.e. with flow-dependent rounding, QPBO.cpp:847-917)."""
import numpy as np
import pytest

from helpers import fusion_problem, glass_problem

pytestmark = pytest.mark.gpu

CASES = [
    # seed, H, W, kernel, tol, integer, nonsub_boost
    (1, 6, 7, 1, 8.0, False, 0.0),
    (2, 12, 14, 1, 8.0, False, 0.0),
    (3, 12, 14, 2, 20.0, False, 0.0),
    (4, 30, 40, 1, 8.0, False, 0.0),
    (5, 30, 40, 1, 8.0, False, 6.0),      # many supermodular terms
    (6, 20, 25, 1, 8.0, True, 0.0),       # integer costs
    (7, 60, 80, 1, 8.0, False, 2.0),
    (8, 1, 12, 1, 8.0, False, 0.0),
    (9, 2, 2, 1, 8.0, False, 0.0),
    # frustrated problems: many unlabelled nodes, weak persistency decides some of them
    ("glass-int-0", 8, 9, 0, 3.0, True, 0.0),
    ("glass-int-1", 8, 9, 0, 3.0, True, 0.0),
    ("glass-int-4", 8, 9, 0, 3.0, True, 0.0),
    ("glass-real-0", 8, 9, 0, 3.0, False, 0.0),
    ("glass-real-3", 8, 9, 0, 3.0, False, 0.0),
    ("glass-int-7", 20, 24, 0, 3.0, True, 0.0),
    ("glass-real-8", 20, 24, 0, 1.0, False, 0.0),
]


def _rel(a, b):
    return abs(a - b) / max(1.0, abs(a), abs(b))


@pytest.mark.parametrize("case", CASES, ids=[str(c[0]) for c in CASES])
def test_rd_matches_reference(case, hip, oracle):
    if not oracle.have_ref_qpbo():
        pytest.skip("oracle/_ref/libref_qpbo.so not present")
    seed, H, W, kernel, tol, integer, boost = case
    if isinstance(seed, str):
        p = glass_problem(int(seed.split("-")[-1]), H, W, field=tol, integer=integer)
    else:
        p = fusion_problem(seed, H, W, kernel=kernel, tol=tol, integer=integer, nonsub_boost=boost)
    args = (p["U0"], p["U1"], p["E00"], p["E01"], p["E10"], p["E11"])
    strong, _, _, _ = oracle.ref_rd(*args, p["conn"], stage=1)
    ref, en_r, lb_r, nu_r = oracle.ref_rd(*args, p["conn"])
    lab, en, lb, nu = hip.rd(*args, p["conn"].T + 1, {})
    # strong persistency is decided by the energy alone
    det = strong >= 0
    assert np.array_equal(lab[det], ref[det])
    assert np.array_equal(lab[det], strong[det])
    # nodes the reference leaves unlabelled after weak persistency lie in one component with
    # their mate: still unlabelled here
    assert np.all(lab[ref < 0] < 0)
    undecided = (~det) & (ref >= 0)
    agree = lab[undecided] == ref[undecided]
    # weak labels may differ only between incomparable components (DFS-order artefact);
    # never produce a worse energy than the reference's labelling
    if not np.all(agree):
        assert en <= en_r + 1e-9 * max(1.0, abs(en_r))
    else:
        assert nu == nu_r
        assert _rel(en, en_r) < 1e-9
    assert _rel(lb, lb_r) < 1e-9
    assert en >= lb - 1e-9 * max(1.0, abs(en))


def test_real_valued_fusion_is_fully_labelled_and_exact(hip, oracle):
    """Teddy-like generic costs: no unlabelled nodes, labels identical to the reference."""
    if not oracle.have_ref_qpbo():
        pytest.skip("oracle/_ref/libref_qpbo.so not present")
    p = fusion_problem(11, 90, 120, kernel=1, tol=8.0)
    args = (p["U0"], p["U1"], p["E00"], p["E01"], p["E10"], p["E11"])
    ref, en_r, lb_r, nu_r = oracle.ref_rd(*args, p["conn"])
    lab, en, lb, nu = hip.rd(*args, p["conn"].T + 1, {})
    assert np.array_equal(lab, ref)
    assert nu == nu_r
    assert _rel(en, en_r) < 1e-9 and _rel(lb, lb_r) < 1e-9


def test_improve_matches_reference(hip, oracle):
    """QPBOI (rd_mex.cpp:91-92): only runs when nodes stay unlabelled; the permutation comes
    from libc rand(), seeded identically for both sides."""
    if not oracle.have_ref_qpbo():
        pytest.skip("oracle/_ref/libref_qpbo.so not present")
    ran = 0
    for seed in range(40, 52):
        p = glass_problem(seed, 8, 9, field=3.0, integer=(seed % 2 == 0))
        args = (p["U0"], p["U1"], p["E00"], p["E01"], p["E10"], p["E11"])
        ref, en_r, lb_r, nu_r = oracle.ref_rd(*args, p["conn"], improve=True, seed=seed)
        oracle.ref_qpbo().ref_srand(seed)
        lab, en, lb, nu = hip.rd(*args, p["conn"].T + 1, {"improve": True})
        assert nu == nu_r
        if nu_r > 0:
            ran += 1
            assert set(np.unique(lab)) <= {0.0, 1.0}
        assert np.array_equal(lab, ref), "seed %d: %d labels differ" % (seed, int((lab != ref).sum()))
        assert _rel(en, en_r) < 1e-9
        assert en >= lb - 1e-9 * max(1.0, abs(en))
    assert ran > 0, "no case exercised Improve"


def test_improve_confined_relabelling_changes_nothing(hip, oracle, monkeypatch):
    """The relabellings inside an Improve step start from the heights of the step's starting flow and only
    look at the tiles the step touched (qpbo.hip, `confined` / `local`); STEREO_HIP_QPBO_CONFINED=0 keeps the
    search from scratch over all nodes.  Same fixpoint: labels, energy and bound bit for bit, on problems
    with both kinds of ambiguous node (neither / both sides connected) and on one with several tiles."""
    if not oracle.have_ref_qpbo():
        pytest.skip("oracle/_ref/libref_qpbo.so not present")
    ran = 0
    for seed, H, W in [(40, 8, 9), (43, 8, 9), (13, 30, 41), (14, 70, 90)]:
        p = glass_problem(seed, H, W, field=3.0, integer=(seed % 2 == 0))
        args = (p["U0"], p["U1"], p["E00"], p["E01"], p["E10"], p["E11"])
        out = {}
        for mode in ("1", "0"):
            monkeypatch.setenv("STEREO_HIP_QPBO_CONFINED", mode)
            oracle.ref_qpbo().ref_srand(seed)
            out[mode] = hip.rd(*args, p["conn"].T + 1, {"improve": True})
        monkeypatch.delenv("STEREO_HIP_QPBO_CONFINED")
        assert np.array_equal(out["1"][0], out["0"][0]), "seed %d" % seed
        assert out["1"][1] == out["0"][1] and out["1"][2] == out["0"][2] and out["1"][3] == out["0"][3]
        ran += int(out["1"][3] > 0)
    assert ran > 0, "no case exercised Improve"


def test_rd_golden_vectors(hip):
    """Committed reference outputs (tests/golden/rd_runs.npz): no oracle/_ref needed."""
    import ctypes
    import os
    from make_golden_rd import RD_RUNS, make_problem
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "rd_runs.npz"))
    libc = ctypes.CDLL(None)
    for name, kind, seed, H, W, params in RD_RUNS:
        p = make_problem(kind, seed, H, W, params)
        args = (p["U0"], p["U1"], p["E00"], p["E01"], p["E10"], p["E11"])
        strong, weak, improved = g[name + "_strong"], g[name + "_weak"], g[name + "_improved"]
        en_r, lb_r, nu_r, en_i = g[name + "_scalars"]
        lab, en, lb, nu = hip.rd(*args, p["conn"].T + 1, {})
        det = strong >= 0
        assert np.array_equal(lab[det], weak[det]), name
        assert np.all(lab[weak < 0] < 0), name
        if np.array_equal(lab, weak):
            assert nu == nu_r and _rel(en, en_r) < 1e-9, name
        assert _rel(lb, lb_r) < 1e-9, name
        libc.srand(seed)
        lab_i, en2, _, _ = hip.rd(*args, p["conn"].T + 1, {"improve": True})
        assert np.array_equal(lab_i, improved), name
        assert _rel(en2, en_i) < 1e-9, name


def test_plan_equals_stateless_and_reference(hip, oracle, monkeypatch):
    """stereo_rd_plan_* (capacities built on the device, buffers reused across moves) must give
    what the stateless stereo_rd gives (graph built on the host per call: STEREO_HIP_RD_CACHE=0), move
    after move, including after a move that left nodes unlabelled."""
    import ctypes
    from stereo_amd.rd import RdPlan
    monkeypatch.setenv("STEREO_HIP_RD_CACHE", "0")
    libc = ctypes.CDLL(None)
    H, W = 20, 24
    probs = [fusion_problem(301, H, W), glass_problem(302, H, W, 3.0, True), fusion_problem(303, H, W, nonsub_boost=5.0),
             glass_problem(304, H, W, 3.0, False), fusion_problem(305, H, W, kernel=2, tol=20.0)]
    plan = RdPlan(H * W, probs[0]["conn"].T)
    for k, p in enumerate(probs):
        args = (p["U0"], p["U1"], p["E00"], p["E01"], p["E10"], p["E11"])
        for improve in (False, True):
            libc.srand(k)
            a = hip.rd(*args, p["conn"].T + 1, {"improve": improve})
            libc.srand(k)
            b = plan.solve(*args, improve=improve)
            assert np.array_equal(a[0], b[0]), (k, improve)
            assert a[3] == b[3] and _rel(a[1], b[1]) < 1e-12 and _rel(a[2], b[2]) < 1e-12, (k, improve)
            if oracle.have_ref_qpbo() and improve:
                r = oracle.ref_rd(*args, p["conn"], improve=True, seed=k)
                assert np.array_equal(r[0], b[0]), k


def test_grid_hint_does_not_change_results(hip):
    """The image-patch work partition (stereo_rd_plan_set_grid) is a schedule, not an algorithm:
    labels / unlabelled counts identical to the index-range partition, energies and bounds to 1e-12."""
    from stereo_amd.rd import RdPlan
    H, W = 70, 90   # several 16 x 32 patches, ragged at both borders
    probs = [fusion_problem(311, H, W), glass_problem(312, H, W, 3.0, True), fusion_problem(313, H, W, nonsub_boost=5.0)]
    plain, hinted = RdPlan(H * W, probs[0]["conn"].T), RdPlan(H * W, probs[0]["conn"].T, grid=(H, W))
    with pytest.raises(hip.StereoHipError, match="must equal N"):
        RdPlan(H * W, probs[0]["conn"].T, grid=(H, W + 1))
    for p in probs:
        args = (p["U0"], p["U1"], p["E00"], p["E01"], p["E10"], p["E11"])
        a, b = plain.solve(*args), hinted.solve(*args)
        assert np.array_equal(a[0], b[0]) and a[3] == b[3]
        assert _rel(a[1], b[1]) < 1e-12 and _rel(a[2], b[2]) < 1e-12


def test_large_grid_move_matches_reference(hip, oracle):
    """A 1000 x 1500 move (2930 tiles: several per workgroup and round, dealt out; heights beyond 16 bits) and a
    frustrated 120 x 160 problem with Improve on the image-patch tiling, against the reference library."""
    import ctypes
    from stereo_amd.rd import RdPlan
    if not oracle.have_ref_qpbo():
        pytest.skip("oracle/_ref/libref_qpbo.so not present")
    H, W = 1000, 1500
    p = fusion_problem(321, H, W, nonsub_boost=3.0)
    args = (p["U0"], p["U1"], p["E00"], p["E01"], p["E10"], p["E11"])
    ref = oracle.ref_rd(*args, p["conn"])
    plan = RdPlan(H * W, p["conn"].T, grid=(H, W))
    for _ in range(2):   # (the second move runs on the buffers, marks and tables the first left behind)
        lab, en, lb, nu = plan.solve(*args)
        if np.array_equal(lab, ref[0]):
            assert nu == ref[3] and _rel(en, ref[1]) < 1e-9
        else:   # weak labels may differ between incomparable components only: never a worse energy
            assert en <= ref[1] + 1e-9 * max(1.0, abs(ref[1]))
            assert np.mean(lab != ref[0]) < 1e-3
        assert _rel(lb, ref[2]) < 1e-9
    plan.close()
    H, W = 120, 160
    p = glass_problem(322, H, W, 3.0, True)
    args = (p["U0"], p["U1"], p["E00"], p["E01"], p["E10"], p["E11"])
    r = oracle.ref_rd(*args, p["conn"], improve=True, seed=7)
    plan = RdPlan(H * W, p["conn"].T, grid=(H, W))
    ctypes.CDLL(None).srand(7)
    b = plan.solve(*args, improve=True)
    assert b[3] == r[3] and b[3] > 0
    assert np.array_equal(b[0], r[0])
    assert _rel(b[1], r[1]) < 1e-9


def test_round_numbers_wrap(hip, monkeypatch):
    """The tiled rounds' marks and the hx words carry round numbers; long launches wrap them (qpbo.hip: mark_all_dirty).
    STEREO_HIP_QPBO_WRAP_AT makes the wrap come after a handful of rounds: moves that need many tiled rounds -- a
    frustrated problem with Improve, a large move with supermodular terms -- must give the labels of the default run."""
    import ctypes
    from stereo_amd.rd import RdPlan
    cases = [(glass_problem(322, 120, 160, 3.0, True), (120, 160), True), (fusion_problem(323, 300, 400, nonsub_boost=3.0), (300, 400), False),
             (glass_problem(324, 60, 70, 1.0, False), (60, 70), True)]
    for p, grid, improve in cases:
        args = (p["U0"], p["U1"], p["E00"], p["E01"], p["E10"], p["E11"])
        out = []
        for wrap in (None, "3", "17"):
            if wrap is None:
                monkeypatch.delenv("STEREO_HIP_QPBO_WRAP_AT", raising=False)
            else:
                monkeypatch.setenv("STEREO_HIP_QPBO_WRAP_AT", wrap)
            plan = RdPlan(grid[0] * grid[1], p["conn"].T, grid=grid)
            ctypes.CDLL(None).srand(11)
            out.append(plan.solve(*args, improve=improve))
            plan.close()
        for o in out[1:]:
            assert np.array_equal(o[0], out[0][0]) and o[3] == out[0][3] and o[1] == out[0][1]


def test_gateway_keeps_the_plan_of_the_last_connectivity(hip, oracle, monkeypatch):
    """stereo_rd (what rd_mex calls) keeps the plan of the last connectivity it saw: same results as the
    stateless path on repeated moves, on a change of connectivity and back, with and without Improve."""
    import ctypes
    libc = ctypes.CDLL(None)
    pa = [fusion_problem(401, 30, 40), glass_problem(402, 30, 40, 3.0, True), fusion_problem(403, 30, 40, nonsub_boost=4.0)]
    pb = [glass_problem(404, 24, 50, 3.0, False), fusion_problem(405, 24, 50)]   # same N, another grid
    seq = [pa[0], pa[1], pb[0], pa[2], pb[1], pa[1]]
    for k, p in enumerate(seq):
        args = (p["U0"], p["U1"], p["E00"], p["E01"], p["E10"], p["E11"])
        for improve in (False, True):
            out = {}
            for mode in ("1", "0"):
                monkeypatch.setenv("STEREO_HIP_RD_CACHE", mode)
                libc.srand(k)
                out[mode] = hip.rd(*args, p["conn"].T + 1, {"improve": improve})
            assert np.array_equal(out["1"][0], out["0"][0]), (k, improve)
            assert out["1"][3] == out["0"][3]
            assert _rel(out["1"][1], out["0"][1]) < 1e-12 and _rel(out["1"][2], out["0"][2]) < 1e-12
    monkeypatch.delenv("STEREO_HIP_RD_CACHE")
