"""CPU tests of the product's host-side graph analysis (stereo_trws_analyze, no GPU
needed) against the oracle's independent restatement and the reference probe data."""
import numpy as np
import pytest

from helpers import grid_conn


def _analyze(N, conn):
    from stereo_amd.trws import analyze
    return analyze(N, conn.T)


@pytest.mark.parametrize("shape", [(6, 8), (5, 5), (12, 14), (1, 9), (9, 1), (2, 2), (3, 40)])
def test_grid_structure_matches_oracle(shape, oracle):
    H, W = shape
    conn = grid_conn(H, W)
    a = _analyze(H * W, conn)
    o = oracle.trws_structure(H * W, conn)
    for k in o:
        assert np.array_equal(a[k], o[k]), k


def test_teddy_size_levels(oracle):
    """SURVEY.md Appendix B (reference probe): 375x450 -> 2020 dependency levels, at most
    374 nodes per level."""
    H, W = 375, 450
    conn = grid_conn(H, W)
    a = _analyze(H * W, conn)
    assert int(a["level"].max()) + 1 == 2020
    assert int(np.bincount(a["level"]).max()) == 374
    o = oracle.trws_structure(H * W, conn)
    assert np.array_equal(a["rank"], o["rank"])
    assert np.array_equal(a["fwd_idx"], o["fwd_idx"]) and np.array_equal(a["bwd_idx"], o["bwd_idx"])
    # levels are a valid schedule: every backward neighbour sits on a lower level
    lvl = a["level"]
    assert np.all(lvl[a["tail"]] < lvl[a["head"]])


def test_random_graphs_match_oracle(oracle):
    rng = np.random.default_rng(5)
    for trial in range(300):
        N = int(rng.integers(3, 60))
        E = int(rng.integers(1, 3 * N))
        a_ = rng.integers(0, N, E)
        b_ = rng.integers(0, N, E)
        m = a_ != b_
        conn = np.stack([a_[m], b_[m]], 1)
        if len(conn) == 0:
            continue
        a = _analyze(N, conn)
        o = oracle.trws_structure(N, conn)
        for k in o:
            assert np.array_equal(a[k], o[k]), (trial, k)


def test_invalid_connectivity_is_reported():
    from stereo_amd import StereoHipError
    from stereo_amd.trws import analyze
    with pytest.raises(StereoHipError, match="out of range"):
        analyze(3, np.array([[0, 5], [1, 1]]))
    with pytest.raises(StereoHipError, match="self loops"):
        analyze(3, np.array([[0, 1], [0, 2]]))


def test_fit_plane_to_points_recovers_a_plane():
    """dispmap_ncc.m:67-92 (host side, no GPU needed): noise-free points give their plane with both
    smoothness kernels."""
    import numpy as np
    from stereo_amd.dispmap import dispmap_ncc

    class Fit(dispmap_ncc):
        def __init__(self, kernel):
            self._kernel = kernel

    rng = np.random.default_rng(3)
    xy = rng.uniform(0, 50, size=(2, 60))
    z = 0.3 * xy[0] - 0.2 * xy[1] + 7
    for kernel in (1, 2):
        p = Fit(kernel).fit_plane_to_points(np.vstack([xy, z]))
        assert np.allclose(p, [-0.3, 0.2, 1.0, -7.0], atol=1e-9)
    # (the reference's IRLS weight is w = sqrt(|residual|), dispmap_ncc.m:82 -- it grows with the
    #  residual; the mirror keeps that, so no robustness claim is tested here)


# ---- chain schedule of the descriptor-driven sweep kernels (host only) ----------------------

def _check_schedule(H, W, capacity):
    from stereo_amd.trws import analyze, schedule, simulate_schedule
    conn = grid_conn(H, W)
    N = H * W
    a = analyze(N, conn.T)
    rank = a["rank"]
    # dependency lists by rank, straight from the oriented edges
    tail_r, head_r = rank[a["tail"]], rank[a["head"]]
    levels = int(a["level"].max()) + 1
    out = {}
    for d in (0, 1):
        s = schedule(N, conn.T, d, capacity)
        R = len(s["ticket_run"])
        assert sorted(s["rank_at"].tolist()) == list(range(N))            # every node visited once
        assert sorted(s["ticket_run"].tolist()) == list(range(R))         # every run dispensed once
        pos_of = np.empty(N, np.int64); pos_of[s["rank_at"]] = np.arange(N)
        run_of_pos = np.repeat(np.arange(R), np.diff(s["run_ptr"]))
        need = [set() for _ in range(N)]
        src, dst = (tail_r, head_r) if d == 0 else (head_r, tail_r)
        for x, y in zip(src.tolist(), dst.tolist()):
            need[y].add(x)
        for r in range(N):
            deps = set(s["dep_rank"][s["dep_ptr"][r]:s["dep_ptr"][r + 1]].tolist())
            pr = int(s["pred_rank"][r])
            if pr >= 0:
                # the LDS hand-over comes from the node visited just before in the same run
                assert pos_of[pr] == pos_of[r] - 1 and run_of_pos[pos_of[pr]] == run_of_pos[pos_of[r]]
                deps.add(pr)
            assert deps == need[r], (r, deps, need[r])
            assert len(s["dep_rank"][s["dep_ptr"][r]:s["dep_ptr"][r + 1]]) <= 4
        # dataflow simulation: with every run resident the sweep takes exactly the DAG depth; with
        # `capacity` workgroups taking tickets in order it still completes (no deadlock)
        mk, ok = simulate_schedule(s, 10 ** 9)
        assert ok and mk == levels
        if capacity:
            mk2, ok2 = simulate_schedule(s, capacity)
            assert ok2, "schedule deadlocks with %d resident workgroups" % capacity
            out[d] = mk2 / levels
    return out


def test_loader_look_ahead_rule_cannot_deadlock():
    """The sweep kernels' second loader waits for a node's foreign dependencies two visits ahead where
    descriptor bit 12 allows it (trws_graph.cpp).  Protocol simulation on the host: with the rule the
    sweep finishes on every grid, both directions; looking ahead EVERYWHERE deadlocks on the two
    interleaved last rows (each feeds the other) -- which is what the rule is for; never looking ahead
    is the old protocol."""
    from stereo_amd.trws import schedule, look_ahead_allowed, simulate_look_ahead
    saw_deadlock = False
    for H, W in [(2, 2), (3, 3), (4, 5), (5, 5), (6, 8), (7, 9), (12, 14), (1, 9), (9, 1), (2, 7), (20, 17)]:
        conn = grid_conn(H, W)
        N = H * W
        for d in (0, 1):
            s = schedule(N, conn.T, d, 0)
            rule = look_ahead_allowed(s, d)
            assert simulate_look_ahead(s, np.zeros(N, bool)), (H, W, d, "one visit ahead")
            assert simulate_look_ahead(s, rule), (H, W, d, "descriptor rule")
            if not simulate_look_ahead(s, np.ones(N, bool)):
                saw_deadlock = True
                assert not rule.all()           # the rule forbids at least one node there
            # most nodes may be looked ahead at (the point of the exercise)
            if H >= 5 and W >= 5:
                assert rule.mean() > 0.7, (H, W, d, rule.mean())
    assert saw_deadlock, "the negative control never deadlocked: the simulation does not model the hazard"


def test_chain_schedule_small_grids():
    for H, W in [(2, 2), (3, 3), (5, 5), (6, 8), (7, 9), (12, 14), (1, 9), (9, 1), (2, 7)]:
        _check_schedule(H, W, 0)
        _check_schedule(H, W, 2)
        _check_schedule(H, W, 3)


def test_chain_schedule_splits_the_interleaved_rows():
    from stereo_amd.trws import schedule
    H, W = 30, 40
    conn = grid_conn(H, W)
    for d in (0, 1):
        s = schedule(H * W, conn.T, d, 0)
        lens = np.sort(np.diff(s["run_ptr"]))
        # border chain + top row is the one long run; no other run is longer than a grid row
        assert lens[-1] == 2 * H + 2 * W - 4 and lens[-2] <= W


def test_chain_schedule_more_rows_than_workgroups():
    slow = _check_schedule(120, 90, 16)
    assert max(slow.values()) < 8.0


def test_improve_permutation_is_the_rand_loop():
    """QPBO_extra.cpp:13-27: perm[i] <-> perm[i + floor(rand() / (RAND_MAX + 1) * (N - i))], N - 1 draws of libc
    rand().  The library draws them without glibc's per-call lock (initstate / random_r / setstate on the
    generator's own state array): same permutation, and rand() continues where N - 1 calls would leave it --
    also after a first use (the self-check of the fast path consumes nothing) and twice in a row."""
    import ctypes
    from stereo_amd import _lib
    L = _lib.lib()
    L.stereo_hip_improve_permutation.argtypes = [ctypes.c_int64, ctypes.c_void_p]
    libc = ctypes.CDLL(None)
    libc.rand.restype = ctypes.c_int
    RAND_MAX = 2147483647

    def slow(N):
        perm = list(range(N))
        for i in range(N - 1):
            j = i + int((libc.rand() / (1.0 + RAND_MAX)) * (N - i))
            j = min(j, N - 1)
            perm[i], perm[j] = perm[j], perm[i]
        return np.array(perm, dtype=np.int32)

    for seed, N in [(3, 1), (3, 2), (5, 17), (7, 5000), (11, 20000)]:
        libc.srand(seed)
        want = [slow(N), slow(N)]
        want_next = libc.rand()
        libc.srand(seed)
        got = []
        for _ in range(2):
            out = np.empty(N, dtype=np.int32)
            assert L.stereo_hip_improve_permutation(N, out.ctypes.data) == 0
            got.append(out)
        assert np.array_equal(got[0], want[0]) and np.array_equal(got[1], want[1]), (seed, N)
        assert libc.rand() == want_next, (seed, N)


def test_improve_permutation_from_four_threads_leaves_rand_intact():
    """The park / resume window swaps the process-wide generator state; concurrent Improve moves on
    several host threads (ctypes releases the GIL) must not leave rand() on one of the library's
    static arrays: after the threads are done, srand(7) gives the sequence it gave before."""
    import ctypes
    import threading
    from stereo_amd import _lib
    L = _lib.lib()
    L.stereo_hip_improve_permutation.argtypes = [ctypes.c_int64, ctypes.c_void_p]
    libc = ctypes.CDLL(None)
    libc.rand.restype = ctypes.c_int
    libc.srand(7)
    before = [libc.rand() for _ in range(16)]
    bad = []

    def work():
        out = np.empty(30000, dtype=np.int32)
        for _ in range(40):
            if L.stereo_hip_improve_permutation(30000, out.ctypes.data) != 0:
                bad.append("rc")
            if not np.array_equal(np.sort(out), np.arange(30000)):
                bad.append("not a permutation")

    threads = [threading.Thread(target=work) for _ in range(4)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    assert not bad, bad
    libc.srand(7)
    assert [libc.rand() for _ in range(16)] == before
    # ... and a seeded permutation is still the rand() loop's
    libc.srand(5)
    r = [libc.rand() for _ in range(16)]
    perm = list(range(17))
    for i in range(16):
        j = min(i + int((r[i] / 2147483648.0) * (17 - i)), 16)
        perm[i], perm[j] = perm[j], perm[i]
    libc.srand(5)
    out = np.empty(17, dtype=np.int32)
    assert L.stereo_hip_improve_permutation(17, out.ctypes.data) == 0
    assert out.tolist() == perm
