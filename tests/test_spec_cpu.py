"""Host side of the speculative schedule (trws_graph.h: Sweep::Spec), no device: which run is cut, how,
and that the protocol -- runner publishes, segments walk and commit in order, a node of the cut run is done when its
segment commits -- terminates on any number of resident workgroups >= 3 (the runner holds one for the whole sweep, the
two interleaved last rows of the grid need one each; the plan asks for eight) and shortens the sweep."""
import numpy as np
import pytest

from helpers import grid_conn


def _both(H, W):
    from stereo_amd.trws import schedule, spec_schedule
    conn = grid_conn(H, W)
    return [(schedule(H * W, conn.T, d), spec_schedule(H * W, conn.T, d)) for d in (0, 1)]


@pytest.mark.parametrize("shape", [(30, 40), (64, 64), (45, 31), (375, 450)])
def test_the_border_chain_is_cut_into_segments(shape):
    H, W = shape
    chain = 2 * (H + W) - 4
    for d, (s, sp) in enumerate(_both(H, W)):
        assert sp is not None
        assert sp["c1"] - sp["c0"] == chain
        L, S = sp["seg_len"], sp["nseg"]
        assert L == 16 and S == chain // L
        segs = np.flatnonzero(sp["kind"] > 0)
        assert np.array_equal(sp["kind"][segs], 1 + np.arange(S))           # consecutive runs, in order
        lens = np.diff(sp["run_ptr"])[segs]
        assert (lens[:-1] == L).all() and L <= lens[-1] < 2 * L and lens.sum() == chain
        # every other run is a run of the chain schedule, unchanged
        other = np.setdiff1d(np.arange(len(sp["kind"])), segs)
        base = set(zip(s["run_ptr"][:-1].tolist(), s["run_ptr"][1:].tolist()))
        assert all((int(sp["run_ptr"][k]), int(sp["run_ptr"][k + 1])) in base for k in other)
        # tickets: the runner's first, the segments in order where the cut run's ticket was, every run exactly once
        t = sp["ticket_run"]
        assert t[0] == -1 and (t == -1).sum() == 1
        at = int(np.flatnonzero(t == segs[0])[0])
        assert np.array_equal(t[at:at + S], segs)
        assert sorted(t[t >= 0].tolist()) == list(range(len(sp["kind"])))
        # a dependency inside the cut run lies in an earlier segment (it is done when that one commits)
        pos_of = {int(s["rank_at"][p]): p for p in range(sp["c0"], sp["c1"])}
        seg_of = lambda p: min((p - sp["c0"]) // L, S - 1)
        for p in range(sp["c0"], sp["c1"]):
            r = int(s["rank_at"][p])
            for x in s["dep_rank"][s["dep_ptr"][r]:s["dep_ptr"][r + 1]]:
                if int(x) in pos_of:
                    assert seg_of(pos_of[int(x)]) < seg_of(p)


def test_graphs_without_a_long_serial_run_are_left_alone():
    from stereo_amd.trws import spec_schedule
    for H, W in ((5, 6), (1, 200), (12, 14)):       # chains shorter than eight segments; a single row
        conn = grid_conn(H, W)
        assert spec_schedule(H * W, conn.T, 0) is None and spec_schedule(H * W, conn.T, 1) is None
    # a ring: one run, nothing beside it to speed up
    n = 400
    ring = np.stack([np.arange(n), (np.arange(n) + 1) % n])
    assert spec_schedule(n, ring, 0) is None


@pytest.mark.parametrize("workgroups", [3, 4, 8, 256, 10 ** 6])
def test_the_protocol_terminates_and_pays(workgroups):
    from stereo_amd.trws import simulate_schedule, simulate_spec_schedule
    H, W = 40, 56
    for s, sp in _both(H, W):
        plain, ok0 = simulate_schedule(s, workgroups)
        depth, ok, commits = simulate_spec_schedule(s, sp, workgroups, visit=1.0, runner_visit=0.25)
        assert ok0 and ok, "deadlock on %d workgroups" % workgroups
        assert (np.diff(commits) >= 0).all() and commits[0] > 0
        if workgroups >= 256:
            # the serial part of the chain costs a quarter per visit instead of one
            # (the runner holds a workgroup of its own from the start of the sweep: with a handful of them that shows)
            assert depth < plain - 0.5 * (2 * H + W), (depth, plain)
        # and a slow runner is no worse than none (it then paces the chain like the plain schedule)
        # (with a handful of workgroups the runner's own costs the rows a worker: no bound asked for there)
        depth2, ok2, _ = simulate_spec_schedule(s, sp, workgroups, visit=1.0, runner_visit=1.0)
        assert ok2 and (workgroups < 256 or depth2 <= plain + 3 * sp["seg_len"])
