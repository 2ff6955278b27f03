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
