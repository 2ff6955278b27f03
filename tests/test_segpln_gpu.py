"""SegPln proposals of dispmap_globalstereo (dispmap_globalstereo.m:60-201, LO-RANSAC :417-466).

CPU: the oracle's restatement (oracle/terms.py: segpln_wta, segpln_planes) recovers planted planes, is
deterministic in its seed and follows the cited lines on hand-checkable inputs.
-m gpu: the device path (stereo_segpln_wta / stereo_segpln_planes through the C ABI) against that restatement --
the plane fits BIT FOR BIT (same generator for the random triples, same association of every sum and product),
the window-matching scores through their winners: equal disparities except where two scores tie to rounding."""
import os

import numpy as np
import pytest

from oracle import terms as ot

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _planted(H, W, seed, noise=0.01, outliers=0.15, holes=0.03):
    rng = np.random.default_rng(seed)
    seg = (np.arange(H)[:, None] // 20) * ((W + 16) // 17) + (np.arange(W)[None, :] // 17) + 1
    x, y = np.meshgrid(np.arange(1, W + 1.0), np.arange(1, H + 1.0))
    d = np.zeros((H, W))
    truth = {}
    for a in np.unique(seg):
        p = (rng.normal(0, 0.05), rng.normal(0, 0.05), -rng.uniform(20, 60))
        truth[int(a)] = p
        m = seg == a
        d[m] = -(p[0] * x[m] + p[1] * y[m] + p[2])
    d += rng.normal(0, noise, d.shape)
    out = rng.random(d.shape) < outliers
    d[out] = rng.uniform(5, 80, int(out.sum()))
    d[rng.random(d.shape) < holes] = 0          # pixels without a match (corr < 0.07 -> 0, :112)
    return d, seg, truth


def test_oracle_recovers_planted_planes_and_is_deterministic():
    d, seg, truth = _planted(40, 50, 0)
    prop, planes, ninl = ot.segpln_planes(d, seg, seed=3)
    prop2, planes2, ninl2 = ot.segpln_planes(d, seg, seed=3)
    assert np.array_equal(prop, prop2) and np.array_equal(planes, planes2) and np.array_equal(ninl, ninl2)
    x, y = np.meshgrid(np.arange(1, 51.0), np.arange(1, 41.0))
    for a, p in truth.items():
        m = seg == a
        fit = -(planes[a - 1, 0] * x[m] + planes[a - 1, 1] * y[m] + planes[a - 1, 2])
        want = -(p[0] * x[m] + p[1] * y[m] + p[2])
        assert np.max(np.abs(fit - want)) < 1.5, (a, planes[a - 1], p)      # within the RANSAC band of the planted plane
        assert ninl[a - 1] > 0.6 * m.sum()
    # the proposal holds [N1 N2 1 N3] on every pixel of a fitted segment (:183-186)
    P = prop.reshape(4, 50, 40).transpose(0, 2, 1)
    for a in truth:
        m = seg == a
        assert np.all(P[0][m] == planes[a - 1, 0]) and np.all(P[2][m] == 1.0) and np.all(P[3][m] == planes[a - 1, 2])


def test_oracle_small_segments_and_degenerate_input():
    # <= 3 points: no RANSAC (:170-175); <= 2: no plane, the proposal keeps [0 0 1 0] (:157-161, :177)
    d = np.full((4, 4), 10.0)
    seg = np.zeros((4, 4), int)
    seg[0, 0] = seg[1, 0] = 1                   # two pixels
    seg[0, 2] = seg[1, 2] = seg[2, 3] = 2       # three pixels, not collinear
    seg[3, 0] = seg[3, 1] = seg[3, 2] = seg[3, 3] = 3   # four collinear pixels: singular normal equations -> 1e-100 (:193-196)
    prop, planes, ninl = ot.segpln_planes(d, seg, seed=1)
    P = prop.reshape(4, 4, 4).transpose(0, 2, 1)
    assert np.array_equal(P[:, 0, 0], [0, 0, 1, 0]) and ninl[0] == 2
    assert ninl[1] == 3 and abs(planes[1, 2] + 10.0) < 1e-9 and abs(planes[1, 0]) < 1e-9     # fronto-parallel at d = 10
    assert np.all(np.isfinite(prop))
    # three distinct indices, reproducible
    s = [ot.segpln_sample(5, 2, t, 7) for t in range(1, 50)]
    assert all(len(set(v)) == 3 and min(v) >= 0 and max(v) < 7 for v in s) and s == [ot.segpln_sample(5, 2, t, 7) for t in range(1, 50)]


def test_oracle_wta_finds_a_planted_shift():
    rng = np.random.default_rng(2)
    H, W, D = 24, 40, 6
    tex = rng.uniform(0, 255, (H, W + D, 3)).round()
    im0 = tex[:, :W]
    im1 = tex[:, 3:3 + W]                        # im1(x) = im0(x + 3): the match of pixel x sits at x - 3
    P = np.tile(np.hstack([np.eye(3), np.zeros((3, 1))])[:, :, None], (1, 1, 2))
    P[0, 3, 1] = -1.0                            # x' = x - d
    wta = ot.segpln_wta([im0, im1], P, np.arange(D, -1.0, -1.0))
    assert wta.shape == (H, W)
    assert np.all(wta[2:-2, D + 2:W - 2] == 3.0)


@pytest.mark.gpu
def test_device_plane_fits_equal_the_oracle_bit_for_bit(hip):
    from stereo_amd import terms as T
    for seed, (H, W) in enumerate(((40, 50), (63, 47))):
        d, seg, _ = _planted(H, W, 10 + seed)
        seg = seg.copy()
        seg[:3, :2] = 0                          # unsegmented pixels stay [0 0 1 0]
        want = ot.segpln_planes(d, seg, seed=7 + seed)
        got = T.segpln_planes(d, seg, seed=7 + seed)
        assert np.array_equal(got[2], want[2]), (got[2], want[2])
        assert np.array_equal(got[1], want[1], equal_nan=True)
        assert np.array_equal(got[0], want[0])
    # degenerate segments: too few points, singular systems
    d = np.full((4, 4), 10.0)
    seg = np.zeros((4, 4), int)
    seg[0, 0] = seg[1, 0] = 1
    seg[0, 2] = seg[1, 2] = seg[2, 3] = 2
    seg[3, :] = 3
    want = ot.segpln_planes(d, seg, seed=1)
    got = T.segpln_planes(d, seg, seed=1)
    assert np.array_equal(got[0], want[0]) and np.array_equal(got[2], want[2])


@pytest.mark.gpu
def test_device_window_matching_equals_the_oracle(hip):
    from stereo_amd import terms as T
    g = np.load(os.path.join(GOLD, "teddy_crop.npz"))
    im0, im1 = g["im0"].astype(np.float64), g["im1"].astype(np.float64)
    P = np.tile(np.hstack([np.eye(3), np.zeros((3, 1))])[:, :, None], (1, 1, 2))
    P[0, 3, 1] = -0.25                           # example_global.m:17-18
    disps = np.arange(60.0, -1.0, -1.0)          # disp_range [0 15], factor 4, descending (:48-49)
    want = ot.segpln_wta([im0, im1], P, disps)
    got = T.segpln_wta([im0, im1], P, disps)
    assert got.shape == want.shape
    differ = got != want
    # (a different winner needs two scores equal to ~1e-15: exp / log differ in the last bit between libm and the device)
    assert differ.mean() < 1e-3, differ.mean()
    assert (want != 0).mean() > 0.3              # the threshold of 0.07 leaves real matches on this pair


@pytest.mark.gpu
def test_segpln_through_the_class(hip):
    """dispmap_globalstereo.segpln(): proposals over block segmentations at several scales (stand-ins for the
    mean-shift maps), fed to binary_fusion: the energy never goes up, and the planes equal the oracle's."""
    g = np.load(os.path.join(GOLD, "teddy_crop.npz"))
    im0, im1 = g["im0"].astype(np.float64), g["im1"].astype(np.float64)
    H, W = im0.shape[:2]
    P = np.tile(np.hstack([np.eye(3), np.zeros((3, 1))])[:, :, None], (1, 1, 2))
    P[0, 3, 1] = -0.25
    seg0 = (np.arange(H)[:, None] // 16) * 10 + (np.arange(W)[None, :] // 24)
    gs = hip.dispmap_globalstereo([im0, im1], P, [0, 15], 4, segment=seg0, rng=np.random.default_rng(0))
    maps = [(np.arange(H)[:, None] // c) * ((W + c - 1) // c) + (np.arange(W)[None, :] // c) + 1 for c in (8, 12, 20)]
    props = gs.segpln(maps, seed=4)
    assert len(props) == 3 and all(p.shape == (4, H * W) for p in props)
    wta_o = ot.segpln_wta([im0, im1], P, gs.disps, col_thresh=gs.options["col_thresh"])
    if np.array_equal(gs.segpln_wta(), wta_o):   # same winners -> the same planes, bit for bit
        for b, seg in enumerate(maps):
            assert np.array_equal(props[b], ot.segpln_planes(wta_o, seg, seed=4 + b)[0])
    e = gs.energy()
    for p in props:
        gs.binary_fusion(p)
        assert gs.energy() <= e * (1 + 1e-12)
        e = gs.energy()


@pytest.mark.gpu
@pytest.mark.parametrize("pair", ["teddy", "baby2"])
def test_device_plane_fits_on_the_reference_segmentations_equal_the_golden_planes(hip, pair):
    """The 14 segmentation maps the reference's own segmenters make of the example pairs (tests/golden/*_segments.npz)
    hold segments from a handful of pixels to 150 000: the small ones are fitted by a wave, those above 4096 pixels by
    a workgroup of 1024 threads (segpln.hip).  Both must give the oracle's planes -- the committed fixture
    (tests/golden/make_golden_segpln.py: oracle/terms.py on the fixture's own winner-takes-all map) -- bit for bit."""
    from example_inputs import proposals_from_planes
    from stereo_amd import terms as T
    sg = np.load(os.path.join(GOLD, "%s_segments.npz" % pair))
    pl = np.load(os.path.join(GOLD, "%s_segpln_planes.npz" % pair))
    want = proposals_from_planes(sg["segments"], [pl["planes_%d" % b] for b in range(14)])
    largest = 0
    for b in range(14):
        seg = sg["segments"][:, :, b]
        largest = max(largest, int(np.bincount(seg.ravel().astype(np.int64))[1:].max()))
        got = T.segpln_planes(pl["wta"], seg, seed=b)[0]
        assert np.array_equal(got, want[b], equal_nan=True), (pair, b, int((got != want[b]).any(axis=0).sum()))
    assert largest > 4096   # (the large-segment kernel was exercised)
    # all fourteen in one call (stereo_segpln_planes_batch: a stream and scratch per map, the maps side by side on the
    # device): the same proposals, planes and inlier counts as one call per map -- twice, the second time on warm scratch
    for rep in range(2):
        batch = T.segpln_planes_batch(pl["wta"], sg["segments"], list(range(14)))
        for b in range(14):
            assert np.array_equal(batch[b][0], want[b], equal_nan=True), (pair, b, rep)
            one = T.segpln_planes(pl["wta"], sg["segments"][:, :, b], seed=b, want_proposal=False)
            assert one[0] is None and np.array_equal(batch[b][1], one[1], equal_nan=True) and np.array_equal(batch[b][2], one[2]), (pair, b, rep)
    # fewer maps than before on the same scratch, in another order, planes only
    few = T.segpln_planes_batch(pl["wta"], [sg["segments"][:, :, b] for b in (13, 2, 7)], [13, 2, 7], want_proposal=False)
    for out, b in zip(few, (13, 2, 7)):
        one = T.segpln_planes(pl["wta"], sg["segments"][:, :, b], seed=b)
        assert out[0] is None and np.array_equal(out[1], one[1], equal_nan=True) and np.array_equal(out[2], one[2]), (pair, b)


@pytest.mark.gpu
def test_batch_call_refuses_bad_arguments(hip):
    import stereo_amd
    from stereo_amd import terms as T
    wta = np.ones((6, 7))
    seg = np.ones((6, 7), np.int32)
    with pytest.raises(stereo_amd.StereoHipError):
        T.segpln_planes_batch(wta, [seg, seg], [0])            # a seed per map
    with pytest.raises(stereo_amd.StereoHipError):
        T.segpln_planes_batch(wta, [seg[:5]], [0])             # the image's shape
    bad = seg.copy(); bad[0, 0] = -1
    with pytest.raises(stereo_amd.StereoHipError):
        T.segpln_planes_batch(wta, [seg, bad], [0, 1])         # labels 0 .. S
    out = T.segpln_planes_batch(wta, [np.zeros((6, 7), np.int32)], [0])   # no segment at all: the default plane everywhere
    assert out[0][1].shape == (0, 3) and np.array_equal(out[0][0], np.tile(np.array([[0.0], [0.0], [1.0], [0.0]]), (1, 42)))
