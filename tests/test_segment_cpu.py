"""The segmenters behind dispmap_globalstereo without a GPU: the product's HOST stages
(stereo_amd/csrc/segment_host.cpp through stereo_segment_ms_luv / _ms_finish / _ms_regions / _gb_regions) against

 * the fixtures tests/golden/{teddy,baby2}_segments.npz, made by the reference's OWN segmenters (oracle/_ref:
   vgg_segment_ms(R, 4, 5, 0) and the 14 maps of dispmap_globalstereo.m:121-134), and
 * where the build container made it, the reference's segmenters live (oracle/_ref) on small random images.

The device stages are stood in for by test infrastructure: the mean-shift filter by oracle/segment_oracle.c run with no
shortcut (speed threshold 0: every pixel's own mode, plus the event flags) -- exactly what ms_filter_kernel returns,
bit for bit (tests/test_segment_gpu.py) --, the graph-based edge weights by their float32 definition in NumPy.
No compute call of the product touches a device here."""
import os

import numpy as np
import pytest

from stereo_amd import segment as S

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _pair(name):
    return np.load(os.path.join(GOLD, "%s_pair.npz" % name))["im0"], np.load(os.path.join(GOLD, "%s_segments.npz" % name))


def gb_weights_numpy(im):
    """segment-image.h:41-46 for sigma = 0 (the mask is [1, 0]: smoothing is the identity): sqrt of the summed squared
    channel differences, single precision, in the reference's order of additions."""
    H, W, _ = im.shape
    ch = [im[:, :, c].astype(np.float32) for c in range(3)]
    w = np.zeros((H, W, 4), np.float32)
    ys, xs = np.arange(H)[:, None], np.arange(W)[None, :]
    for t, (dx, dy) in enumerate(((1, 0), (0, 1), (1, 1), (1, -1))):
        y2, x2 = ys + dy, xs + dx
        ok = (x2 < W) & (y2 >= 0) & (y2 < H)
        y2c, x2c = np.clip(y2, 0, H - 1), np.clip(x2, 0, W - 1)
        s = np.zeros((H, W), np.float32)
        for c in ch:
            d = (c - c[y2c, x2c]).astype(np.float32)
            s = (s + d * d).astype(np.float32)
        w[:, :, t] = np.where(ok, np.sqrt(s), 0).astype(np.float32)
    return np.ascontiguousarray(w.reshape(H * W, 4))


def ms_through_the_host_stages(oracle, im, h_s, h_r, min_sz):
    H, W, _ = im.shape
    luv = S.ms_luv(im)
    assert np.array_equal(luv.view(np.uint32), oracle.rgb_to_luv(im).view(np.uint32))
    own, events = oracle.ms_filter(luv, H, W, h_s, h_r, speed_threshold=0.0)          # what the device stage returns
    filtered, walked = S.ms_finish(im, h_s, h_r, own, events)
    return S.ms_regions(filtered, H, W, h_r, min_sz), filtered, int(events.sum()), walked


@pytest.mark.parametrize("name,params,which", [("teddy", (4, 5.0, 0), "segment"), ("teddy", (2, 3.0, 20), 1),
                                               ("baby2", (4, 5.0, 0), "segment"), ("baby2", (3, 4.5, 30), 2)])
def test_mean_shift_host_stages_reproduce_the_reference_maps(name, params, which, oracle):
    im, gold = _pair(name)
    want = gold["segment"] if which == "segment" else gold["segments"][:, :, which]
    seg, filtered, n_events, walked = ms_through_the_host_stages(oracle, im, *params)
    assert np.array_equal(seg, want)
    # the finishing stage IS the reference's scan-order filter: equal to the restatement run with the shortcuts on
    H, W, _ = im.shape
    seq, _ = oracle.ms_filter(oracle.rgb_to_luv(im), H, W, params[0], params[1])
    assert np.array_equal(filtered.view(np.uint32), seq.view(np.uint32))
    assert 0 < walked <= n_events < H * W // 2    # flat regions only: a few per cent of the image is walked on the host


def test_without_the_shortcuts_the_maps_differ(oracle):
    """What pins the threshold: with speedThreshold = 0 (no basin of attraction at all) the Teddy map is another one."""
    im, gold = _pair("teddy")
    H, W, _ = im.shape
    own, _ = oracle.ms_filter(S.ms_luv(im), H, W, 4, 5.0, speed_threshold=0.0)
    seg = S.ms_regions(own, H, W, 5.0, 0)
    assert seg.max() != gold["segment"].max() and not np.array_equal(seg, gold["segment"])


@pytest.mark.parametrize("name", ["teddy", "baby2"])
def test_graph_based_host_stage_reproduces_the_seven_reference_maps(name):
    im, gold = _pair(name)
    H, W, _ = im.shape
    w = gb_weights_numpy(im)
    for b, m in enumerate(S.MULTS[7:]):
        seg = S.gb_regions(w, H, W, 100.0 * m, 10.0 * m, compress=1)
        assert np.array_equal(seg, gold["segments"][:, :, 7 + b]), (name, 7 + b)


def _random_images():
    rng = np.random.default_rng(7)
    yield "noise", rng.integers(0, 256, (23, 31, 3), dtype=np.uint8)
    yield "few colours", rng.integers(0, 3, (19, 27, 3), dtype=np.uint8) * 100            # equal colours everywhere: shortcuts galore
    flat = np.full((16, 21, 3), 90, np.uint8)
    flat[5:9, 4:15] = (200, 30, 60)
    yield "two flat regions", flat
    yield "one colour", np.full((9, 12, 3), 17, np.uint8)
    g = np.linspace(0, 255, 40).astype(np.uint8)
    yield "ramp", np.stack([np.tile(g, (25, 1))] * 3, axis=2)
    yield "one column", rng.integers(0, 256, (14, 1, 3), dtype=np.uint8)
    yield "one row", rng.integers(0, 256, (1, 18, 3), dtype=np.uint8)


def test_host_stages_against_the_reference_segmenters_live(oracle):
    if not oracle.have_ref_segment() or oracle.ref_segment_gb_lib() is None:
        pytest.skip("oracle/_ref not built")
    for name, im in _random_images():
        H, W, _ = im.shape
        for h_s, h_r, mn in ((1, 1.5, 0), (2, 6.5, 4), (4, 5.0, 0), (3, 20.0, 10)):
            if name == "one colour" and mn > 0:
                continue        # (one region and a minimum size: the reference reads a null adjacency list)
            seg = ms_through_the_host_stages(oracle, im, h_s, h_r, mn)[0]
            assert np.array_equal(seg, oracle.ref_segment_ms(im, h_s, h_r, mn)), (name, h_s, h_r, mn)
        w = gb_weights_numpy(im)
        for k, mn, compress in ((50.0, 2, 1), (300.0, 10, 1), (120.0, 0, 0)):
            seg = S.gb_regions(w, H, W, k, mn, compress=compress)
            assert np.array_equal(seg, oracle.ref_segment_gb(im, 0, k, mn, compress)), (name, k, mn, compress)


def test_argument_checks_follow_the_gateways():
    from stereo_amd import StereoHipError
    with pytest.raises(StereoHipError, match="A must be an HxWx3 uint8 array."):     # vgg_segment_ms.cxx:26-27
        S.vgg_segment_ms(np.zeros((4, 5, 3)), 4, 5, 0)
    with pytest.raises(StereoHipError, match="A must be an HxWx3 uint8 array."):     # vgg_segment_gb.cxx:29-30
        S.vgg_segment_gb(np.zeros((4, 5), np.uint8), 0, 100, 10, 1)
    with pytest.raises(StereoHipError, match="zero or negative"):                    # msImageProcessor.cpp:3820-3824
        S.ms_finish(np.zeros((4, 5, 3), np.uint8), 0, 5.0, np.zeros((20, 3), np.float32), np.zeros(20, np.uint8))
