"""-m gpu: the segmenters behind dispmap_globalstereo, device stages included.

 * ms_filter_kernel: every pixel's own mode and the event flags equal the restated filter's (oracle/segment_oracle.c,
   no shortcut) bit for bit;
 * vgg_segment_ms / vgg_segment_gb end to end: the edge-weight map and all 14 SegPln maps of both example pairs equal
   the reference's own (tests/golden/*_segments.npz, made by oracle/_ref), label for label;
 * where oracle/_ref travelled with the snapshot: small images with every kind of degenerate content against the
   reference's segmenters live, the graph-based one with real smoothing (sigma > 0);
 * dispmap_globalstereo(images, P, ...) WITHOUT `segment=` builds the edge weights a MATLAB user gets."""
import os

import numpy as np
import pytest

from test_segment_cpu import _pair, _random_images, gb_weights_numpy

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("name,h_s,h_r", [("teddy", 4, 5.0), ("baby2", 7, 10.5), ("teddy", 1, 1.5)])
def test_filter_kernel_equals_the_restated_filter(name, h_s, h_r, hip, oracle):
    from stereo_amd import segment as S
    im, _ = _pair(name)
    H, W, _c = im.shape
    own, events = S.ms_own(im, h_s, h_r)
    own_o, events_o = oracle.ms_filter(oracle.rgb_to_luv(im), H, W, h_s, h_r, speed_threshold=0.0)
    assert np.array_equal(own.view(np.uint32), own_o.view(np.uint32))
    assert np.array_equal(events, events_o)


@pytest.mark.parametrize("name", ["teddy", "baby2"])
def test_all_maps_of_an_example_pair_equal_the_references(name, hip):
    from stereo_amd import segment as S
    im, gold = _pair(name)
    assert np.array_equal(S.vgg_segment_ms(im, 4, 5, 0), gold["segment"])              # dispmap_globalstereo.m:391-392
    maps = S.segpln_segments(im.astype(np.float64))                                     # :116-134 (uint8() of the double image)
    for b in range(14):
        assert np.array_equal(maps[:, :, b], gold["segments"][:, :, b]), (name, b)


def test_edge_weight_kernel_equals_its_definition(hip):
    from stereo_amd import segment as S
    im, _ = _pair("teddy")
    assert np.array_equal(S.gb_weights(im, 0).view(np.uint32), gb_weights_numpy(im).view(np.uint32))


def test_small_images_against_the_reference_live(hip, oracle):
    from stereo_amd import segment as S
    if not oracle.have_ref_segment() or oracle.ref_segment_gb_lib() is None:
        pytest.skip("oracle/_ref did not travel")
    for name, im in _random_images():
        for h_s, h_r, mn in ((1, 1.5, 0), (2, 6.5, 4), (4, 5.0, 0), (3, 20.0, 10)):
            if name == "one colour" and mn > 0:
                continue
            assert np.array_equal(S.vgg_segment_ms(im, h_s, h_r, mn), oracle.ref_segment_ms(im, h_s, h_r, mn)), (name, h_s, h_r, mn)
        for sigma, k, mn, compress in ((0, 50.0, 2, 1), (0.5, 300.0, 10, 1), (0.8, 120.0, 0, 0), (1.7, 500.0, 20, 1)):
            assert np.array_equal(S.vgg_segment_gb(im, sigma, k, mn, compress), oracle.ref_segment_gb(im, sigma, k, mn, compress)), (
                name, sigma, k, mn, compress)


def test_globalstereo_object_from_two_images(hip):
    """preprocess() (dispmap_globalstereo.m:377-414) from the images alone: the same edge weights as with the reference's
    segmentation handed in, and segpln() on the maps the object makes itself."""
    import stereo_amd
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "teddy_pair.npz"))
    _, gold = _pair("teddy")
    images = [g["im0"].astype(np.float64), g["im1"].astype(np.float64)]
    P = np.tile(np.hstack([np.eye(3), np.zeros((3, 1))])[:, :, None], (1, 1, 2)); P[0, 3, 1] = -0.25
    start = np.random.default_rng(0).random(images[0].shape[:2]) * 236.0
    a = stereo_amd.dispmap_globalstereo(images, P, (0, 59), 4, start_disparity=start)
    b = stereo_amd.dispmap_globalstereo(images, P, (0, 59), 4, segment=gold["segment"], start_disparity=start)
    assert np.array_equal(a.segment, gold["segment"]) and np.array_equal(a.smooth_weights, b.smooth_weights)
    assert a.energy() == b.energy()
    pa, pb = a.segpln(seed=0), b.segpln([gold["segments"][:, :, k] for k in range(14)], seed=0)
    assert len(pa) == 14 and all(np.array_equal(x, y) for x, y in zip(pa, pb))
