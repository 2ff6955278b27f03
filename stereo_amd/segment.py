"""Host-side mirror of the two segmentation gateways dispmap_globalstereo calls:

    vgg_segment_ms(A, h_s, h_r, min_sz)             imrender/vgg/vgg_segment_ms.cxx:18-87  (mean shift, EDISON)
    vgg_segment_gb(A, sigma, k, min_sz, compress)   imrender/vgg/vgg_segment_gb.cxx:21-87  (Felzenszwalb-Huttenlocher)

over stereo_segment_* of libstereo_hip.so (include/stereo_hip.h): the per-pixel stages run on the device, the serial graph
work on the host, label maps equal the reference's pixel for pixel.  Same argument meaning and error texts as the gateways.
"""
import ctypes as C
import os

import numpy as np

from . import _lib
from ._lib import StereoHipError

_u8p = C.POINTER(C.c_uint8)
_u32p = C.POINTER(C.c_uint32)
_fp = C.POINTER(C.c_float)

MULTS = (1, 2, 3, 4, 5, 6, 7, 3, 5, 8, 12, 24, 50, 100)   # dispmap_globalstereo.m:122
SEGMENT_PARAMS = (1.0, 1.5, 10.0, 100.0)                  # :121


def _image(A, who):
    A = np.asarray(A)
    if A.dtype != np.uint8 or A.ndim != 3 or A.shape[2] != 3:
        raise StereoHipError("A must be an HxWx3 uint8 array.")          # vgg_segment_ms.cxx:26-27, vgg_segment_gb.cxx:29-30
    return np.asfortranarray(A), A.shape[0], A.shape[1]


def to_uint8(image):
    """MATLAB's uint8(double image): round half away from zero, saturate (dispmap_globalstereo.m:68,378); a single
    channel is repeated three times (:117-119, :386-388)."""
    im = np.asarray(image)
    if im.dtype != np.uint8:
        im = np.clip(np.floor(np.asarray(im, np.float64) + 0.5), 0, 255).astype(np.uint8)
    if im.ndim == 2:
        im = im[:, :, None]
    if im.shape[2] == 1:
        im = np.repeat(im, 3, axis=2)
    return im


def vgg_segment_ms(A, h_s, h_r, min_sz):
    """S = vgg_segment_ms(A, h_s, h_r, min_sz): H x W uint32 labels from 1."""
    A, H, W = _image(A, "vgg_segment_ms")
    out = np.zeros((H, W), np.uint32, order="F")
    err = _lib.errbuf()
    rc = _lib.lib().stereo_segment_ms(A.ctypes.data_as(_u8p), H, W, C.c_double(h_s), C.c_double(h_r), C.c_double(min_sz),
                                      out.ctypes.data_as(_u32p), err, C.c_size_t(len(err)))
    _lib.check(rc, err)
    return np.ascontiguousarray(out)


def vgg_segment_gb(A, sigma, k, min_sz, compress=0):
    """S = vgg_segment_gb(A, sigma, k, min_sz, compress): H x W uint32; union-find roots, or 1, 2, ... by first appearance
    (compress); 0 / the label of 0 on the last row and column, which the library never writes."""
    A, H, W = _image(A, "vgg_segment_gb")
    out = np.zeros((H, W), np.uint32, order="F")
    err = _lib.errbuf()
    rc = _lib.lib().stereo_segment_gb(A.ctypes.data_as(_u8p), H, W, C.c_double(sigma), C.c_double(k), C.c_double(min_sz),
                                      C.c_int(1 if compress else 0), out.ctypes.data_as(_u32p), err, C.c_size_t(len(err)))
    _lib.check(rc, err)
    return np.ascontiguousarray(out)


def segpln_segments(image, workers=None):
    """The 14 segmentation maps segpln builds its proposals on (dispmap_globalstereo.m:121-134): mean shift at seven
    scales, the graph-based segmenter at seven scales.  H x W x 14 uint32, map b = [:, :, b] contiguous (column major).
    The maps do not depend on each other and most of a map's time is its host stage (region graph / std::sort +
    union-find, one core each): they are made side by side by `workers` threads (default: one per map, at most the
    host's cores) -- the library calls release the interpreter lock and share the device."""
    R = to_uint8(image)

    def one(b):
        sp = [p * MULTS[b] for p in SEGMENT_PARAMS]
        return vgg_segment_ms(R, sp[0], sp[1], sp[2]) if b < 7 else vgg_segment_gb(R, 0, sp[3], sp[2], 1)

    nmaps = len(MULTS)
    if workers is None:
        workers = min(nmaps, os.cpu_count() or 1)
    if workers > 1:
        from concurrent.futures import ThreadPoolExecutor
        with ThreadPoolExecutor(int(workers)) as pool:
            maps = list(pool.map(one, range(nmaps)))
    else:
        maps = [one(b) for b in range(nmaps)]
    out = np.zeros((R.shape[0], R.shape[1], nmaps), np.uint32, order="F")   # (map b = out[:, :, b] contiguous, column major:
    for b, m in enumerate(maps):                                            #  what the plane fits read without a copy)
        out[:, :, b] = m
    return out


# ---- the stages, cut at the device / host boundaries (tests; a caller who keeps the filtered image) --------------------
def ms_luv(A):
    A, H, W = _image(A, "ms_luv")
    luv = np.zeros((H * W, 3), np.float32)
    err = _lib.errbuf()
    _lib.check(_lib.lib().stereo_segment_ms_luv(A.ctypes.data_as(_u8p), H, W, luv.ctypes.data_as(_fp), err, C.c_size_t(len(err))), err)
    return luv


def ms_own(A, h_s, h_r):
    """Device stage: every pixel's own mode (H*W x 3, pixel y * W + x) and the event flags (see stereo_hip.h)."""
    A, H, W = _image(A, "ms_own")
    own = np.zeros((H * W, 3), np.float32)
    events = np.zeros(H * W, np.uint8)
    err = _lib.errbuf()
    rc = _lib.lib().stereo_segment_ms_own(A.ctypes.data_as(_u8p), H, W, C.c_double(h_s), C.c_double(h_r), own.ctypes.data_as(_fp),
                                          events.ctypes.data_as(_u8p), err, C.c_size_t(len(err)))
    _lib.check(rc, err)
    return own, events


def ms_finish(A, h_s, h_r, own, events):
    """Host stage: the reference's filtered image from the device stage's output; also returns how many pixels were walked again."""
    A, H, W = _image(A, "ms_finish")
    own = np.ascontiguousarray(own, np.float32)
    events = np.ascontiguousarray(events, np.uint8)
    if own.shape != (H * W, 3) or events.shape != (H * W,):
        raise StereoHipError("ms_finish: own must be H*W x 3, events H*W")
    out = np.zeros((H * W, 3), np.float32)
    walked = C.c_int64()
    err = _lib.errbuf()
    rc = _lib.lib().stereo_segment_ms_finish(A.ctypes.data_as(_u8p), H, W, C.c_double(h_s), C.c_double(h_r), own.ctypes.data_as(_fp),
                                             events.ctypes.data_as(_u8p), out.ctypes.data_as(_fp), C.byref(walked), err, C.c_size_t(len(err)))
    _lib.check(rc, err)
    return out, int(walked.value)


def ms_regions(filtered, H, W, h_r, min_sz):
    f = np.ascontiguousarray(filtered, np.float32)
    if f.shape != (H * W, 3):
        raise StereoHipError("ms_regions: filtered must be H*W x 3")
    out = np.zeros((H, W), np.uint32, order="F")
    err = _lib.errbuf()
    rc = _lib.lib().stereo_segment_ms_regions(f.ctypes.data_as(_fp), H, W, C.c_double(h_r), C.c_double(min_sz), out.ctypes.data_as(_u32p),
                                              err, C.c_size_t(len(err)))
    _lib.check(rc, err)
    return np.ascontiguousarray(out)


def gb_weights(A, sigma):
    """Device stage: 4 edge weights per pixel y * W + x (right, down, down-right, up-right)."""
    A, H, W = _image(A, "gb_weights")
    w = np.zeros((H * W, 4), np.float32)
    err = _lib.errbuf()
    _lib.check(_lib.lib().stereo_segment_gb_weights(A.ctypes.data_as(_u8p), H, W, C.c_double(sigma), w.ctypes.data_as(_fp), err,
                                                    C.c_size_t(len(err))), err)
    return w


def gb_regions(weights, H, W, k, min_sz, compress=0):
    w = np.ascontiguousarray(weights, np.float32)
    if w.shape != (H * W, 4):
        raise StereoHipError("gb_regions: weights must be H*W x 4")
    out = np.zeros((H, W), np.uint32, order="F")
    err = _lib.errbuf()
    rc = _lib.lib().stereo_segment_gb_regions(w.ctypes.data_as(_fp), H, W, C.c_double(k), C.c_double(min_sz), C.c_int(1 if compress else 0),
                                              out.ctypes.data_as(_u32p), err, C.c_size_t(len(err)))
    _lib.check(rc, err)
    return np.ascontiguousarray(out)
