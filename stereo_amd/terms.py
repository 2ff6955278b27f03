"""Host-side mirrors of the reference's term builders / cost-volume functions over
the C ABI (``stereo_pairwise_terms`` ... ``stereo_globalstereo_unary``).  Arrays use
MATLAB shapes: planes 4 x N, points 2 x N, connectivity 2 x E (ZERO based here, these
are internal helpers of the dispmap classes), images H x W x C."""
import ctypes as C

import numpy as np

from . import _lib
from ._lib import StereoHipError


def _f(a):
    return np.asfortranarray(a, dtype=np.float64)


def _p(a, t=C.c_double):
    return a.ctypes.data_as(C.POINTER(t)) if a is not None else None


def _call(fn, *args):
    err = _lib.errbuf()
    rc = fn(*args, err, C.c_size_t(len(err)))
    _lib.check(rc, err)


def _conn(conn0):
    c = np.asarray(conn0)
    if c.ndim != 2 or c.shape[0] != 2:
        raise StereoHipError("connectivity must be 2 x E")
    return np.asfortranarray(c, dtype=np.uint32)


def get_points(H, W):
    """dispmap_super.m:275-278"""
    cols, rows = np.meshgrid(np.arange(1, W + 1, dtype=np.float64), np.arange(1, H + 1, dtype=np.float64))
    return np.asfortranarray(np.stack([cols.T.ravel(), rows.T.ravel()]))


def construct_neighborhood(H, W):
    """dispmap_super.m:279-302 -> 2 x E zero-based [ind1; ind2]."""
    nodenr = np.arange(H * W, dtype=np.int64).reshape(W, H).T
    s1, f1 = nodenr[:-1, :].T.ravel(), nodenr[1:, :].T.ravel()
    s2, f2 = nodenr[:, :-1].T.ravel(), nodenr[:, 1:].T.ravel()
    return np.stack([np.concatenate([s1, f1, s2, f2]), np.concatenate([f1, s1, f2, s2])])


def pairwise_terms(kernel, conn0, points, assignment, proposal, weights, tol, d_min=0.0, d_step=0.0):
    conn = _conn(conn0)
    E, N = conn.shape[1], points.shape[1]
    points, assignment = _f(points), _f(assignment)
    proposal = _f(proposal) if proposal is not None else None
    weights = _f(np.asarray(weights, np.float64).reshape(-1))
    out = [np.zeros(E) for _ in range(4)]
    _call(_lib.lib().stereo_pairwise_terms, C.c_int(int(kernel)), C.c_int64(N), C.c_int64(E),
          _p(conn, C.c_uint32), _p(points), _p(assignment), _p(proposal), _p(weights), C.c_double(tol),
          C.c_double(d_min), C.c_double(d_step), _p(out[0]), _p(out[1]), _p(out[2]), _p(out[3]))
    return out if proposal is not None else out[0]


def trws_positions(conn0, points, proposals, d_min=0.0, d_step=0.0):
    """proposals: list of 4 x N plane arrays -> q, qprim (K x E, Fortran order)."""
    conn = _conn(conn0)
    E, N, K = conn.shape[1], points.shape[1], len(proposals)
    stack = np.asfortranarray(np.stack([np.asarray(P, np.float64) for P in proposals], axis=2))  # 4 x N x K
    q = np.zeros((K, E), order="F")
    qprim = np.zeros((K, E), order="F")
    _call(_lib.lib().stereo_trws_positions, C.c_int64(N), C.c_int64(E), C.c_int(K), _p(conn, C.c_uint32),
          _p(_f(points)), _p(stack), C.c_double(d_min), C.c_double(d_step), _p(q), _p(qprim))
    return q, qprim


def ncc_volume(im0, im1, disparities, patchsize=2, layout=0):
    im0, im1 = _f(im0), _f(im1)
    H, W, Cn = im0.shape
    if Cn != 3:
        raise StereoHipError("compute_ncc needs RGB images")
    d = _f(np.asarray(disparities, np.float64).reshape(-1))
    D = d.shape[0]
    out = np.zeros((H, W, D), order="F") if layout == 0 else np.zeros((D, H * W), order="F")
    _call(_lib.lib().stereo_ncc_volume, _p(im0), _p(im1), C.c_int(H), C.c_int(W), _p(d), C.c_int(D),
          C.c_int(int(patchsize)), C.c_int(int(layout)), _p(out))
    return out


def _vol_dims(ncc, layout, H, W):
    if layout == 0:
        return ncc.shape[0], ncc.shape[1], ncc.shape[2]
    return H, W, ncc.shape[0]


def ncc_unary(ncc, disparities, unary_weight, assignment, layout=0, shape=None):
    H, W, D = _vol_dims(ncc, layout, *(shape or (0, 0)))
    U = np.zeros(H * W)
    _call(_lib.lib().stereo_ncc_unary, _p(_f(ncc)), C.c_int(H), C.c_int(W), C.c_int(D), C.c_int(layout),
          _p(_f(np.asarray(disparities, np.float64).reshape(-1))), C.c_double(unary_weight),
          _p(_f(assignment)), _p(U))
    return U


def ncc_best_disp(ncc, disparities, layout=0, shape=None):
    H, W, D = _vol_dims(ncc, layout, *(shape or (0, 0)))
    best = np.zeros((H, W), order="F")
    _call(_lib.lib().stereo_ncc_best_disp, _p(_f(ncc)), C.c_int(H), C.c_int(W), C.c_int(D), C.c_int(layout),
          _p(_f(np.asarray(disparities, np.float64).reshape(-1))), _p(best))
    return best


def globalstereo_unary(im0, im1, P2, d_min, d_step, col_thresh, assignment):
    im0, im1 = _f(im0), _f(im1)
    if im0.ndim == 2:
        im0, im1 = im0[:, :, None], im1[:, :, None]
    H, W, Cn = im0.shape
    U = np.zeros(H * W)
    _call(_lib.lib().stereo_globalstereo_unary, _p(_f(im0)), _p(_f(im1)), C.c_int(H), C.c_int(W), C.c_int(Cn),
          _p(_f(np.asarray(P2, np.float64))), C.c_double(d_min), C.c_double(d_step), C.c_double(col_thresh),
          _p(_f(assignment)), _p(U))
    return U


def segpln_wta(images, P, disps, col_thresh=30.0, window=2, min_corr=0.07):
    """dispmap_globalstereo.m:72-113 on the device (stereo_segpln_wta): the winner-takes-all disparity map behind
    the SegPln proposals.  images: list of (H, W, C) arrays, the first one the reference image; P (3, 4, n);
    disps: the class's self.disps (descending).  -> (H, W) array."""
    ims = [np.asarray(im, np.float64) for im in images]
    ims = [im[:, :, None] if im.ndim == 2 else im for im in ims]
    H, W, Cn = ims[0].shape
    if any(im.shape != ims[0].shape for im in ims):
        raise StereoHipError("segpln_wta: every image must have the reference image's H x W x C")
    if np.asarray(P).size != 12 * len(ims):
        raise StereoHipError("segpln_wta: P must be 3 x 4 x (number of images)")
    stack = np.concatenate([_f(im).reshape(-1, order="F") for im in ims])
    Pm = _f(np.asarray(P, np.float64).reshape(3, 4, len(ims)))
    d = _f(np.asarray(disps, np.float64).reshape(-1))
    out = np.zeros((H, W), order="F")
    _call(_lib.lib().stereo_segpln_wta, _p(stack), C.c_int(len(ims)), C.c_int(H), C.c_int(W), C.c_int(Cn), _p(Pm), _p(d),
          C.c_int(d.shape[0]), C.c_double(float(col_thresh)), C.c_int(int(window)), C.c_double(float(min_corr)), _p(out))
    return out


def segpln_planes(wta, segments, seed=0, rt=0.1, max_samples=500, want_proposal=True):
    """One SegPln proposal (dispmap_globalstereo.m:140-197) over a caller-supplied segmentation (labels 1 .. S,
    0 = no segment), LO-RANSAC + least-squares plane per segment on the device (stereo_segpln_planes).
    -> (proposal 4 x N Fortran order -- None without want_proposal: the planes alone --, planes S x 3, inlier counts S)."""
    wta = _f(np.asarray(wta, np.float64))
    H, W = wta.shape
    seg = np.asfortranarray(np.asarray(segments).astype(np.int32))
    if seg.shape != (H, W):
        raise StereoHipError("segpln_planes: segments must have the image's shape")
    S = int(seg.max()) if seg.size else 0
    prop = np.zeros((4, H * W), order="F") if want_proposal else None
    planes = np.zeros((3, max(S, 1)), order="F")
    ninl = np.zeros(max(S, 1), np.int32)
    _call(_lib.lib().stereo_segpln_planes, _p(wta), _p(seg, C.c_int32), C.c_int(H), C.c_int(W), C.c_double(float(rt)),
          C.c_uint64(int(seed) & 0xFFFFFFFFFFFFFFFF), C.c_int(int(max_samples)), _p(prop) if want_proposal else None, C.c_int(S), _p(planes),
          _p(ninl, C.c_int32))
    return prop, planes[:, :S].T.copy(), ninl[:S].copy()


def segpln_planes_batch(wta, segment_maps, seeds, rt=0.1, max_samples=500, want_proposal=True):
    """segpln_planes for several maps in one call (stereo_segpln_planes_batch): the maps run side by side on the device,
    the results are those of one call per map, bit for bit.  segment_maps: a sequence of H x W label arrays (or
    H x W x M); seeds: one per map.  -> list of (proposal, planes, inlier counts), as segpln_planes returns them."""
    wta = _f(np.asarray(wta, np.float64))
    H, W = wta.shape
    if isinstance(segment_maps, np.ndarray) and segment_maps.ndim == 3:
        segment_maps = [segment_maps[:, :, b] for b in range(segment_maps.shape[2])]
    def labels(seg):   # int32 or uint32, column major: passed as it is; anything else: one copy
        seg = np.asarray(seg)
        if seg.dtype in (np.int32, np.uint32) and seg.flags["F_CONTIGUOUS"]:
            return seg.view(np.int32)
        return np.array(seg, dtype=np.int32, order="F")
    segs = [labels(seg) for seg in segment_maps]
    M = len(segs)
    seeds = [int(v) & 0xFFFFFFFFFFFFFFFF for v in seeds]
    if M < 1 or len(seeds) != M or any(seg.shape != (H, W) for seg in segs):
        raise StereoHipError("segpln_planes_batch: one seed and one H x W label array per map")
    S = [int(seg.max()) if seg.size else 0 for seg in segs]
    props = [np.zeros((4, H * W), order="F") if want_proposal else None for _ in range(M)]
    planes = [np.zeros((3, max(s, 1)), order="F") for s in S]
    ninl = [np.zeros(max(s, 1), np.int32) for s in S]
    PD, PI = C.POINTER(C.c_double), C.POINTER(C.c_int32)
    seg_p = (PI * M)(*[_p(seg, C.c_int32) for seg in segs])
    prop_p = (PD * M)(*[(_p(a) if a is not None else None) for a in props])
    plane_p = (PD * M)(*[_p(a) for a in planes])
    ninl_p = (PI * M)(*[_p(a, C.c_int32) for a in ninl])
    _call(_lib.lib().stereo_segpln_planes_batch, _p(wta), seg_p, C.c_int(M), C.c_int(H), C.c_int(W), C.c_double(float(rt)),
          (C.c_uint64 * M)(*seeds), C.c_int(int(max_samples)), prop_p, (C.c_int * M)(*S), plane_p, ninl_p)
    return [(props[m], planes[m][:, :S[m]].T.copy(), ninl[m][:S[m]].copy()) for m in range(M)]
