"""Device-resident fusion moves (``stereo_fusion_*``): the state of one dispmap object --
connectivity, weights, current assignment, the unary source -- stays in HBM; a binary
fusion move uploads the proposal and returns four scalars."""
import ctypes as C

import numpy as np

from . import _lib
from ._lib import StereoHipError  # noqa: F401


def _f(a):
    return np.asfortranarray(a, dtype=np.float64)


def _p(a, t=C.c_double):
    return a.ctypes.data_as(C.POINTER(t)) if a is not None else None


class PlaneProposal:
    """A proposal given by a table of planes instead of a 4 x N array: one plane for every pixel
    (what dispmap_ncc.generate_new_plane_RANSAC and the fronto-parallel proposals of example_ncc.m
    are, before their repmat) or one plane per image segment (`segments`: N ids, pixel order
    col * H + row; the SegPln proposals of dispmap_globalstereo.m:154-192).  The device-resident
    moves build the 4 x N array in HBM; expand() gives the reference-shaped array."""

    def __init__(self, planes, segments=None):
        self.planes = np.asfortranarray(np.asarray(planes, np.float64).reshape(4, -1))
        self.segments = None if segments is None else np.ascontiguousarray(np.asarray(segments).reshape(-1), dtype=np.int32)
        if self.segments is None and self.planes.shape[1] != 1:
            raise StereoHipError("more than one plane needs segment ids")

    def expand(self, N):
        idx = np.zeros(N, np.int64) if self.segments is None else self.segments.astype(np.int64)
        return np.asfortranarray(self.planes[:, idx])


class FusionContext:
    def __init__(self, H, W, kernel, tol, conn0, weights, d_min=0.0, d_step=0.0):
        conn = np.asfortranarray(np.asarray(conn0), dtype=np.uint32)
        if conn.ndim != 2 or conn.shape[0] != 2:
            raise StereoHipError("connectivity must be 2 x E")
        w = _f(np.asarray(weights, np.float64).reshape(-1))
        self.N, self.E = H * W, conn.shape[1]
        self._h = C.c_void_p()
        err = _lib.errbuf()
        rc = _lib.lib().stereo_fusion_create(C.c_int(H), C.c_int(W), C.c_int(int(kernel)), C.c_double(tol),
                                            C.c_int64(self.E), _p(conn, C.c_uint32), _p(w), C.c_double(d_min),
                                            C.c_double(d_step), C.byref(self._h), err, C.c_size_t(len(err)))
        _lib.check(rc, err)

    def __del__(self):
        h = getattr(self, "_h", None)
        if h:
            _lib.lib().stereo_fusion_destroy(h)
            self._h = None

    def _call(self, fn, *args):
        err = _lib.errbuf()
        _lib.check(fn(self._h, *args, err, C.c_size_t(len(err))), err)

    def unary_ncc(self, ncc, disparities, unary_weight):
        ncc, d = _f(ncc), _f(np.asarray(disparities, np.float64).reshape(-1))
        self._call(_lib.lib().stereo_fusion_unary_ncc, _p(ncc), C.c_int(d.shape[0]), _p(d), C.c_double(unary_weight))

    def unary_globalstereo(self, im0, im1, P2, col_thresh):
        im0, im1 = _f(im0), _f(im1)
        Cn = 1 if im0.ndim == 2 else im0.shape[2]
        self._call(_lib.lib().stereo_fusion_unary_globalstereo, _p(im0), _p(im1), C.c_int(Cn),
                   _p(_f(np.asarray(P2, np.float64))), C.c_double(col_thresh))

    def set_assignment(self, assignment):
        e = C.c_double()
        self._call(_lib.lib().stereo_fusion_set_assignment, _p(_f(assignment)), C.byref(e))
        return e.value

    def get_assignment(self):
        a = np.zeros((4, self.N), order="F")
        e = C.c_double()
        self._call(_lib.lib().stereo_fusion_get_assignment, _p(a), C.byref(e))
        return a, e.value

    def binary(self, proposal, improve=False):
        """-> (stored energy after the move, rd energy, rd lower bound, num_unlabelled)"""
        e, re_, lb, nu = C.c_double(), C.c_double(), C.c_double(), C.c_double()
        self._call(_lib.lib().stereo_fusion_binary, _p(_f(proposal)), C.c_int(int(bool(improve))), C.byref(e),
                   C.byref(re_), C.byref(lb), C.byref(nu))
        return e.value, re_.value, lb.value, nu.value

    def binary_planes(self, proposal, improve=False):
        """binary() with a PlaneProposal: the 4 x N proposal is built on the device."""
        e, re_, lb, nu = C.c_double(), C.c_double(), C.c_double(), C.c_double()
        seg = proposal.segments
        self._call(_lib.lib().stereo_fusion_binary_planes, _p(proposal.planes), C.c_int(proposal.planes.shape[1]),
                   _p(seg, C.c_int32) if seg is not None else None, C.c_int(int(bool(improve))), C.byref(e),
                   C.byref(re_), C.byref(lb), C.byref(nu))
        return e.value, re_.value, lb.value, nu.value

    def fit_plane(self, x, y, r):
        """dispmap_ncc.m:48-92 on the device -> (plane [a b 1 d], number of pixels inside the radius)"""
        pl = np.zeros(4)
        n = C.c_double()
        self._call(_lib.lib().stereo_fusion_fit_plane, C.c_double(float(x)), C.c_double(float(y)), C.c_double(float(r)),
                   _p(pl), C.byref(n))
        return pl, int(n.value)

    def fuse_until_convergence(self, ids, maxiter, proposals=None, planes=None, improve=False):
        """The whole revisit schedule of dispmap_super.m:85-152 in one native call.  proposals: list of
        4 x N arrays, or planes: 4 x n table of single planes.  -> (energies E as the reference keeps them)"""
        ids = np.ascontiguousarray(ids, np.int64).reshape(-1)
        if proposals is not None:
            stack = np.asfortranarray(np.stack([np.asarray(P, np.float64) for P in proposals], axis=2))  # 4 x N x n
            n, pp, tp = stack.shape[2], _p(stack), None
        else:
            tab = np.asfortranarray(np.asarray(planes, np.float64).reshape(4, -1))
            n, pp, tp = tab.shape[1], None, _p(tab)
        cap = int(maxiter) + 2
        E = np.zeros(cap)
        nE = C.c_double()
        self._call(_lib.lib().stereo_fusion_fuse_until_convergence, pp, tp, C.c_int(n), _p(ids, C.c_int64),
                   C.c_int64(ids.size), C.c_int64(int(maxiter)), C.c_int(int(bool(improve))), C.byref(nE), _p(E),
                   C.c_int64(cap))
        return E[:int(nE.value)]

    def fit_planes(self, xs, ys, r):
        """fit_plane for many centres in one launch -> (planes 4 x n, pixel counts n)"""
        xs = np.ascontiguousarray(xs, np.float64).reshape(-1)
        ys = np.ascontiguousarray(ys, np.float64).reshape(-1)
        if xs.shape != ys.shape or xs.size < 1:
            raise StereoHipError("fit_planes: xs and ys must have the same, non-zero length")
        pl = np.zeros((4, xs.size), order="F")
        cnt = np.zeros(xs.size)
        self._call(_lib.lib().stereo_fusion_fit_planes, _p(xs), _p(ys), C.c_int(int(xs.size)), C.c_double(float(r)), _p(pl), _p(cnt))
        return pl, cnt.astype(np.int64)

    def simultaneous_planes(self, planes, maxiter=1000, max_relgap=0.0):
        """simultaneous() with K single-plane proposals (4 x K) built on the device."""
        planes = np.asfortranarray(np.asarray(planes, np.float64).reshape(4, -1))
        e, te, lb, it = C.c_double(), C.c_double(), C.c_double(), C.c_double()
        self._call(_lib.lib().stereo_fusion_simultaneous_planes, _p(planes), C.c_int(planes.shape[1]), C.c_double(maxiter),
                   C.c_double(max_relgap), C.byref(e), C.byref(te), C.byref(lb), C.byref(it))
        return e.value, te.value, lb.value, it.value

    def simultaneous(self, proposals, maxiter=1000, max_relgap=0.0):
        """proposals: list of 4 x N plane arrays (the current assignment is appended on the device).
        -> (stored energy afterwards, trws energy, lower bound, iterations)"""
        stack = np.asfortranarray(np.stack([np.asarray(P, np.float64) for P in proposals], axis=2))  # 4 x N x K
        e, te, lb, it = C.c_double(), C.c_double(), C.c_double(), C.c_double()
        self._call(_lib.lib().stereo_fusion_simultaneous, _p(stack), C.c_int(len(proposals)), C.c_double(maxiter),
                   C.c_double(max_relgap), C.byref(e), C.byref(te), C.byref(lb), C.byref(it))
        return e.value, te.value, lb.value, it.value
