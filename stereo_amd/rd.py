"""Mirror of the reference's ``rd.m`` wrapper (roof-duality binary fusion).

``rd(U0, U1, E00, E01, E10, E11, connectivity, options)`` -- ``U0``/``U1`` N x 1,
``E**`` 1 x E, ``connectivity`` 2 x E ONE based (rd.m:21 subtracts 1),
``options`` a dict with ``improve`` (default False, rd_mex.cpp:34).  Returns
``(solution, energy, lower_bound, num_unlabelled)`` with solution in {-1, 0, 1}.
"""
import ctypes as C

import numpy as np

from . import _lib
from ._lib import StereoHipError


def _v(a):
    return np.ascontiguousarray(np.asarray(a, dtype=np.float64).reshape(-1))


def _p(a, t=C.c_double):
    return a.ctypes.data_as(C.POINTER(t))


def rd(U0, U1, E00, E01, E10, E11, connectivity, options=None):
    options = dict(options or {})
    improve = bool(options.pop("improve", False))
    if options:
        raise StereoHipError("unknown option(s): %s" % ", ".join(sorted(options)))
    U0, U1, E00, E01, E10, E11 = map(_v, (U0, U1, E00, E01, E10, E11))
    c = np.asarray(connectivity)
    if c.ndim != 2 or c.shape[0] != 2:
        raise StereoHipError("connectivity must be 2 x E")       # rd_mex.cpp:49
    N, E = U0.shape[0], c.shape[1]
    if c.size:
        assert c.min() > 0                                          # rd.m:5
        assert c.max() <= N                                         # rd.m:6
    # rd_mex.cpp:36-48
    if U1.shape[0] != N:
        raise StereoHipError("U0 and U1 must have the same length")
    if not (E00.shape[0] == E01.shape[0] == E10.shape[0] == E11.shape[0] == E):
        raise StereoHipError("E00, E01, E10, E11 and connectivity must agree in length")
    conn = np.asfortranarray(c.astype(np.int64) - 1, dtype=np.uint32)
    lab = np.zeros(N)
    en, lb, nu = C.c_double(), C.c_double(), C.c_double()
    err = _lib.errbuf()
    rc = _lib.lib().stereo_rd(_p(U0), _p(U1), _p(E00), _p(E01), _p(E10), _p(E11),
                              _p(conn, C.c_uint32), C.c_int64(N), C.c_int64(E),
                              C.c_int(int(improve)), _p(lab), C.byref(en), C.byref(lb),
                              C.byref(nu), err, C.c_size_t(len(err)))
    _lib.check(rc, err)
    return lab, en.value, lb.value, nu.value


class RdPlan:
    """Device-resident roof-duality solver for one connectivity (stereo_rd_plan_*): repeated
    binary fusions on one image reuse the doubled-graph layout and all device buffers."""

    def __init__(self, N, connectivity0, grid=None):
        """grid = (H, W): the nodes are the pixels of an H x W image numbered col*H + row (a work
        partition hint, stereo_rd_plan_set_grid; results do not depend on it)."""
        c = np.asarray(connectivity0)
        if c.ndim != 2 or c.shape[0] != 2:
            raise StereoHipError("connectivity must be 2 x E")
        self._conn = np.asfortranarray(c, dtype=np.uint32)
        self.N, self.E = int(N), int(self._conn.shape[1])
        self._h = C.c_void_p()
        err = _lib.errbuf()
        L = _lib.lib()
        L.stereo_rd_plan_destroy.restype = None
        rc = L.stereo_rd_plan_create(C.c_int64(self.N), C.c_int64(self.E), _p(self._conn, C.c_uint32),
                                     C.byref(self._h), err, C.c_size_t(len(err)))
        _lib.check(rc, err)
        if grid is not None:
            rc = L.stereo_rd_plan_set_grid(self._h, C.c_int(int(grid[0])), C.c_int(int(grid[1])), err,
                                           C.c_size_t(len(err)))
            _lib.check(rc, err)

    def close(self):
        if self._h:
            _lib.lib().stereo_rd_plan_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def solve(self, U0, U1, E00, E01, E10, E11, improve=False):
        U0, U1, E00, E01, E10, E11 = map(_v, (U0, U1, E00, E01, E10, E11))
        assert U0.shape[0] == self.N and U1.shape[0] == self.N
        assert E00.shape[0] == E01.shape[0] == E10.shape[0] == E11.shape[0] == self.E
        lab = np.zeros(self.N)
        en, lb, nu = C.c_double(), C.c_double(), C.c_double()
        err = _lib.errbuf()
        rc = _lib.lib().stereo_rd_plan_solve(self._h, _p(U0), _p(U1), _p(E00), _p(E01), _p(E10), _p(E11),
                                             C.c_int(int(bool(improve))), _p(lab), C.byref(en), C.byref(lb),
                                             C.byref(nu), err, C.c_size_t(len(err)))
        _lib.check(rc, err)
        return lab, en.value, lb.value, nu.value
