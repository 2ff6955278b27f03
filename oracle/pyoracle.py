"""TEST INFRASTRUCTURE ONLY -- ctypes front-end of the CPU oracle.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline``
leg may import this module.  The product package ``stereo_amd`` never does.

Two libraries sit behind it:

* ``oracle/liboracle.so``   -- this repo's own restatement (plain C, ``make -C oracle oracle``)
* ``oracle/_ref/*.so``      -- the reference's own QPBO library and TRW-S type
  classes, compiled from ``/root/reference`` where that tree exists (build
  container); prebuilt files travel to the GPU box.  Optional: ``have_ref_*``.
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))

_dp = C.POINTER(C.c_double)
_u32p = C.POINTER(C.c_uint32)
_i64p = C.POINTER(C.c_int64)
_i32p = C.POINTER(C.c_int32)


def _load(path):
    return C.CDLL(path) if os.path.exists(path) else None


_lib = None
_ref_qpbo = None
_ref_types = None
_ref_segment = None


def lib():
    global _lib
    if _lib is None:
        p = os.path.join(_HERE, "liboracle.so")
        if not os.path.exists(p):
            raise RuntimeError("oracle/liboracle.so missing: run `make -C oracle oracle` "
                               "(or __graft_entry__.build())")
        _lib = C.CDLL(p)
        _lib.oracle_update_message.restype = C.c_double
        _lib.oracle_update_message_envelope.restype = C.c_double
    return _lib


def ref_qpbo():
    global _ref_qpbo
    if _ref_qpbo is None:
        _ref_qpbo = _load(os.path.join(_HERE, "_ref", "libref_qpbo.so")) or False
    return _ref_qpbo or None


def ref_types():
    global _ref_types
    if _ref_types is None:
        _ref_types = _load(os.path.join(_HERE, "_ref", "libref_trws_types.so")) or False
        if _ref_types:
            _ref_types.ref_update_message.restype = C.c_double
            _ref_types.ref_vec_min.restype = C.c_double
    return _ref_types or None


def ref_segment():
    global _ref_segment
    if _ref_segment is None:
        _ref_segment = _load(os.path.join(_HERE, "_ref", "libref_segment_ms.so")) or False
    return _ref_segment or None


def have_ref_segment():
    return ref_segment() is not None


_ref_segment_gb = None


def ref_segment_gb_lib():
    global _ref_segment_gb
    if _ref_segment_gb is None:
        _ref_segment_gb = _load(os.path.join(_HERE, "_ref", "libref_segment_gb.so")) or False
    return _ref_segment_gb or None


def have_ref_qpbo():
    return ref_qpbo() is not None


def have_ref_types():
    return ref_types() is not None


def _d(a):
    a = np.ascontiguousarray(a, dtype=np.float64)
    return a, a.ctypes.data_as(_dp)


def _conn(conn):
    """conn: (E,2) zero-based pairs (row e = edge e) -> contiguous (E,2) uint32."""
    c = np.asarray(conn)
    if c.ndim != 2 or c.shape[1] != 2:
        raise ValueError("connectivity must be E x 2")
    c = np.ascontiguousarray(c, dtype=np.uint32)
    return c, c.ctypes.data_as(_u32p)


# --------------------------------------------------------------------- TRW-S

def trws(kernel, unary, conn, q, qprim, alphas, tol, maxiter=1000, max_relgap=0.0, mode=0,
         use_ref_types=False, want_trace=False, ordering=0):
    """unary (N,K), q/qprim (E,K) row-major == MATLAB K x N / K x E column-major.
    conn zero-based (E,2).  Returns labels (1-based float64 like the gateway),
    energy, lower bound, iterations [, trace]."""
    unary, pu = _d(unary)
    q, pq = _d(q)
    qprim, pqp = _d(qprim)
    alphas, pa = _d(alphas)
    c, pc = _conn(conn)
    N, K = unary.shape
    E = c.shape[0]
    assert q.shape == (E, K) and qprim.shape == (E, K) and alphas.shape == (E,)
    lab = np.zeros(N)
    en, lb, it = C.c_double(), C.c_double(), C.c_double()
    msg_fn = col_fn = None
    if use_ref_types:
        r = ref_types()
        if r is None:
            raise RuntimeError("oracle/_ref/libref_trws_types.so not available")
        msg_fn = C.cast(r.ref_update_message, C.c_void_p)
        col_fn = C.cast(r.ref_add_column, C.c_void_p)
    trace = np.zeros((int(maxiter), 3)) if want_trace else None
    lib().oracle_set_ordering(C.c_int(int(ordering)))   # 0: SetAutomaticOrdering, 1: node index order
    rc = lib().oracle_trws(C.c_int(int(kernel)), pu, pc, pq, pqp, pa, C.c_double(tol),
                           C.c_double(maxiter), C.c_double(max_relgap), C.c_int(K),
                           C.c_int64(N), C.c_int64(E), C.c_int(mode), msg_fn, col_fn,
                           lab.ctypes.data_as(_dp), C.byref(en), C.byref(lb), C.byref(it),
                           trace.ctypes.data_as(_dp) if want_trace else None)
    lib().oracle_set_ordering(C.c_int(0))
    if rc:
        raise RuntimeError("oracle_trws failed rc=%d" % rc)
    out = (lab, en.value, lb.value, it.value)
    if want_trace:
        out = out + (trace[: int(it.value)],)
    return out


def trws_structure(N, conn):
    c, pc = _conn(conn)
    E = c.shape[0]
    rank = np.zeros(N, np.int64)
    tail = np.zeros(E, np.int64)
    head = np.zeros(E, np.int64)
    dirn = np.zeros(E, np.int32)
    fptr = np.zeros(N + 1, np.int64)
    bptr = np.zeros(N + 1, np.int64)
    fidx = np.zeros(E, np.int64)
    bidx = np.zeros(E, np.int64)
    rc = lib().oracle_trws_structure(C.c_int64(N), C.c_int64(E), pc,
                                     rank.ctypes.data_as(_i64p), tail.ctypes.data_as(_i64p),
                                     head.ctypes.data_as(_i64p), dirn.ctypes.data_as(_i32p),
                                     fptr.ctypes.data_as(_i64p), fidx.ctypes.data_as(_i64p),
                                     bptr.ctypes.data_as(_i64p), bidx.ctypes.data_as(_i64p))
    if rc:
        raise RuntimeError("oracle_trws_structure failed rc=%d" % rc)
    return dict(rank=rank, tail=tail, head=head, dir=dirn, fwd_ptr=fptr, fwd_idx=fidx,
                bwd_ptr=bptr, bwd_idx=bidx)


def _message(fn, kernel, Di, gamma, msg, q, qprim, alpha, lam, dirn, mdir):
    Di, pD = _d(Di)
    m = np.array(msg, dtype=np.float64, copy=True)
    q, pq = _d(q)
    qprim, pqp = _d(qprim)
    K = Di.shape[0]
    v = fn(C.c_int(kernel), C.c_int(K), pD, C.c_double(gamma), m.ctypes.data_as(_dp), pq, pqp,
           C.c_double(alpha), C.c_double(lam), C.c_int(dirn), C.c_int(mdir))
    return m, v


def update_message(kernel, Di, gamma, msg, q, qprim, alpha, lam, dirn, mdir, impl="brute"):
    """impl: 'brute' | 'envelope' (restatements) | 'ref' (reference type classes)."""
    if impl == "brute":
        fn = lib().oracle_update_message
    elif impl == "envelope":
        fn = lib().oracle_update_message_envelope
    elif impl == "ref":
        fn = ref_types().ref_update_message
    else:
        raise ValueError(impl)
    return _message(fn, kernel, Di, gamma, msg, q, qprim, alpha, lam, dirn, mdir)


def add_column(kernel, q, qprim, alpha, lam, ksource, dest, dirn, mdir, impl="brute"):
    q, pq = _d(q)
    qprim, pqp = _d(qprim)
    d = np.array(dest, dtype=np.float64, copy=True)
    fn = lib().oracle_add_column if impl == "brute" else ref_types().ref_add_column
    fn(C.c_int(kernel), C.c_int(q.shape[0]), pq, pqp, C.c_double(alpha), C.c_double(lam),
       C.c_int(int(ksource)), d.ctypes.data_as(_dp), C.c_int(dirn), C.c_int(mdir))
    return d


# --------------------------------------------------------------------- QPBO

def ref_rd(U0, U1, E00, E01, E10, E11, conn, improve=False, stage=0, seed=None):
    """The REFERENCE QPBO library behind the restated rd_mex gateway (oracle/_ref).
    conn (E,2) zero based.  stage 1 stops after Solve() (strong persistency only)."""
    L = ref_qpbo()
    if L is None:
        raise RuntimeError("oracle/_ref/libref_qpbo.so not available")
    U0, p0 = _d(U0); U1, p1 = _d(U1)
    E00, a = _d(E00); E01, b = _d(E01); E10, c_ = _d(E10); E11, d = _d(E11)
    cc, pc = _conn(conn)
    N, E = U0.shape[0], cc.shape[0]
    if seed is not None:
        L.ref_srand(C.c_uint(seed))
    lab = np.zeros(N)
    en, lb, nu = C.c_double(), C.c_double(), C.c_double()
    rc = L.ref_rd_stage(p0, p1, a, b, c_, d, pc, C.c_int64(N), C.c_int64(E), C.c_int(int(improve)),
                        C.c_int(stage), lab.ctypes.data_as(_dp), C.byref(en), C.byref(lb), C.byref(nu))
    if rc:
        raise RuntimeError("ref_rd failed")
    return lab, en.value, lb.value, nu.value


def rd(U0, U1, E00, E01, E10, E11, conn, improve=False, stage=0, seed=None):
    """This repo's restatement (oracle/qpbo_oracle.c); same interface as ref_rd."""
    U0, p0 = _d(U0); U1, p1 = _d(U1)
    E00, a = _d(E00); E01, b = _d(E01); E10, c_ = _d(E10); E11, d = _d(E11)
    cc, pc = _conn(conn)
    N, E = U0.shape[0], cc.shape[0]
    if seed is not None:
        C.CDLL(None).srand(C.c_uint(seed))
    lab = np.zeros(N)
    en, lb, nu = C.c_double(), C.c_double(), C.c_double()
    rc = lib().oracle_rd(p0, p1, a, b, c_, d, pc, C.c_int64(N), C.c_int64(E), C.c_int(int(improve)),
                         C.c_int(stage), lab.ctypes.data_as(_dp), C.byref(en), C.byref(lb), C.byref(nu))
    if rc:
        raise RuntimeError("oracle_rd failed rc=%d" % rc)
    return lab, en.value, lb.value, nu.value


# --------------------------------------------------------------------- segmentation

def ref_segment_ms(image, h_s=4, h_r=5.0, min_sz=0):
    """The REFERENCE mean-shift segmenter (imrender/vgg/seg_ms through oracle/_ref) behind the
    restated vgg_segment_ms gateway: image H x W x 3 uint8 -> H x W uint32 segment ids from 1
    (dispmap_globalstereo.m:391-392 with ojw_default_options.m:68 seg_params = [4 5 0])."""
    L = ref_segment()
    if L is None:
        raise RuntimeError("oracle/_ref/libref_segment_ms.so not available")
    im = np.asarray(image)
    if im.dtype != np.uint8 or im.ndim != 3 or im.shape[2] != 3:
        raise ValueError("A must be an HxWx3 uint8 array.")     # vgg_segment_ms.cxx:26-27
    H, W, _ = im.shape
    A = np.asfortranarray(im)                                    # MATLAB's H x W x 3 column-major
    out = np.zeros((H, W), dtype=np.uint32, order="F")
    err = C.create_string_buffer(512)
    rc = L.ref_segment_ms(A.ctypes.data_as(C.POINTER(C.c_uint8)), C.c_int(H), C.c_int(W),
                          C.c_int(int(h_s)), C.c_float(float(h_r)), C.c_int(int(min_sz)),
                          out.ctypes.data_as(_u32p), err, C.c_size_t(512))
    if rc:
        raise RuntimeError("ref_segment_ms: %s" % err.value.decode())
    return np.ascontiguousarray(out)


# What msImageProcessor::speedThreshold holds when the reference's gateway reads it uninitialised (oracle/segment_oracle.c)
MS_SPEED_THRESHOLD = float(np.float32(4.5912142885138307e-41))


def rgb_to_luv(image):
    """oracle/segment_oracle.c: msImageProcessor.cpp:835-875 on an H x W x 3 uint8 image -> H*W x 3 float32, row-major pixels."""
    im = np.ascontiguousarray(image, dtype=np.uint8)
    H, W, _ = im.shape
    luv = np.zeros((H * W, 3), np.float32)
    lib().oracle_rgb_to_luv(im.ctypes.data_as(C.POINTER(C.c_uint8)), C.c_int64(H * W), luv.ctypes.data_as(C.POINTER(C.c_float)))
    return luv


def ms_filter(luv, H, W, h_s, h_r, speed_threshold=MS_SPEED_THRESHOLD, event_threshold=MS_SPEED_THRESHOLD):
    """oracle/segment_oracle.c: the reference's mean-shift filter (msImageProcessor.cpp:3803-4303) -> (filtered image
    H*W x 3 float32, event flags H*W uint8).  speed_threshold 0: every pixel walks its own whole trajectory."""
    luv = np.ascontiguousarray(luv, np.float32)
    out = np.zeros_like(luv)
    events = np.zeros(H * W, np.uint8)
    rc = lib().oracle_ms_filter(luv.ctypes.data_as(C.POINTER(C.c_float)), C.c_int(H), C.c_int(W), C.c_int(int(h_s)), C.c_float(h_r),
                                C.c_float(speed_threshold), C.c_float(event_threshold), events.ctypes.data_as(C.POINTER(C.c_uint8)),
                                out.ctypes.data_as(C.POINTER(C.c_float)))
    if rc:
        raise MemoryError("oracle_ms_filter")
    return out, events


def ref_segment_gb(image, sigma, k, min_sz, compress=1):
    """The REFERENCE graph-based segmenter (imrender/vgg/seg_gb) behind the restated vgg_segment_gb
    gateway: image H x W x 3 uint8 -> H x W uint32 (dispmap_globalstereo.m:132)."""
    L = ref_segment_gb_lib()
    if L is None:
        raise RuntimeError("oracle/_ref/libref_segment_gb.so not available")
    im = np.asarray(image)
    if im.dtype != np.uint8 or im.ndim != 3 or im.shape[2] != 3:
        raise ValueError("A must be an HxWx3 uint8 array.")     # vgg_segment_gb.cxx:29-30
    H, W, _ = im.shape
    A = np.asfortranarray(im)
    out = np.zeros((H, W), dtype=np.uint32, order="F")
    L.ref_segment_gb(A.ctypes.data_as(C.POINTER(C.c_uint8)), C.c_int(H), C.c_int(W), C.c_float(float(sigma)),
                     C.c_float(float(k)), C.c_int(int(min_sz)), C.c_int(int(compress)), out.ctypes.data_as(_u32p))
    return np.ascontiguousarray(out)


def ref_segpln_segments(image):
    """The 14 segmentation maps of segpln (dispmap_globalstereo.m:121-134): mean shift at seven scales,
    then the graph-based segmenter at seven scales -> H x W x 14 uint32."""
    segment_params = np.array([1, 1.5, 10, 100.0])
    mults = [1, 2, 3, 4, 5, 6, 7, 3, 5, 8, 12, 24, 50, 100]
    maps = []
    for b, m in enumerate(mults):
        sp = segment_params * m
        if b < 7:
            maps.append(ref_segment_ms(image, sp[0], sp[1], sp[2]))
        else:
            maps.append(ref_segment_gb(image, 0, sp[3], sp[2], 1))
    return np.stack(maps, axis=2)
