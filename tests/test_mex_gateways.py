"""The mex gateway sources (mex/*.cpp) compiled and CALLED: this image has no MATLAB, so
tests/mexhost holds a small in-memory host for the subset of the documented MATLAB C API the
gateways use.  CPU part: they compile (-Wall -Werror) and their argument checks raise the
reference's messages (cpp/rd_mex.cpp:20-21, cpp/trws_mex.cpp:152-153) before anything touches a
GPU.  GPU part: a call through mexFunction gives exactly what the ctypes binding gives."""
import os
import sys

import numpy as np
import pytest

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "mexhost"))

from helpers import grid_conn, trws_problem


@pytest.mark.parametrize("name", ["rd_mex", "trws_mex", "fusion_mex", "vgg_segment_ms", "vgg_segment_gb"])
def test_gateway_compiles(name):
    import host
    assert os.path.exists(host.build(name))


def test_rd_gateway_argument_checks():
    import host
    g = host.Gateway("rd_mex")
    with pytest.raises(host.MexError, match="nrhs == 7"):
        g.call(4, np.zeros(3), np.zeros(3))
    conn = np.zeros((2, 2), np.uint32)
    ok = [np.zeros((3, 1)), np.zeros((3, 1)), np.zeros((1, 2)), np.zeros((1, 2)), np.zeros((1, 2)), np.zeros((1, 2)), conn]
    with pytest.raises(host.MexError, match="nlhs == 4"):
        g.call(3, *ok)
    with pytest.raises(host.MexError, match="uint32"):
        g.call(4, *(ok[:6] + [conn.astype(np.float64)]))
    with pytest.raises(host.MexError, match="1 x E"):
        g.call(4, *(ok[:2] + [np.zeros((2, 2))] + ok[3:]))


def test_trws_gateway_argument_checks():
    import host
    g = host.Gateway("trws_mex")
    with pytest.raises(host.MexError, match="nrhs == 8"):
        g.call(4, np.int32(1))
    K, N, E = 3, 4, 2
    args = [np.int32(1), np.zeros((K, N)), np.zeros((2, E), np.uint32), np.zeros((K, E)), np.zeros((K, E)), np.zeros((E, 1)),
            2.0, {"maxiter": 5.0}]
    with pytest.raises(host.MexError, match="int32"):
        g.call(4, *([1.0] + args[1:]))
    with pytest.raises(host.MexError, match="unary.M == q.M"):
        g.call(4, *(args[:3] + [np.zeros((K + 1, E))] + args[4:]))


def test_segmenter_gateways_argument_checks():
    """imrender/vgg/vgg_segment_ms.cxx:21-27, vgg_segment_gb.cxx:24-30: the reference's messages."""
    import host
    for name in ("vgg_segment_ms", "vgg_segment_gb"):
        g = host.Gateway(name)
        with pytest.raises(host.MexError, match="Unexpected number of input arguments."):
            g.call(1, np.zeros((4, 5, 3), np.uint8), 4.0)
        with pytest.raises(host.MexError, match="Unexpected number of output arguments."):
            g.call(3, np.zeros((4, 5, 3), np.uint8), 4.0, 5.0, 0.0)
        with pytest.raises(host.MexError, match="A must be an HxWx3 uint8 array."):
            g.call(1, np.zeros((4, 5, 3)), 4.0, 5.0, 0.0)


@pytest.mark.gpu
def test_segmenter_gateways_equal_the_binding(hip):
    import host
    from stereo_amd import segment as S
    im = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "teddy_crop.npz"))
    key = [k for k in im.files if im[k].ndim == 3 and im[k].shape[2] == 3][0]
    A = S.to_uint8(im[key])
    ms = host.Gateway("vgg_segment_ms").call(1, A, 4.0, 5.0, 0.0)[0]
    assert ms.dtype == np.uint32 and np.array_equal(ms, S.vgg_segment_ms(A, 4, 5, 0))
    gb = host.Gateway("vgg_segment_gb").call(1, A, 0.0, 300.0, 30.0, 1.0)[0]
    assert np.array_equal(gb, S.vgg_segment_gb(A, 0, 300, 30, 1))


@pytest.mark.gpu
def test_trws_gateway_equals_the_binding(hip):
    import host
    from stereo_amd.trws import trws
    H, W, K = 7, 9, 6
    p = trws_problem(211, H, W, K, kind="general")
    g = host.Gateway("trws_mex")
    for opts in ([{"maxiter": 4.0, "max_relgap": -1.0}],):   # the options struct of trws.m (nrhs is exactly 8, trws_mex.cpp:152)
        lab, en, lb, it = g.call(4, np.int32(1), p["unary"].T, p["conn"].T.astype(np.uint32), p["q"].T, p["qprim"].T,
                                 p["alphas"].reshape(-1, 1), 2.0, *opts)
        lab2, en2, lb2, it2 = trws(1, p["unary"].T, p["conn"].T + 1, p["q"].T, p["qprim"].T, p["alphas"], 2.0, dict(maxiter=4, max_relgap=-1))
        assert np.array_equal(lab.ravel(), np.asarray(lab2).ravel()) and en[0, 0] == en2 and lb[0, 0] == lb2 and it[0, 0] == it2 == 4
    with pytest.raises(host.MexError, match="Unsupported kernel"):   # trws_mex.cpp:128-131
        g.call(4, np.int32(3), p["unary"].T, p["conn"].T.astype(np.uint32), p["q"].T, p["qprim"].T, p["alphas"].reshape(-1, 1), 2.0, {})


@pytest.mark.gpu
def test_rd_gateway_equals_the_binding(hip):
    import host
    from stereo_amd.rd import rd
    rng = np.random.default_rng(5)
    H, W = 8, 9
    conn = grid_conn(H, W)
    N, E = H * W, conn.shape[0]
    U0, U1 = rng.normal(size=(N, 1)), rng.normal(size=(N, 1))
    E00, E01, E10, E11 = (rng.uniform(0, 2, size=(1, E)) for _ in range(4))
    g = host.Gateway("rd_mex")
    for improve in (False, True):
        lab, en, lb, unl = g.call(4, U0, U1, E00, E01, E10, E11, conn.T.astype(np.uint32), {"improve": float(improve)})
        lab2, en2, lb2, unl2 = rd(U0, U1, E00, E01, E10, E11, conn.T + 1, dict(improve=improve))
        assert np.array_equal(lab.ravel(), np.asarray(lab2).ravel()) and en[0, 0] == en2 and lb[0, 0] == lb2 and unl[0, 0] == unl2


@pytest.mark.gpu
def test_c_abi_from_a_plain_c_host(hip, tmp_path):
    """tools/abi_smoke.c: gcc, no Python, no torch -- the system HIP runtime, as under MATLAB."""
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = str(tmp_path / "abi_smoke")
    subprocess.run(["gcc", "-I" + os.path.join(root, "include"), os.path.join(root, "tools", "abi_smoke.c"), "-L" + os.path.join(root, "stereo_amd"),
                    "-lstereo_hip", "-Wl,-rpath," + os.path.join(root, "stereo_amd"), "-o", exe], check=True)
    r = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "stereo_rd rc 0 energy 2 bound 2 unlabelled 0" in r.stdout and "stereo_trws rc 0 energy 3 bound 3 iterations 5" in r.stdout


@pytest.mark.gpu
def test_fusion_gateway_equals_the_binding(hip):
    """fusion_mex (the optional gateway of the device-resident fusion context): the same sequence of
    commands through mexFunction and through the ctypes binding, equal energies and assignments."""
    import host
    from oracle import terms as ot
    from stereo_amd.fusion import FusionContext
    from stereo_amd import terms as T
    H, W, D = 9, 11, 6
    N = H * W
    conn = T.construct_neighborhood(H, W)
    w = np.linspace(0.5, 1.5, conn.shape[1])
    rng = np.random.default_rng(3)
    ncc = np.asfortranarray(rng.uniform(-1, 1, size=(H, W, D)))
    disps = np.arange(float(D))
    ctx = FusionContext(H, W, 1, 2.0, conn, w)
    ctx.unary_ncc(ncc, disps, 3.0)
    g = host.Gateway("fusion_mex")
    (h,) = g.call(1, "create", float(H), float(W), np.int32(1), 2.0, conn.astype(np.uint32), w.reshape(1, -1), 0.0, 0.0)
    assert h.dtype == np.uint64 and h[0, 0] != 0
    with pytest.raises(host.MexError, match="Overload unary_cost"):   # dispmap_super.m:200-203
        g.call(1, "set_assignment", h, ot.fronto_parallel(1.0, N))
    g.call(0, "unary_ncc", h, ncc, disps.reshape(1, -1), 3.0)
    a = ot.fronto_parallel(1.0, N)
    (e,) = g.call(1, "set_assignment", h, a)
    assert e[0, 0] == ctx.set_assignment(a)
    for d, improve in ((3.0, 0.0), (4.0, 1.0)):
        prop = ot.fronto_parallel(d, N)
        out = g.call(4, "binary", h, prop, improve)
        ref = ctx.binary(prop, improve=bool(improve))
        assert tuple(float(x[0, 0]) for x in out) == tuple(ref)
    props = [ot.fronto_parallel(d, N) for d in (0.0, 2.0, 5.0)]
    out = g.call(4, "simultaneous", h, np.stack(props, axis=2), 6.0, -1.0)
    ref = ctx.simultaneous(props, maxiter=6, max_relgap=-1.0)
    assert tuple(float(x[0, 0]) for x in out) == tuple(ref)
    got, e2 = g.call(2, "get_assignment", h, float(N))
    want, e3 = ctx.get_assignment()
    assert np.array_equal(got, want) and e2[0, 0] == e3
    with pytest.raises(host.MexError, match="unknown command"):
        g.call(0, "nonsense", h)
    g.call(0, "destroy", h)
