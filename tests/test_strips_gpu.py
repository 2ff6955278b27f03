"""-m gpu: row-strip TRW-S (the multi-GPU decomposition) as LOGICAL strips on one GPU.

G plans, one per band of rows, run concurrently on one device and hand boundary messages, flags
and labels to each other exactly as they would across GPUs (the same stores, aimed at the
neighbour plan's arrays).  Bar: labels bit-identical to the single plan AND to the oracle;
energy / lower bound are per-strip partial sums added in strip order, so they agree to rounding
(1e-12 relative, stated in include/stereo_hip.h), and exactly when there is one strip."""
import numpy as np
import pytest

from helpers import trws_problem

pytestmark = pytest.mark.gpu

REL = 1e-12


def _close(a, b):
    return abs(a - b) <= REL * max(abs(a), abs(b), 1.0)


def _run_strips(p, kernel, K, H, W, G, tol, iters, shared_pos=None, wg=None):
    from stereo_amd.strips import make_strips
    s = make_strips(kernel, K, H, W, p["conn"].T, G, workgroups_per_strip=wg)
    if shared_pos is not None:
        s.upload(p["unary"].T, p["alphas"], tol, positions=shared_pos)
    else:
        s.upload(p["unary"].T, p["alphas"], tol, q=p["q"].T, qprim=p["qprim"].T)
    done, _ = s.iterate(iters, max_relgap=-1e300)
    assert done == iters
    r = s.result()
    path = s.path()
    infos = [pl.info() for pl in s.plans]
    s.close()
    return r, path, infos


CASES = [
    # seed, H, W, K, kernel, kind, integer, tol, iters
    (101, 60, 70, 16, 1, "general", False, 3.0, 4),   # pipelined kernel, per-edge positions
    (102, 24, 31, 12, 2, "general", False, 4.0, 4),   # quadratic kernel
    (103, 18, 22, 9, 1, "general", True, 2.0, 5),     # integer costs: exact ties, serial envelopes
    (104, 30, 26, 40, 1, "fronto", False, 5.0, 4),    # shared positions, windowed min-plus
    (105, 9, 40, 8, 1, "general", False, 2.0, 4),     # strips of 2-3 rows
]


@pytest.mark.parametrize("case", CASES, ids=[str(c[0]) for c in CASES])
@pytest.mark.parametrize("G", [2, 4])
def test_logical_strips_match_single_plan_and_oracle(case, G, hip, oracle):
    from stereo_amd.trws import TrwsPlan
    seed, H, W, K, kernel, kind, integer, tol, iters = case
    p = trws_problem(seed, H, W, K, kind=kind, integer=integer)
    lab_o, en_o, lb_o, it_o = oracle.trws(kernel, p["unary"], p["conn"], p["q"], p["qprim"], p["alphas"], tol,
                                          iters, -1e300, mode=1)
    pos = np.arange(K, dtype=np.float64) if kind == "fronto" else None
    one = TrwsPlan(kernel, K, H * W, p["conn"].T)
    if pos is not None:
        one.upload(p["unary"].T, p["alphas"], tol, positions=pos)
    else:
        one.upload(p["unary"].T, p["alphas"], tol, q=p["q"].T, qprim=p["qprim"].T)
    one.iterate(iters, max_relgap=-1e300)
    lab1, en1, lb1, _ = one.result()
    assert np.array_equal(lab1, lab_o) and en1 == en_o and lb1 == lb_o
    (lab, en, lb, it), path, infos = _run_strips(p, kernel, K, H, W, G, tol, iters, shared_pos=pos)
    assert path == one.path()
    assert it == it_o
    assert sum(i["own_nodes"] for i in infos) == H * W
    assert np.array_equal(lab, lab_o), "labels differ at %d nodes" % int((lab != lab_o).sum())
    assert _close(en, en_o) and _close(lb, lb_o), (en, en_o, lb, lb_o)


def test_one_strip_is_the_single_plan(hip, oracle):
    """nstrips = 1 through the strip entry points: the same bits as the plain plan, sums included."""
    p = trws_problem(106, 14, 15, 10, kind="general")
    lab_o, en_o, lb_o, _ = oracle.trws(1, p["unary"], p["conn"], p["q"], p["qprim"], p["alphas"], 2.0, 4, -1e300, mode=1)
    (lab, en, lb, _), _, _ = _run_strips(p, 1, 10, 14, 15, 1, 2.0, 4)
    assert np.array_equal(lab, lab_o) and en == en_o and lb == lb_o


@pytest.mark.parametrize("G", [2, 4])
def test_wide_kernel_strips_k256(G, hip, oracle):
    """K = 256 on the wide-label kernel (the 3000 x 2000 x 256 configuration's kernel): strips vs
    the oracle on a grid the oracle finishes in seconds, and vs the single plan on a larger one."""
    from stereo_amd.trws import TrwsPlan
    K = 256
    pos = np.arange(K, dtype=np.float64)
    H, W = 12, 14
    p = trws_problem(107, H, W, K, kind="fronto")
    lab_o, en_o, lb_o, _ = oracle.trws(1, p["unary"], p["conn"], p["q"], p["qprim"], p["alphas"], 8.0, 2, -1e300, mode=1)
    (lab, en, lb, _), path, _ = _run_strips(p, 1, K, H, W, G, 8.0, 2, shared_pos=pos)
    assert path == 3
    assert np.array_equal(lab, lab_o) and _close(en, en_o) and _close(lb, lb_o)
    H, W = 64, 72
    p = trws_problem(108, H, W, K, kind="fronto")
    one = TrwsPlan(1, K, H * W, p["conn"].T)
    one.upload(p["unary"].T, p["alphas"], 8.0, positions=pos)
    one.iterate(3, max_relgap=-1e300)
    lab1, en1, lb1, _ = one.result()
    (lab, en, lb, _), path, _ = _run_strips(p, 1, K, H, W, G, 8.0, 3, shared_pos=pos)
    assert path == 3
    assert np.array_equal(lab, lab1) and _close(en, en1) and _close(lb, lb1)
    # the truncated quadratic kernel on the same layout (round 4)
    H, W = 12, 14
    p = trws_problem(109, H, W, K, kind="fronto")
    lab_o, en_o, lb_o, _ = oracle.trws(2, p["unary"], p["conn"], p["q"], p["qprim"], p["alphas"], 64.0, 2, -1e300, mode=1)
    (lab, en, lb, _), path, _ = _run_strips(p, 2, K, H, W, G, 64.0, 2, shared_pos=pos)
    assert path == 3
    assert np.array_equal(lab, lab_o) and _close(en, en_o) and _close(lb, lb_o)


@pytest.mark.parametrize("G", [2, 3])
@pytest.mark.parametrize("integer", [False, True], ids=["real", "integer"])
def test_k128_kernel_strips_with_per_edge_positions(G, integer, hip, oracle):
    """64 < K <= 128 with per-edge positions (the 3D-label moves of config 5 with many proposals):
    the two-labels-per-lane pipelined kernel as strips, vs the oracle on a small grid and vs the
    single plan on a larger one.  Integer costs force exact ties through the serial envelopes."""
    from stereo_amd.trws import TrwsPlan
    K = 100
    H, W = 10, 12
    p = trws_problem(131 + integer, H, W, K, kind="general", integer=integer)
    lab_o, en_o, lb_o, _ = oracle.trws(1, p["unary"], p["conn"], p["q"], p["qprim"], p["alphas"], 6.0, 3, -1e300, mode=1)
    (lab, en, lb, _), path, _ = _run_strips(p, 1, K, H, W, G, 6.0, 3)
    assert path == 4
    assert np.array_equal(lab, lab_o) and _close(en, en_o) and _close(lb, lb_o)
    H, W = 48, 52
    p = trws_problem(133 + integer, H, W, K, kind="general", integer=integer)
    one = TrwsPlan(1, K, H * W, p["conn"].T)
    one.upload(p["unary"].T, p["alphas"], 6.0, q=p["q"].T, qprim=p["qprim"].T)
    one.iterate(3, max_relgap=-1e300)
    lab1, en1, lb1, _ = one.result()
    assert one.path() == 4
    (lab, en, lb, _), path, _ = _run_strips(p, 1, K, H, W, G, 6.0, 3)
    assert path == 4
    assert np.array_equal(lab, lab1) and _close(en, en1) and _close(lb, lb1)


def test_strips_with_fewer_workgroups_than_runs(hip, oracle):
    """Each strip has more rows than workgroups: runs wait for tickets inside a strip while the
    strips wait for each other across the boundary -- no deadlock, same labels."""
    H, W, K = 90, 8, 6
    p = trws_problem(109, H, W, K, kind="fronto")
    pos = np.arange(K, dtype=np.float64)
    lab_o, en_o, lb_o, _ = oracle.trws(1, p["unary"], p["conn"], p["q"], p["qprim"], p["alphas"], 2.0, 3, -1e300, mode=1)
    (lab, en, lb, _), _, infos = _run_strips(p, 1, K, H, W, 3, 2.0, 3, shared_pos=pos, wg=4)
    assert min(i["runs_forward"] for i in infos) > 4
    assert np.array_equal(lab, lab_o) and _close(en, en_o) and _close(lb, lb_o)


def test_unconnected_strip_refuses_to_iterate(hip):
    from stereo_amd.strips import _StripPlan, row_strip_owner, _conn_f
    p = trws_problem(110, 8, 9, 5, kind="general")
    owner = row_strip_owner(8, 9, 2)
    pl = _StripPlan(1, 5, 72, _conn_f(p["conn"].T), owner, 2, 0, 0, 0, None)
    pl.upload(p["unary"].T, p["alphas"], 2.0, q=p["q"].T, qprim=p["qprim"].T)
    with pytest.raises(hip.StereoHipError, match="not connected"):
        pl.issue()
    pl.close()


def test_new_inputs_restart_the_minimisation(hip, oracle):
    """upload() after iterate() implies a reset (the look-ahead forward sweep ran on the old
    inputs): the second problem's result is the oracle's from-scratch result."""
    from stereo_amd.trws import TrwsPlan
    p1 = trws_problem(111, 10, 12, 8, kind="general")
    p2 = trws_problem(112, 10, 12, 8, kind="general")
    plan = TrwsPlan(1, 8, 120, p1["conn"].T)
    plan.upload(p1["unary"].T, p1["alphas"], 2.0, q=p1["q"].T, qprim=p1["qprim"].T)
    plan.iterate(3, max_relgap=-1e300)
    plan.upload(p2["unary"].T, p2["alphas"], 2.0, q=p2["q"].T, qprim=p2["qprim"].T)
    plan.iterate(4, max_relgap=-1e300)
    lab, en, lb, it = plan.result()
    lab_o, en_o, lb_o, it_o = oracle.trws(1, p2["unary"], p2["conn"], p2["q"], p2["qprim"], p2["alphas"], 2.0, 4,
                                          -1e300, mode=1)
    assert it == it_o and np.array_equal(lab, lab_o) and en == en_o and lb == lb_o


@pytest.mark.parametrize("shape", [(40, 46, 16), (30, 24, 256)], ids=["k16", "k256"])
def test_two_processes_hand_over_through_ipc(shape, hip):
    """One strip per PROCESS: IPC export / open of the neighbour's arrays, flags across processes,
    all_gather of the partial sums (tools/strips_ipc_check.py; both ranks share the GPU here)."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
           "127.0.0.1", "--master-port", "29533", os.path.join(root, "tools", "strips_ipc_check.py")] + [str(v) for v in shape]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "IPC_STRIPS_OK" in r.stdout, r.stdout[-2000:] + r.stderr[-2000:]


def test_two_processes_on_two_devices(hip):
    """The real multi-GPU path, wherever two GPUs are visible: rank g on device g, RCCL for the handle
    exchange and the 2-double all_gather, fine-grained neighbour arrays mapped through HIP IPC, peer
    stores over xGMI polled from the neighbour's HBM; two problems in a row (the second upload
    resets arrays the neighbour writes into).  Skipped on the 1-GPU boxes of this pool."""
    import os
    import subprocess
    import sys
    if hip.device_count() < 2:
        pytest.skip("one GPU visible: the cross-device hand-over needs two")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    for shape in ((40, 46, 16), (30, 24, 256)):
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
               "127.0.0.1", "--master-port", "29541", os.path.join(root, "tools", "strips_ipc_check.py")] + [str(v) for v in shape]
        r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900)
        assert r.returncode == 0 and "IPC_STRIPS_OK" in r.stdout, r.stdout[-2000:] + r.stderr[-2000:]
        assert "backend nccl devices [0, 1]" in r.stdout, r.stdout[-2000:]


def test_one_process_driving_two_devices(hip, oracle):
    """TrwsStrips(devices=[0, 1]): one process, one strip per GPU, peer access instead of IPC; every plan
    entry point selects its plan's device (DeviceScope).  Skipped with one GPU."""
    from stereo_amd.strips import TrwsStrips, row_strip_owner
    if hip.device_count() < 2:
        pytest.skip("one GPU visible")
    H, W, K = 24, 20, 12
    p = trws_problem(161, H, W, K, kind="general")
    lab_o, en_o, lb_o, _ = oracle.trws(1, p["unary"], p["conn"], p["q"], p["qprim"], p["alphas"], 2.5, 4, -1e300, mode=1)
    s = TrwsStrips(1, K, H * W, p["conn"].T, row_strip_owner(H, W, 2), 2, devices=[0, 1])
    for _ in range(2):   # (bound twice: the second upload resets the strips)
        s.upload(p["unary"].T, p["alphas"], 2.5, q=p["q"].T, qprim=p["qprim"].T)
        s.iterate(4, max_relgap=-1e300)
        lab, en, lb, _ = s.result()
        assert np.array_equal(lab, lab_o) and _close(en, en_o) and _close(lb, lb_o)
    s.close()


def test_a_starved_wait_reports_what_it_waited_for(hip, monkeypatch):
    """A strip whose neighbour never runs: the sweep gives up after STEREO_HIP_TRWS_SPIN_SECONDS of wall
    clock (not after a poll count) and the error names the strip, the waiting visit and the flag."""
    from stereo_amd.strips import make_strips
    monkeypatch.setenv("STEREO_HIP_TRWS_SPIN_SECONDS", "0.2")
    H, W, K = 10, 9, 6
    p = trws_problem(171, H, W, K, kind="general")
    s = make_strips(1, K, H, W, p["conn"].T, 2)
    s.upload(p["unary"].T, p["alphas"], 2.0, q=p["q"].T, qprim=p["qprim"].T)
    with pytest.raises(hip.StereoHipError, match=r"gave up waiting on a dependency flag: strip 1 of 2 .* NEIGHBOURING strip"):
        s.plans[1].issue()          # strip 0 is never launched: strip 1 starves on its first boundary node
        s.plans[1].collect()
    s.close()


@pytest.mark.parametrize("G", [2, 3])
def test_strips_with_the_index_order_option(G, hip, oracle):
    """Row strips under STEREO_TRWS_ORDER_INDEX (runs are columns there, every one of them crosses
    every strip boundary): same labels as the oracle run in that order."""
    from stereo_amd.strips import make_strips
    from stereo_amd.trws import ORDER_INDEX
    H, W, K = 21, 17, 10
    p = trws_problem(120, H, W, K, kind="general")
    lab_o, en_o, lb_o, _ = oracle.trws(1, p["unary"], p["conn"], p["q"], p["qprim"], p["alphas"], 2.5, 4, -1e300,
                                       mode=1, ordering=1)
    s = make_strips(1, K, H, W, p["conn"].T, G, message_mode=ORDER_INDEX)
    s.upload(p["unary"].T, p["alphas"], 2.5, q=p["q"].T, qprim=p["qprim"].T)
    s.iterate(4, max_relgap=-1e300)
    lab, en, lb, _ = s.result()
    s.close()
    assert np.array_equal(lab, lab_o) and _close(en, en_o) and _close(lb, lb_o)


def test_a_strip_stores_its_own_rows_only(hip, oracle):
    """Strip-local storage: each of G strips holds N/G nodes + a halo row per neighbour and the
    edges at its own nodes; strip-local arrays bound on the device (bind_device_strip, the
    multi-GPU entry: no rank ever materialises the whole volume) give the oracle's labels."""
    import torch
    from stereo_amd.strips import make_strips
    H, W, K, G = 40, 36, 16, 4
    p = trws_problem(131, H, W, K, kind="general")
    lab_o, en_o, lb_o, _ = oracle.trws(1, p["unary"], p["conn"], p["q"], p["qprim"], p["alphas"], 2.0, 3, -1e300, mode=1)
    s = make_strips(1, K, H, W, p["conn"].T, G)
    N, E = H * W, p["conn"].shape[0]
    keep = []
    for g, pl in enumerate(s.plans):
        nodes, n_own, edges = pl.layout()
        assert n_own == np.sum(s._owner == g) and np.all(s._owner[nodes[:n_own]] == g)
        assert np.all(np.diff(nodes[:n_own]) > 0) and np.all(np.diff(nodes[n_own:]) > 0) and np.all(np.diff(edges) > 0)
        halo = nodes[n_own:]
        assert np.all(s._owner[halo] != g) and len(halo) <= 2 * (W + 2)
        touching = (s._owner[p["conn"][:, 0]] == g) | (s._owner[p["conn"][:, 1]] == g)
        assert np.array_equal(edges, np.nonzero(touching)[0])
        assert len(nodes) < N // G + 2 * (W + 2) + W and len(edges) < E // G + 6 * W
        dev = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
        u, a = dev(p["unary"][nodes]), dev(p["alphas"][edges])          # rows = nodes / edges, K fastest
        q, qp = dev(p["q"][edges]), dev(p["qprim"][edges])
        keep += [u, a, q, qp]
        pl.bind_device_strip(u.data_ptr(), a.data_ptr(), 2.0, d_q=q.data_ptr(), d_qprim=qp.data_ptr())
    torch.cuda.synchronize()
    done, _ = s.iterate(3, max_relgap=-1e300)
    lab, en, lb, _ = s.result()
    s.close()
    assert done == 3 and np.array_equal(lab, lab_o) and _close(en, en_o) and _close(lb, lb_o)


def test_strips_in_the_minplus_mode(hip, oracle):
    """MINPLUS messages with K = 200 run the wide kernel's plain min-plus branch; as two strips (group launch)
    they still give the brute-force oracle's labels."""
    from stereo_amd.strips import make_strips
    from stereo_amd.trws import MESSAGES_MINPLUS
    H, W, K = 14, 12, 200
    p = trws_problem(141, H, W, K, kind="fronto")
    pos = np.arange(K, dtype=np.float64)
    lab_o, en_o, lb_o, _ = oracle.trws(1, p["unary"], p["conn"], p["q"], p["qprim"], p["alphas"], 6.0, 3, -1e300, mode=0)
    s = make_strips(1, K, H, W, p["conn"].T, 2, message_mode=MESSAGES_MINPLUS)
    s.upload(p["unary"].T, p["alphas"], 6.0, positions=pos)
    s.iterate(3, max_relgap=-1e300)
    lab, en, lb, _ = s.result()
    assert s.path() == 3
    s.close()
    assert np.array_equal(lab, lab_o) and _close(en, en_o) and _close(lb, lb_o)


def test_strips_stop_test_with_zero_energy(hip, oracle):
    """(E - LB) / E with E == 0 is inf or NaN in minimize.cpp:105, not an error: an all-zero problem
    iterates to maxiter (found by tools/stress_trws.py)."""
    from stereo_amd.strips import make_strips
    H, W, K = 6, 7, 5
    p = trws_problem(151, H, W, K, kind="fronto")
    p["unary"][:] = 0.0
    pos = np.arange(K, dtype=np.float64)
    lab_o, en_o, lb_o, it_o = oracle.trws(1, p["unary"], p["conn"], p["q"], p["qprim"], p["alphas"], 2.0, 3, 0.0, mode=1)
    s = make_strips(1, K, H, W, p["conn"].T, 2)
    s.upload(p["unary"].T, p["alphas"], 2.0, positions=pos)
    done, stopped = s.iterate(3, max_relgap=0.0)
    lab, en, lb, _ = s.result()
    s.close()
    assert en_o == 0.0 and en == 0.0 and done == it_o and np.array_equal(lab, lab_o)


@pytest.mark.parametrize("G", [2, 4])
def test_the_gateway_shards_over_strips(G, hip, monkeypatch):
    """stereo_trws -- what trws_mex reaches -- with STEREO_HIP_GPUS=G: the image grid is recognised from the connectivity
    and cut into G row strips (logical strips on this box's one device, device g on a node that has G), behind the
    unchanged trws.m boundary: same labels, energy, bound, iteration count; a graph that is no image grid stays whole."""
    from stereo_amd import _lib
    from helpers import trws_problem
    strips_used = lambda: int(_lib.lib().stereo_trws_gateway_strips())
    for seed, H, W, K, kernel, kind, tol in ((41, 24, 19, 9, 1, "general", 2.0), (42, 20, 26, 16, 1, "fronto", 3.0),
                                             (43, 18, 16, 7, 2, "general", 4.0), (44, 16, 12, 130, 1, "fronto", 20.0)):
        p = trws_problem(seed, H, W, K, kind=kind)
        args = (kernel, p["unary"].T, p["conn"].T + 1, p["q"].T, p["qprim"].T, p["alphas"], tol, dict(maxiter=5, max_relgap=0.0))
        monkeypatch.delenv("STEREO_HIP_GPUS", raising=False)
        lab0, en0, lb0, it0 = hip.trws(*args)
        assert strips_used() == 1
        monkeypatch.setenv("STEREO_HIP_GPUS", str(G))
        lab, en, lb, it = hip.trws(*args)
        assert strips_used() == G
        assert it == it0 and np.array_equal(lab, lab0) and en == en0 and lb == lb0
        lab2, en2, lb2, it2 = hip.trws(*args)     # the cached strips once more
        assert np.array_equal(lab2, lab0) and en2 == en0 and lb2 == lb0
    # not an image grid (a ring): one device, whatever the switch says
    n = 40
    rng = np.random.default_rng(5)
    conn = np.stack([np.arange(n), (np.arange(n) + 1) % n]) + 1
    q = rng.uniform(0, 8, (5, n)); qp = rng.uniform(0, 8, (5, n))
    hip.trws(1, rng.uniform(0, 9, (5, n)), conn, q, qp, np.ones(n), 2.0, dict(maxiter=3))
    assert strips_used() == 1
