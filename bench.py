#!/usr/bin/env python
"""bench.py -- headline benchmark of the TRW-S fusion path on MI355X.

Workload at N=1 (BASELINE.json configs[1]): the NCC cost volume of a synthetic
Teddy-sized image pair, 450 x 375 pixels x 60 fronto-parallel disparity labels
(unary = 40 (1 - ncc), q = qprim = 0..59, alphas = 1, tol = 8, kernel 1 --
SURVEY.md 8(d) "Config 2").  A "step" is one
TRW-S iteration (forward sweep, backward sweep with lower bound, primal
labelling + energy, stop test), exactly what Minimize_TRW_S does per iteration
(cpp/trw-s/minimize.cpp:31-113).  Inputs are resident in HBM before the timed
region.

With --gpus N > 1 the top-level line is the north star's scaling figure instead: ONE synthetic
3000 x 2000 x 256-label image (BASELINE.json configs[3]) tiled into N row strips, one per GPU
(stereo_amd.strips: peer stores over xGMI, RCCL only for two doubles per iteration), strong
scaling, `value` = iterations/s of that one image over exactly --steps timed iterations.  Before
anything is timed a PREFLIGHT runs two small strip problems across the N real devices and compares
them with the single plan on rank 0 (`preflight`).  The one-Teddy-pair-per-GPU replica figure is a
labelled extra (`replicas`), never `value`.  At N = 1 the same configs[3] figure is the `scale` object.

One JSON line is printed by rank 0.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: 8.0 TB/s spec
PROFILE_ROUND = "r06"    # roofline.traffic is read from THIS round's committed PMC passes only (profiles/r06_*), never older ones


def synthetic_volume(H, W, K, seed):
    """Planted piecewise-planar disparity + noise, as a (N, K) unary in [0, 40)."""
    rng = np.random.default_rng(seed)
    rows, cols = np.meshgrid(np.arange(H), np.arange(W), indexing="ij")
    truth = (0.25 * K + 0.5 * K * (cols / W) + 0.1 * K * np.sin(rows / 37.0)).clip(0, K - 1)
    truth[(rows // 60 + cols // 75) % 2 == 0] *= 0.6
    lab = np.arange(K)[None, None, :]
    cost = np.minimum(np.abs(lab - truth[:, :, None]) / 4.0, 1.0) * 30.0
    cost = cost + rng.uniform(0, 10, size=cost.shape)
    return np.ascontiguousarray(cost.transpose(1, 0, 2).reshape(H * W, K))  # node id = col*H + row


def synthetic_volume_device(H, W, K, seed, dev, nodes=None):
    """The same construction on the device (torch RNG), for volumes too large to build on the host
    in reasonable time (3000 x 2000 x 256: 12 GB).  nodes (device int64 tensor): keep only the rows
    of these nodes, in this order -- a strip's share of the SAME volume (the random stream is
    drawn chunk by chunk for all nodes, so every rank sees the values a single GPU would)."""
    import torch
    g = torch.Generator(device=dev)
    g.manual_seed(seed)
    cols = torch.arange(W, device=dev, dtype=torch.float64)[:, None]
    rows = torch.arange(H, device=dev, dtype=torch.float64)[None, :]
    truth = (0.25 * K + 0.5 * K * (cols / W) + 0.1 * K * torch.sin(rows / 37.0)).clamp(0, K - 1)
    mask = ((torch.div(rows, 60, rounding_mode="floor") + torch.div(cols, 75, rounding_mode="floor")) % 2) == 0
    truth = torch.where(mask, truth * 0.6, truth).reshape(H * W, 1)  # node id = col*H + row
    out = torch.empty(H * W if nodes is None else nodes.numel(), K, dtype=torch.float64, device=dev)
    lab = torch.arange(K, device=dev, dtype=torch.float64)[None, :]
    step = max(1, (1 << 26) // K)
    for a in range(0, H * W, step):
        b = min(H * W, a + step)
        chunk = torch.clamp((lab - truth[a:b]).abs() / 4.0, max=1.0) * 30.0
        chunk += torch.rand(b - a, K, generator=g, device=dev, dtype=torch.float64) * 10.0
        if nodes is None:
            out[a:b] = chunk
        else:
            here = (nodes >= a) & (nodes < b)
            out[here] = chunk[nodes[here] - a]
    return out


def synthetic_pair(H, W, D, seed=0):
    """A smooth random texture and its warp by a planted disparity field: an image pair for the
    NCC cost volume (dispmap_ncc) of the end-to-end fusion figure."""
    rng = np.random.default_rng(seed)
    tex = rng.uniform(0, 255, size=(H, W + D, 3))
    for _ in range(3):
        tex = (tex + np.roll(tex, 1, 0) + np.roll(tex, 1, 1)) / 3
    disp = (0.15 * D + 0.5 * D * (np.arange(W)[None, :] / W) + 0.08 * D * np.sin(np.arange(H)[:, None] / 40.0)).astype(int)
    im0 = tex[:, :W]
    cols = np.clip(np.arange(W)[None, :] + disp, 0, W + D - 1)
    im1 = np.stack([np.take_along_axis(tex[:, :, c], cols, 1) for c in range(3)], axis=2)
    return im0, im1


def algorithmic_bytes_per_sweep_pair(info_deg, K):
    """per node and sweep read D (K) + every incident message,
    write the outgoing ones; 8-byte reals.  info_deg = (nf, nb) arrays by node."""
    nf, nb = info_deg
    fwd = (1 + nf + nb + nf) * K * 8.0
    bwd = (1 + nf + nb + nb) * K * 8.0
    return float(fwd.sum() + bwd.sum())


def scale_leg(args, rank, local_rank, world, dist, dev, index_order=False, minplus=False, steps=None, warmup=None):
    """Strong scaling of ONE large image (BASELINE.json configs[3]: synthetic 3000 x 2000 x 256-label
    volume) over the `world` GPUs: rank g owns band g of the rows (stereo_amd.strips), boundary
    messages / flags / labels go to the neighbour GPU as peer stores over xGMI, energy and bound are
    reduced as two doubles per iteration.  At world == 1 it is the plain single-GPU plan, so the N=1
    figure of a scaling run is the single-GPU figure.  Returns the `scale` object (rank 0) or None."""
    import torch
    from stereo_amd import dist as D
    from stereo_amd.trws import TrwsPlan
    from stereo_amd.strips import TrwsStripRank, row_strip_owner
    from helpers import grid_conn
    from stereo_amd import _lib
    _cus = _lib.lib().stereo_hip_device_cus
    H, W, K = args.scale_height, args.scale_width, args.scale_labels
    N = H * W
    steps = args.scale_steps if steps is None else steps
    warmup = args.scale_warmup if warmup is None else warmup
    t_setup = time.perf_counter()
    conn = grid_conn(H, W)
    E = conn.shape[0]
    d_pos = torch.arange(K, dtype=torch.float64, device=dev)
    # labelled extras: STEREO_TRWS_ORDER_INDEX (not the gateway's node order), STEREO_TRWS_MESSAGES_MINPLUS
    # (plain min-plus messages: the reference's bits unless a message's certificate would have failed)
    mode = (0x100 if index_order else 0) | (1 if minplus else 0)
    if world == 1:
        solver = TrwsPlan(1, K, N, conn.T, message_mode=mode)
        plan = solver
        d_unary = synthetic_volume_device(H, W, K, 1, dev)
        d_alpha = torch.ones(E, dtype=torch.float64, device=dev)
        torch.cuda.synchronize()
        solver.bind_device(d_unary.data_ptr(), d_alpha.data_ptr(), 8.0, d_positions=d_pos.data_ptr(),
                           keepalive=(d_unary, d_alpha, d_pos))
        stored = (N, E)
    else:
        # a rank stores its band only: the rows of its own nodes + one halo row per neighbour and the
        # messages of the edges at its own nodes (strip-local ids, stereo_trws_plan_strip_layout)
        solver = TrwsStripRank(1, K, H, W, conn.T, rank, world, dist, dev, message_mode=mode,
                               max_workgroups=((int(_cus()) or 256) // world if args.one_gpu else 0))
        plan = solver.plan
        nodes, n_own, edges = plan.layout()
        d_unary = synthetic_volume_device(H, W, K, 1, dev, nodes=torch.from_numpy(nodes.astype(np.int64)).to(dev))
        d_alpha = torch.ones(len(edges), dtype=torch.float64, device=dev)
        torch.cuda.synchronize()
        # (through the rank object: it brackets the state reset with barriers, so that no rank issues its
        #  first sweep into arrays a slower neighbour has yet to wipe)
        solver.bind_device_strip(d_unary.data_ptr(), d_alpha.data_ptr(), 8.0, d_positions=d_pos.data_ptr(),
                                 keepalive=(d_unary, d_alpha, d_pos))
        stored = (len(nodes), len(edges))
    plan.stats(reset=True)
    setup_s = time.perf_counter() - t_setup
    NEVER = -1e300
    solver.iterate(warmup, max_relgap=NEVER)
    plan.stats(reset=True)
    plan.serial_messages(reset=True)
    D.barrier(dist, dev)
    t0 = time.perf_counter()
    done, _ = solver.iterate(steps, max_relgap=NEVER)
    assert done == steps
    D.barrier(dist, dev)
    dt = D.max_over_ranks(dist, time.perf_counter() - t0, dev)
    sweep_ms, launches = plan.stats()
    serial = D.sum_over_ranks(dist, plan.serial_messages(), dev)
    if world == 1:
        lab, energy, lb, _ = solver.result(want_labels=True)
        crc = int(np.asarray(lab, dtype=np.int64).sum())
        own_nodes = N
    else:
        idx, lab = solver.own_labels()
        crc = int(D.sum_over_ranks(dist, float(np.asarray(lab, dtype=np.int64).sum()), dev))
        energy, lb = solver.energy, solver.lb
        own_nodes = len(idx)
    # algorithmic bytes of one sweep launch on THIS rank: 13 K R per interior node; the
    # per-node figure with the real degrees is within 0.1 % of it at this size
    bytes_launch = 13.0 * K * 8.0 * own_nodes
    avg_launch_s = (sweep_ms * 1e-3) / max(launches, 1)
    frac = bytes_launch / avg_launch_s / 1e9 / HBM_PEAK_GBS
    frac_min = -D.max_over_ranks(dist, -frac, dev)
    frac_max = D.max_over_ranks(dist, frac, dev)
    # who took part: one line per rank (device, strip size) gathered on rank 0 -- a reader of a multi-GPU
    # record sees at once whether N different devices ran N strips of one image
    import socket
    me = {"rank": rank, "device": int(dev.index if dev.index is not None else 0), "host": socket.gethostname(),
          "own_nodes": int(own_nodes), "sweep_launch_ms": avg_launch_s * 1e3}
    ranks = [me]
    if dist is not None and world > 1:
        gathered = [None] * world
        dist.all_gather_object(gathered, me)
        ranks = gathered
    if rank != 0:
        return None
    # N = 1 results of the default scale workload (driver-run BENCH_r02 / BENCH_r03, 1 warm-up + 3 timed
    # iterations): labels are the same at any N, energy and bound agree to 1e-12 relative
    ref1 = None
    if (H, W, K, warmup + steps, index_order, minplus) == (2000, 3000, 256, 4, False, False):
        ref1 = {"label_sum": 624442294, "energy": 47220100.82134177, "lower_bound": 47214056.1673238}
    n1 = None
    if ref1 is not None:
        n1 = dict(ref1, label_sum_equal=bool(crc == ref1["label_sum"]),
                  energy_rel_diff=abs(energy - ref1["energy"]) / abs(ref1["energy"]),
                  lower_bound_rel_diff=abs(lb - ref1["lower_bound"]) / abs(ref1["lower_bound"]),
                  source="committed constants of the N = 1 run (BENCH_r02.json scale object)")
    return {"workload": "configs[3]: synthetic %dx%dx%d-label cost volume, TRW-S, kernel 1, tol 8, ONE image tiled "
                        "into %d row strips" % (W, H, K, world),
            "n_gpus": world, "scaling": "strong", "value": steps / dt, "unit": "iterations/s",
            "ms_per_iteration": dt / steps * 1e3, "steps": steps, "warmup": warmup,
            "bytes_per_launch_per_gpu": bytes_launch,
            "hbm_frac_per_gpu": [frac_min, frac_max], "sweep_launch_ms_rank0": avg_launch_s * 1e3,
            "serial_envelope_messages": serial, "energy": energy, "lower_bound": lb, "label_sum": crc,
            "transport": "none (one GPU)" if world == 1 else "peer stores over xGMI into HIP-IPC-mapped neighbour arrays + flag; "
                         "RCCL all_gather of 2 doubles per iteration",
            "stored_per_gpu": {"nodes": int(stored[0]), "edges": int(stored[1]),
                               "unary_and_message_bytes": int((stored[0] + stored[1]) * K * 8)},
            "setup_s": setup_s,
            "ranks": ranks, "distinct_devices": len({(r["host"], r["device"]) for r in ranks}),
            "collective_backend": (dist.get_backend() if dist is not None and world > 1 else "none"),
            "n1_reference": n1,
            "speculative_chain": (plan.spec_stats() if world == 1 and hasattr(plan, "spec_stats") else
                                  {"active": False, "why": "strips keep the plain chain schedule"}),
            "note": "the north star's 1 -> N scaling curve (one image, strong scaling): at N > 1 this object IS the "
                    "top-level value; at N = 1 the top-level value is the Teddy headline and this is the curve's first point"}


def preflight(rank, world, dist, dev, one_gpu):
    """Before anything is timed at N > 1: two small problems solved by `world` row strips on the `world`
    REAL devices (the same TrwsStripRank path as the scale leg: HIP-IPC mappings, peer stores, flags,
    the 2-double all_gather) against the single plan on rank 0 -- K = 16 with per-edge positions on the
    K <= 64 kernel and K = 256 fronto-parallel labels on the wide kernel, 2 iterations each.  Returns
    the `preflight` object on rank 0: ok, and per case labels_equal / energy / bound differences."""
    import torch
    from stereo_amd import _lib
    from stereo_amd.strips import TrwsStripRank
    from stereo_amd.trws import TrwsPlan
    from helpers import trws_problem
    cases = []
    ok = True
    for (H, W, K) in ((64, 96, 16), (max(32, 4 * world), 24, 256)):
        case = {"grid": "%dx%dx%d" % (W, H, K), "strips": world}
        try:
            kind = "fronto" if K > 64 else "general"
            p = trws_problem(7, H, W, K, kind=kind)
            cus = int(_lib.lib().stereo_hip_device_cus()) or 256
            s = TrwsStripRank(1, K, H, W, p["conn"].T, rank, world, dist, dev,
                              max_workgroups=max(2, cus // world) if one_gpu else 0)
            pos = np.arange(K, dtype=np.float64)
            if kind == "fronto":
                s.upload(p["unary"].T, p["alphas"], 8.0, positions=pos)
            else:
                s.upload(p["unary"].T, p["alphas"], 3.0, q=p["q"].T, qprim=p["qprim"].T)
            done, _ = s.iterate(2, max_relgap=-1e300)
            idx, lab = s.own_labels()
            parts = [None] * world
            dist.all_gather_object(parts, (idx, lab))
            if rank == 0:
                full = np.zeros(H * W)
                for i, l in parts:
                    full[i] = l
                one = TrwsPlan(1, K, H * W, p["conn"].T)
                if kind == "fronto":
                    one.upload(p["unary"].T, p["alphas"], 8.0, positions=pos)
                else:
                    one.upload(p["unary"].T, p["alphas"], 3.0, q=p["q"].T, qprim=p["qprim"].T)
                one.iterate(2, max_relgap=-1e300)
                lab1, en1, lb1, _ = one.result()
                one.close()
                case.update(kernel_path=int(s.plan.path()), labels_equal=bool(np.array_equal(full, lab1)),
                            label_sum=int(full.sum()), label_sum_single_plan=int(np.asarray(lab1).sum()),
                            energy=s.energy, energy_single_plan=en1, lower_bound=s.lb, lower_bound_single_plan=lb1,
                            energy_rel_diff=abs(s.energy - en1) / max(abs(en1), 1e-300),
                            lower_bound_rel_diff=abs(s.lb - lb1) / max(abs(lb1), 1e-300))
                case["ok"] = bool(case["labels_equal"] and case["energy_rel_diff"] <= 1e-12 and case["lower_bound_rel_diff"] <= 1e-12)
            s.close()
        except Exception as exc:
            case.update(ok=False, error="%s: %s" % (type(exc).__name__, exc))
        ok = ok and bool(case.get("ok", rank != 0))
        cases.append(case)
    return {"ok": ok, "cases": cases,
            "what": "row strips on the N devices of this run vs the single plan on rank 0, 2 iterations, before any timing; "
                    "labels bit-exact, energy / bound to 1e-12 relative (strip-ordered partial sums)"} if rank == 0 else None


def cpu_all_cores(unary, conn, K, iters):
    """The oracle (oracle/trws_oracle.c, 1 thread per problem -- the reference is single-threaded) on
    C independent copies of the same Teddy-sized problem at once, one per host core up to 64 (each copy
    holds ~0.5 GB of messages): iterations/s of all copies together.  Reported baseline."""
    from concurrent.futures import ThreadPoolExecutor
    from oracle import pyoracle
    E = conn.shape[0]
    host = os.cpu_count() or 1
    try:
        host_avail = len(os.sched_getaffinity(0))
    except Exception:
        host_avail = host
    C_ = max(1, min(host_avail, 64))
    q = np.tile(np.arange(K, dtype=np.float64), (E, 1))
    ones = np.ones(E)

    def one(_):
        r = pyoracle.trws(1, unary, conn, q, q, ones, 8.0, maxiter=iters, max_relgap=-1e300, mode=1, want_trace=True)
        return float(r[4][-1, 2])
    t0 = time.perf_counter()
    with ThreadPoolExecutor(C_) as ex:
        secs = list(ex.map(one, range(C_)))
    wall = time.perf_counter() - t0
    return {"value": C_ * iters / max(secs), "unit": "iterations/s", "cores": C_, "host_cores": host,
            "host_cores_available": host_avail, "kind": "port",
            "sample": "%d independent copies of the problem x %d iterations, one oracle thread each (ctypes releases the "
                      "GIL), slowest copy's iteration time; %.1f s wall incl. every copy's setup" % (C_, iters, wall)}


def hard_moves_leg(which, H, W, im0_synth):
    """QPBO hard moves/s through the mirrored class (state resident in HBM, one proposal uploaded per move), then --
    outside the timed region -- the SAME moves through the reference's QPBO library on the host (oracle/_ref, srand()
    before every move as tests/test_globalstereo_gpu.py does): `cpu_reference` with `labels_equal`."""
    import ctypes
    import stereo_amd
    from stereo_amd import terms as T
    P34 = np.tile(np.hstack([np.eye(3), np.zeros((3, 1))])[:, :, None], (1, 1, 2))
    P34[0, 3, 1] = -0.25
    gold = os.path.join(ROOT, "tests", "golden")
    grng = np.random.default_rng(0)
    if which == "teddy":
        g = np.load(os.path.join(gold, "teddy_pair.npz"))
        im0, im1 = g["im0"].astype(np.float64), g["im1"].astype(np.float64)
        H, W = im0.shape[:2]
        sg = np.load(os.path.join(gold, "teddy_segments.npz"))
        gs = stereo_amd.dispmap_globalstereo([im0, im1], P34, (0, 59), 4, segment=sg["segment"], rng=grng)
        t1 = time.perf_counter()
        props = gs.segpln([sg["segments"][:, :, b] for b in range(14)], seed=0)     # window matching + LO-RANSAC on the device
        t_props = time.perf_counter() - t1
        what = ("example_global.m on the Teddy pair: edge weights from the reference's mean-shift segmentation, the 14 SegPln "
                "proposals on the reference's 14 segmentation maps (planes fitted on the device)")
    else:
        sys.path.insert(0, os.path.join(ROOT, "examples"))
        from example_global import piecewise_planar
        im0, im1 = synthetic_pair(H, W, 60)
        gconn = T.construct_neighborhood(H, W)
        gimg = im0.transpose(1, 0, 2).reshape(H * W, -1)
        same = np.abs(gimg[gconn[0]] - gimg[gconn[1]]).sum(axis=1) < 30.0
        gs = stereo_amd.dispmap_globalstereo([im0, im1], P34, (0, 59), 4, smooth_weights=np.where(same, 108.0, 9.0) * 2.0, rng=grng)
        props = [piecewise_planar(H, W, cell, grng, gs.d_min, gs.d_min + gs.d_step) for cell in (8, 12, 16, 24, 32, 48, 64) for _ in range(2)]
        t_props = None
        what = "examples/example_global.py on the synthetic pair: 14 block-wise random-plane proposals, colour-difference edge weights"
    N = H * W
    start = np.array(gs.assignment)
    libc = ctypes.CDLL(None)
    unl = []
    t1 = time.perf_counter()
    for k, pl in enumerate(props):
        libc.srand(1000 + k)
        unl.append(gs.binary_fusion(pl)[2])
    tg = time.perf_counter() - t1
    free_assignment = np.array(gs.assignment)     # the device's own trajectory: never resynchronised with the reference's
    res = {"moves_per_s": len(props) / tg, "ms_per_move": tg / len(props) * 1e3, "energy": gs.energy(), "moves": len(props),
           "grid": "%dx%d" % (W, H), "moves_with_unlabelled_nodes": int(sum(1 for u in unl if u > 0)),
           "what": what + "; binary fusion with QPBO + weak persistency + Improve"}
    if t_props is not None:
        res["proposals_s"] = t_props
    from oracle import pyoracle, terms as ot
    if pyoracle.have_ref_qpbo():
        # the same moves on the host: NumPy pairwise terms, the device's unaries (exp / log differ in the last bit between
        # libm and the device), the reference library with Improve.  Only the library call is timed.
        i1, i2 = ot.construct_neighborhood(H, W)
        conn = np.stack([i1, i2], 1)
        pts = ot.get_points(H, W)
        disp = lambda a, p: ot.globalstereo_rescale(ot.disparity_from_assignment(a, p), gs.d_min, gs.d_step)
        un = lambda a: T.globalstereo_unary(im0, im1, gs.P2, gs.d_min, gs.d_step, gs.options["col_thresh"], np.asfortranarray(a))
        a = start.copy()
        gs.assignment = start.copy()
        t_ref = 0.0
        differ_total, ties_only, unl_equal = 0, True, True
        for k, pl in enumerate(props):
            pl = np.asarray(pl.expand(N) if hasattr(pl, "expand") else pl)
            U0, U1 = un(a), un(pl)
            E = ot.all_pairwise_costs(1, np.asarray(gs.smooth_weights), gs.tol, a, pl, i1, i2, pts, disp_fn=disp)
            t1 = time.perf_counter()
            lab, e_r, lb_r, nu_r = pyoracle.ref_rd(U0, U1, *E, conn, improve=True, seed=1000 + k)
            t_ref += time.perf_counter() - t1
            a[:, lab == 1] = pl[:, lab == 1]
            libc.srand(1000 + k)
            nu = gs.binary_fusion(pl)[2]
            unl_equal = unl_equal and nu == nu_r
            d = (gs.assignment != a).any(0)
            if d.any():
                differ_total += int(d.sum())
                ties_only = ties_only and bool(np.all(np.abs(U0[d] - U1[d]) <= 1e-14))
                gs.assignment = a.copy()      # the same state on both sides for the next move
        res["cpu_reference"] = {"moves_per_s": len(props) / t_ref, "ms_per_move": t_ref / len(props) * 1e3, "kind": "reference",
                                "cores": 1, "what": "the reference's QPBO v1.3 library (oracle/_ref), Solve + weak persistency + "
                                "Improve, library call only (terms excluded)",
                                "labels_equal": bool(differ_total == 0),
                                "pixels_differing_over_all_moves": differ_total, "differences_only_at_unary_ties": ties_only,
                                "ties_note": "|U0 - U1| <= 1e-14: the reference's own label there depends on the order of its input edges "
                                             "(tests/test_oracle_qpbo.py); equal everywhere on an exact grid of inputs (tests/test_globalstereo_gpu.py)",
                                "num_unlabelled_equal": bool(unl_equal), "energy": float(e_r)}
        # what the near-tie contract means over a whole run: the device's free-running trajectory (the timed loop above)
        # against the reference's own (a: driven by the reference's labels alone)
        fd = (free_assignment != a).any(0)
        res["free_running_vs_reference"] = {
            "what": "all %d moves WITHOUT resynchronising: planes of the device's own trajectory against the trajectory the reference's "
                    "library drives on its own labels" % len(props),
            "pixels_with_another_plane": int(fd.sum()), "pixels": int(N),
            "energy_device": float(res["energy"]), "energy_reference": float(e_r),
            "energy_rel_diff": float(abs(res["energy"] - e_r) / abs(e_r))}
    return res


def segmentation_leg():
    """the segmenters behind dispmap_globalstereo on the Teddy pair's reference image -- vgg_segment_ms(R, 4, 5, 0)
    for the edge weights (dispmap_globalstereo.m:391) and the 14 maps of segpln (:121-134) -- against the committed maps of the
    reference's own segmenters and, where oracle/_ref travelled, against those segmenters timed on this host."""
    from stereo_amd import segment as S
    gold = os.path.join(ROOT, "tests", "golden")
    im = np.load(os.path.join(gold, "teddy_pair.npz"))["im0"]
    sg = np.load(os.path.join(gold, "teddy_segments.npz"))
    S.vgg_segment_ms(np.ascontiguousarray(im[:48, :64]), 2, 3, 0)   # (first-call costs out of the figures)
    t = time.perf_counter(); seg = S.vgg_segment_ms(im, 4, 5, 0); t_ms = time.perf_counter() - t
    t = time.perf_counter(); maps = S.segpln_segments(im); t_maps = time.perf_counter() - t
    t = time.perf_counter(); own, ev = S.ms_own(im, 4, 5); t_own = time.perf_counter() - t
    out = {"what": "vgg_segment_ms(R, 4, 5, 0) and the 14 SegPln maps of the Teddy pair's reference image (450x375): mean-shift filter one "
                   "thread per pixel on the device, the reference's scan-order shortcuts / region graph / sort + union-find on the host",
           "segment_ms": t_ms * 1e3, "segments": int(seg.max()), "fourteen_maps_ms": t_maps * 1e3,
           "filter_stage_ms": t_own * 1e3, "pixels_walked_again_on_the_host": int(ev.sum()),
           "equal_to_the_reference_maps": bool(np.array_equal(seg, sg["segment"]) and np.array_equal(maps, sg["segments"]))}
    # the plane fits on those maps (segpln :140-197, stereo_segpln_planes_batch) against the committed oracle planes
    from stereo_amd import terms as T
    pl = np.load(os.path.join(gold, "teddy_segpln_planes.npz"))
    T.segpln_planes_batch(pl["wta"], maps, list(range(14)))
    t = time.perf_counter(); fits = T.segpln_planes_batch(pl["wta"], maps, list(range(14))); t_fit = time.perf_counter() - t
    t = time.perf_counter(); T.segpln_planes_batch(pl["wta"], maps, list(range(14)), want_proposal=False); t_fit0 = time.perf_counter() - t
    out["plane_fits"] = {"what": "LO-RANSAC + least-squares plane of every segment of the 14 maps in one call (12 766 + 469 + 94 segments, the largest "
                                 "154 396 pixels), seeds 0 .. 13, on the committed winner-takes-all map",
                         "ms_with_the_14_proposal_arrays": t_fit * 1e3, "ms_planes_and_inlier_counts": t_fit0 * 1e3,
                         "proposals_equal_to_the_oracle_fixture": bool(all(np.array_equal(
                             fits[b][0], pl["planes_%d" % b][maps[:, :, b].T.reshape(-1).astype(np.int64) - 1].T, equal_nan=True) for b in range(14)))}
    from oracle import pyoracle
    if pyoracle.have_ref_segment() and pyoracle.ref_segment_gb_lib() is not None:
        t = time.perf_counter(); r = pyoracle.ref_segment_ms(im, 4, 5.0, 0); t_r = time.perf_counter() - t
        t = time.perf_counter(); rm = pyoracle.ref_segpln_segments(im); t_rm = time.perf_counter() - t
        out["cpu_reference"] = {"kind": "reference", "cores": 1, "segment_ms": t_r * 1e3, "fourteen_maps_ms": t_rm * 1e3,
                                "equal": bool(np.array_equal(r, seg) and np.array_equal(rm, maps))}
    return out


def gateway_leg(G, H=750, W=1000, K=60, iters=10):
    """the C-ABI entry the reference's gateway reaches (stereo_trws, host arrays as trws.m:33 hands them)
    sharding one image over G devices BY ITSELF (STEREO_HIP_GPUS=G: one process, row strips, peer access between the
    devices -- or logical strips where the process sees fewer devices), against the same call on one device.  Runs in a
    process of its own (python bench.py --gateway-leg G): at N > 1 rank 0 starts it after the ranks' own work."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import stereo_amd
    from stereo_amd import _lib
    from helpers import grid_conn
    conn1 = grid_conn(H, W).T + 1
    E = conn1.shape[1]
    h_un = np.asfortranarray(synthetic_volume(H, W, K, 3).T)                                  # K x N
    h_q = np.asfortranarray(np.tile(np.arange(K, dtype=np.float64)[:, None], (1, E)))          # K x E (dispmap_super.m:177-183)
    opts = dict(maxiter=iters, max_relgap=-1e300)
    res = {}
    for tag, g in (("strips", G), ("one_device", 1)):
        os.environ["STEREO_HIP_GPUS"] = str(g)
        _lib.lib().stereo_trws_cache_clear()
        t = []
        for _ in range(2):
            t0 = time.perf_counter()
            r = stereo_amd.trws(1, h_un, conn1, h_q, h_q, np.ones(E), 8.0, opts)
            t.append(time.perf_counter() - t0)
        res[tag] = (r, t, int(_lib.lib().stereo_trws_gateway_strips()))
    _lib.lib().stereo_trws_cache_clear()
    (a, ta, sa), (b, tb, _) = res["strips"], res["one_device"]
    return {"what": "stereo_trws with STEREO_HIP_GPUS=%d on a %dx%dx%d volume (host arrays, %d iterations, labels back) against the same call "
                    "on one device" % (G, W, H, K, iters),
            "strips": sa, "devices_visible": int(stereo_amd.device_count()), "distinct_devices": min(sa, int(stereo_amd.device_count())),
            "first_call_s": ta[0], "second_call_s": ta[1], "one_device_second_call_s": tb[1],
            "labels_equal": bool(np.array_equal(a[0], b[0])), "energy_equal": bool(a[1] == b[1]), "lower_bound_equal": bool(a[2] == b[2]),
            "iterations_equal": bool(a[3] == b[3])}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--height", type=int, default=375)
    ap.add_argument("--width", type=int, default=450)
    ap.add_argument("--labels", type=int, default=60)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true",
                    help="skip the labelled index-order extra (profiling: its launches are the same kernel and "
                         "would be averaged into the headline kernel's rocprofv3 statistics)")
    ap.add_argument("--cpu-iters", type=int, default=10)
    ap.add_argument("--no-cpu-all-cores", action="store_true", help="skip the all-cores leg of the CPU baseline")
    ap.add_argument("--message-mode", choices=["exact", "minplus"], default="exact")
    ap.add_argument("--volume", choices=["ncc", "noise"], default="ncc",
                    help="ncc: NCC cost volume of an image pair (the named workload); noise: planted signal + noise")
    ap.add_argument("--pair", choices=["teddy", "synthetic"], default="teddy",
                    help="image pair behind the NCC volume: the reference's Teddy pair (tests/golden/teddy_pair.npz = "
                         "data/teddy/im2.png, im6.png; 450x375 only) or the synthetic textured pair")
    ap.add_argument("--no-scale", action="store_true", help="skip the 3000x2000x256 strong-scaling leg")
    ap.add_argument("--gateway-leg", type=int, default=0, metavar="G",
                    help="run ONLY the gateway leg (stereo_trws sharding over G devices by itself) in this process and print its JSON")
    ap.add_argument("--no-gateway-leg", action="store_true", help="N > 1: do not start the gateway leg's process")
    ap.add_argument("--scale-height", type=int, default=2000)
    ap.add_argument("--scale-width", type=int, default=3000)
    ap.add_argument("--scale-labels", type=int, default=256)
    ap.add_argument("--scale-steps", type=int, default=3)
    ap.add_argument("--scale-warmup", type=int, default=1)
    ap.add_argument("--scale-options", action="store_true",
                    help="N > 1: also time the labelled min-plus / index-order options of the scale leg (at N = 1 they always run)")
    ap.add_argument("--no-scale-n1", action="store_true",
                    help="N > 1: skip rank 0's own single-GPU run of the same image, same --steps / --warmup, after the strips "
                         "(the curve's first point measured in the same job, and a label / energy cross-check of the strips)")
    ap.add_argument("--no-preflight", action="store_true")
    args = ap.parse_args()
    if args.gateway_leg > 0:
        print(json.dumps(gateway_leg(args.gateway_leg)))
        return

    import torch
    from stereo_amd import dist as D
    # STEREO_BENCH_ONE_GPU=1: development aid for boxes with a single GPU -- all ranks share GPU 0
    # (rendezvous over gloo; the strips' workgroups split the CUs) so that the N > 1 code path, IPC
    # hand-over included, can be run end to end.  Not a measurement.
    one_gpu = bool(os.environ.get("STEREO_BENCH_ONE_GPU"))
    rank, local_rank, world, dist = D.init(("gloo" if one_gpu else "nccl") if int(os.environ.get("WORLD_SIZE", "1")) > 1 else None)
    if one_gpu:
        local_rank = 0
    args.one_gpu = one_gpu
    torch.cuda.set_device(local_rank)

    import stereo_amd
    from stereo_amd import _lib
    from stereo_amd.trws import TrwsPlan, analyze
    from helpers import grid_conn
    _lib.lib().stereo_hip_set_device(local_rank)

    H, W, K = args.height, args.width, args.labels
    N = H * W
    conn = grid_conn(H, W)
    E = conn.shape[0]
    dev = torch.device("cuda", local_rank)
    # who is here: one entry per rank as the collective backend sees it (host, device index, device name, uuid-ish bus id)
    import socket
    props = torch.cuda.get_device_properties(local_rank)
    me = {"rank": rank, "local_rank": local_rank, "host": socket.gethostname(), "device": local_rank, "name": props.name,
          "pci_bus_id": getattr(props, "pci_bus_id", None), "cus": getattr(props, "multi_processor_count", None)}
    ranks_seen = [me]
    if dist is not None and world > 1:
        ranks_seen = [None] * world
        dist.all_gather_object(ranks_seen, me)
    pre = None
    if world > 1 and not args.no_preflight:
        pre = preflight(rank, world, dist, dev, one_gpu)
    big = N * K > (1 << 28)
    volume = args.volume if not big else "noise"
    if volume == "ncc":
        # the named workload: NCC cost volume of an image pair (dispmap_ncc.m:116-198, computed by the
        # product's own kernel, label fastest) -> unary = unary_weight * (1 - ncc), weight 40
        # (example_ncc.m:13-16).  The masked columns left of each disparity are flat zeros (:190-191),
        # so the volume has exact ties like the real Teddy volume and part of the messages take the
        # reference's serial envelope construction.
        from stereo_amd import terms as T
        teddy = os.path.join(ROOT, "tests", "golden", "teddy_pair.npz")
        if args.pair == "teddy" and (H, W) == (375, 450) and os.path.exists(teddy):
            g = np.load(teddy)      # every rank solves the same Teddy problem (weak scaling: one pair per GPU)
            im0, im1 = g["im0"].astype(np.float64), g["im1"].astype(np.float64)
            pair = "teddy"
        else:
            im0, im1 = synthetic_pair(H, W, K, seed=rank)
            pair = "synthetic"
        ncc = T.ncc_volume(im0, im1, np.arange(K, dtype=np.float64), 2, layout=1)     # K x N, label fastest
        h_unary = np.ascontiguousarray(40.0 * (1.0 - ncc.T))                          # (N, K)
        d_unary = torch.from_numpy(h_unary).to(dev)
    elif big:
        d_unary = synthetic_volume_device(H, W, K, 1 + rank, dev)
        pair = "none"
    else:
        d_unary = torch.from_numpy(synthetic_volume(H, W, K, seed=1 + rank)).to(dev)
        pair = "none"
    plan = TrwsPlan(1, K, N, conn.T, message_mode=0 if args.message_mode == "exact" else 1)
    # inputs live in HBM as torch tensors (plumbing only) and are bound, not copied
    d_alpha = torch.ones(E, dtype=torch.float64, device=dev)
    d_pos = torch.arange(K, dtype=torch.float64, device=dev)
    torch.cuda.synchronize()
    plan.bind_device(d_unary.data_ptr(), d_alpha.data_ptr(), 8.0, d_positions=d_pos.data_ptr(),
                     keepalive=(d_unary, d_alpha, d_pos))
    plan.stats(reset=True)  # enables per-iteration sweep timing with HIP events

    def barrier():
        D.barrier(dist, dev)

    NEVER = -1e300  # (E-LB)/E < max_relgap can fire with max_relgap = 0 once E ~ LB; time exactly K steps
    plan.iterate(args.warmup, max_relgap=NEVER)
    plan.stats(reset=True)
    plan.serial_messages(reset=True)
    barrier()
    t0 = time.perf_counter()
    done, _ = plan.iterate(args.steps, max_relgap=NEVER)  # returns after the last scalars reached the host
    assert done == args.steps, (done, args.steps)
    barrier()
    dt = time.perf_counter() - t0
    # every rank ran `steps` iterations on its own image pair: whole-job rate = all units / slowest rank
    rate, dt = D.throughput(dist, args.steps, dt, dev)
    sweep_ms, sweep_launches = plan.stats()
    serial_msgs = plan.serial_messages()
    _, energy, lb, iters = plan.result(want_labels=False)
    # labelled extra, never `value`: the same volume with the nodes visited in index order
    # (STEREO_TRWS_ORDER_INDEX: MRFEnergy without SetAutomaticOrdering; H + W - 1 dependency levels,
    # no serial border chain) -- a valid TRW-S schedule whose results are not the gateway's
    from stereo_amd.trws import ORDER_INDEX
    alt_dt = alt_en = alt_lb = float("nan")
    if not args.no_extras:
        alt = TrwsPlan(1, K, N, conn.T, message_mode=(0 if args.message_mode == "exact" else 1) | ORDER_INDEX)
        alt.bind_device(d_unary.data_ptr(), d_alpha.data_ptr(), 8.0, d_positions=d_pos.data_ptr())
        alt.iterate(args.warmup, max_relgap=NEVER)
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        alt.iterate(args.steps, max_relgap=NEVER)
        alt_dt = time.perf_counter() - t1
        _, alt_en, alt_lb, _ = alt.result(want_labels=False)
        alt.close()

    if rank == 0:
        a = analyze(N, conn.T)
        nf = np.diff(a["fwd_ptr"]).astype(np.float64)
        nb = np.diff(a["bwd_ptr"]).astype(np.float64)
        bytes_pair = algorithmic_bytes_per_sweep_pair((nf, nb), K)
        launches_per_iter = sweep_launches / max(args.steps, 1)
        bytes_per_launch = bytes_pair / launches_per_iter
        avg_launch_s = (sweep_ms * 1e-3) / max(sweep_launches, 1)
        achieved = bytes_per_launch / avg_launch_s / 1e9
        # HBM traffic per launch from the committed PMC passes of this same workload (profiles/),
        # corrected as MI355X_MICROARCH.md prescribes; null if the workload differs from them
        # (only this round's passes count: a shape or volume without one reports null and says why)
        traffic = None
        pmc = {(375, 450, 60, "ncc", "teddy"): "%s_trws_teddy60_pmc_hbm.json" % PROFILE_ROUND,
               (2000, 3000, 256, "noise", "none"): "%s_trws_wide256_3000x2000_pmc_hbm.json" % PROFILE_ROUND}.get((H, W, K, volume, pair))
        traffic_source = "no PMC pass of this workload (%dx%dx%d, %s volume, %s pair) is committed for %s" % (W, H, K, volume, pair, PROFILE_ROUND)
        if pmc and os.path.exists(os.path.join(ROOT, "profiles", pmc)):
            traffic = json.load(open(os.path.join(ROOT, "profiles", pmc)))["per_launch"]["hbm_bytes_corrected"]
            traffic_source = "profiles/%s (committed rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this workload; NOT measured in this run)" % pmc
        elif pmc:
            traffic_source = "profiles/%s not committed yet" % pmc
        spec_stats = plan.spec_stats()
        out = {
            "metric": "TRW-S fusion iterations/sec, 450x375x60 labels",
            "value": rate,
            "unit": "iterations/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f64",
            "data": ("teddy pair (the reference's data/teddy/im2.png, im6.png, committed as tests/golden/teddy_pair.npz)"
                     if pair == "teddy" else "synthetic"),
            "config": {"workload": "configs[1]: %s, %dx%dx%d labels, TRW-S (trws.m drop-in), "
                                   "kernel 1, tol 8, alphas 1, fronto-parallel labels" % (
                                       "NCC cost volume (5x5xRGB, unary weight 40) of %s" % (
                                           "the Teddy pair" if pair == "teddy" else "a synthetic image pair") if volume == "ncc"
                                       else "synthetic cost volume (planted signal + noise)", W, H, K),
                       "nodes": N, "directed_edges": E, "message_mode": "exact (reference envelope)" if args.message_mode == "exact" else "minplus",
                       "parallelism": "independent image pair per GPU" if world > 1 else "1 GPU"},
            "ranks_seen": ranks_seen, "distinct_devices": len({(r["host"], r["device"]) for r in ranks_seen}),
            "collective_backend": (dist.get_backend() if dist is not None and world > 1 else "none"),
            "serial_envelope_messages": serial_msgs,
            "serial_envelope_fraction": serial_msgs / (2.0 * E * max(args.steps, 1)),  # 2E message updates per iteration
            "final_energy": energy, "final_lower_bound": lb, "iterations_done": iters,
            "speculative_chain": dict(spec_stats, what="the border chain's schedule (DESIGN 4.5): a runner computes the chain's min-plus recurrence ahead, "
                                      "segments of 16 visits recompute every visit certified, side by side, and commit in order after comparing "
                                      "what they started from; second_walks = segments walked again because the runner's row was not the "
                                      "reference's; STEREO_HIP_TRWS_SPEC=0 gives the plain schedule, same bits"),
            "index_order_option": {"iterations_per_s": args.steps / alt_dt, "ms_per_step": alt_dt / args.steps * 1e3,
                                   "energy": alt_en, "lower_bound": alt_lb,
                                   "note": "STEREO_TRWS_ORDER_INDEX on rank 0's volume: node index order (MRFEnergy without "
                                           "SetAutomaticOrdering), H+W-1 dependency levels; NOT the gateway's labels -- "
                                           "reported next to the headline, never as `value`"},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "traffic_source": traffic_source,
                         "kernel": ({4: "trws_pipe2_kernel", 3: "trws_wide_kernel", 2: "trws_pipe_kernel", 1: "trws_persistent_kernel", 0: "trws_sweep_kernel (per level)"}[plan.path()]
                                    if not spec_stats["active"] else {3: "trws_wide_spec_kernel", 2: "trws_pipe_spec_kernel"}[plan.path()])
                                   + " (one persistent launch per sweep)", "bytes_per_launch": bytes_per_launch,
                         "avg_launch_us": avg_launch_s * 1e6, "launches_per_step": launches_per_iter},
        }
        if args.no_extras:
            out.pop("index_order_option")
        if world == 1 and not args.no_cpu_baseline:  # reported baseline: rank 0 at N=1 only
            from oracle import pyoracle
            q = np.tile(np.arange(K, dtype=np.float64), (E, 1))
            ci = max(args.cpu_iters, 1)
            unary = d_unary.cpu().numpy()
            r = pyoracle.trws(1, unary, conn, q, q, np.ones(E), 8.0, maxiter=ci, max_relgap=-1e300,
                              mode=1, want_trace=True)
            secs = float(r[4][-1, 2])
            # "+ final energy vs ref": the same ci iterations from zero messages on the GPU, outside the
            # timed region, against the oracle run that is being timed anyway
            plan.reset()
            plan.iterate(ci, max_relgap=-1e300)
            g_lab, g_en, g_lb, _ = plan.result(want_labels=True)
            out["energy_vs_ref"] = {"iterations": ci, "labels_equal": bool(np.array_equal(g_lab, r[0])),
                                    "energy_equal": bool(g_en == r[1]), "lower_bound_equal": bool(g_lb == r[2]),
                                    "energy": g_en, "energy_ref": float(r[1]), "lower_bound": g_lb,
                                    "lower_bound_ref": float(r[2]), "ref": "oracle/trws_oracle.c (envelope messages)"}
            out["cpu_baseline"] = {"value": ci / secs, "unit": "iterations/s", "cores": 1,
                                   "kind": "port", "host_cores": os.cpu_count(),
                                   "sample": "%d iterations of the same %dx%dx%d volume, oracle/trws_oracle.c "
                                             "(envelope messages), setup excluded" % (ci, W, H, K),
                                   "note": "the port is ~3x faster than the reference mex it stands for (BASELINE.md section 2: "
                                           "the real trws_mex probed at ~0.25 it/s on this workload) -- value / cpu_baseline is "
                                           "NOT the speed-up over the reference"}
            if not args.no_cpu_all_cores:
                try:
                    out["cpu_baseline"]["all_cores"] = cpu_all_cores(unary, conn, K, min(ci, 3))
                except Exception as exc:
                    out["cpu_baseline"]["all_cores"] = {"error": "%s: %s" % (type(exc).__name__, exc)}
        out["fp32_variant"] = ("not offered: SURVEY 8(d) allows fp32 messages for configs[3]/[4]; measured in round 4 "
                               "(profiles/r04_fp32_cones.txt: -9 % per iteration at 3000x2000x256, bound moves by 4e-10, and no "
                               "exactness certificate exists in fp32, so every message would take the serial construction); "
                               "one exact f64 path is kept instead of a second inexact one")
        if world == 1 and not args.no_cpu_baseline:
            # the drop-in call itself: stereo_trws / stereo_rd exactly as trws.m:33 / rd.m:21 hand the
            # arrays over -- HOST K x E q and qprim, K x N unary, labels back -- set-up, transfers and everything included
            try:
                h_un = np.asfortranarray(d_unary.cpu().numpy().T)                       # K x N
                h_q = np.asfortranarray(np.tile(np.arange(K, dtype=np.float64)[:, None], (1, E)))   # K x E, as dispmap_super.m:177-183 builds it
                conn1 = conn.T + 1
                _lib.lib().stereo_trws_cache_clear()
                tb = []
                for _ in range(3):
                    t1 = time.perf_counter()
                    bl, be, bb, bi = stereo_amd.trws(1, h_un, conn1, h_q, h_q, np.ones(E), 8.0, dict(maxiter=args.steps, max_relgap=NEVER))
                    tb.append(time.perf_counter() - t1)
                _lib.lib().stereo_trws_cache_clear()
                os.environ["STEREO_HIP_TRWS_CACHE"] = "0"
                try:
                    t1 = time.perf_counter()
                    ul, ue, ub, ui = stereo_amd.trws(1, h_un, conn1, h_q, h_q, np.ones(E), 8.0, dict(maxiter=args.steps, max_relgap=NEVER))
                    t_unc = time.perf_counter() - t1
                finally:
                    del os.environ["STEREO_HIP_TRWS_CACHE"]   # (never left behind for the later legs)
                resident_s = dt
                out["trws_boundary"] = {
                    "what": "stereo_trws (what trws_mex reaches) with HOST arrays as trws.m:33 hands them: unary K x N (%.0f MB), q and "
                            "qprim K x E (%.0f MB each), %d iterations, labels back; python-side checks of the binding included"
                            % (h_un.nbytes / 1e6, h_q.nbytes / 1e6, args.steps),
                    "first_call_s": tb[0], "second_call_s": tb[1], "third_call_s": tb[2],
                    "uncached_call_s": t_unc, "uncached_what": "STEREO_HIP_TRWS_CACHE=0: plan per call, K x E arrays uploaded (round 4's path)",
                    "resident_plan_s_for_the_same_iterations": resident_s,
                    "second_call_over_resident": tb[1] / resident_s,
                    "iterations_per_s_second_call": args.steps / tb[1],
                    "equal_to_uncached": bool(np.array_equal(bl, ul) and be == ue and bb == ub and bi == ui)}
            except Exception as exc:
                out["trws_boundary"] = {"error": "%s: %s" % (type(exc).__name__, exc)}
            # secondary figure: QPBO binary fusion moves/s on a Teddy-sized move (host arrays in,
            # labels out -- PCIe inclusive), next to the reference QPBO library where it travelled
            try:
                from helpers import fusion_problem
                from stereo_amd.rd import RdPlan
                fp = fusion_problem(5, H, W, kernel=1, tol=8.0)
                fargs = (fp["U0"], fp["U1"], fp["E00"], fp["E01"], fp["E10"], fp["E11"])
                rp = RdPlan(N, fp["conn"].T, grid=(H, W))
                rp.solve(*fargs)
                t1 = time.perf_counter()
                for _ in range(10):
                    flab, fen, flb, fnu = rp.solve(*fargs)
                tf = (time.perf_counter() - t1) / 10
                extra = {"moves_per_s": 1.0 / tf, "ms_per_move": tf * 1e3, "grid": "%dx%d" % (W, H),
                         "energy": fen, "lower_bound": flb, "unlabelled": fnu, "includes": "H2D of the six term arrays + D2H of labels"}
                # ... and the gateway entry stereo_rd itself (rd.m:21): first call builds the plan, later calls reuse it
                try:
                    _lib.lib().stereo_rd_cache_clear()
                    trd = []
                    for _ in range(4):
                        t1 = time.perf_counter()
                        glab = stereo_amd.rd(*fargs, fp["conn"].T + 1, {})[0]
                        trd.append(time.perf_counter() - t1)
                    extra["rd_boundary"] = {"what": "stereo_rd (what rd_mex reaches) with host arrays as rd.m:21 hands them, labels back",
                                            "first_call_ms": trd[0] * 1e3, "cached_call_ms": min(trd[1:]) * 1e3,
                                            "labels_equal_to_plan": bool(np.array_equal(glab, flab))}
                except Exception as exc:
                    extra["rd_boundary"] = {"error": "%s: %s" % (type(exc).__name__, exc)}
                from oracle import pyoracle
                if pyoracle.have_ref_qpbo():
                    t1 = time.perf_counter()
                    rlab, ren, rlb, rnu = pyoracle.ref_rd(*fargs, fp["conn"])
                    tr_ = time.perf_counter() - t1
                    extra["cpu_reference"] = {"moves_per_s": 1.0 / tr_, "kind": "reference", "cores": 1,
                                              "labels_equal": bool(np.array_equal(rlab, flab))}
                # whole fusion moves through the mirrored class (dispmap_super.m:61-84 + update_energy):
                # NCC volume, assignment and all terms resident in HBM, one proposal uploaded per move
                im0, im1 = synthetic_pair(H, W, K)
                dm = stereo_amd.dispmap_ncc([im0, im1], np.arange(K, dtype=np.float64), 1, 40.0, 8.0)
                planes = [np.stack([np.zeros(N), np.zeros(N), np.ones(N), np.full(N, -float(d))])
                          for d in np.linspace(2, K - 3, 9)]
                dm.binary_fusion(planes[0])
                t1 = time.perf_counter()
                for pl in planes[1:]:
                    dm.binary_fusion(pl)
                tm = (time.perf_counter() - t1) / (len(planes) - 1)
                extra["end_to_end"] = {"moves_per_s": 1.0 / tm, "ms_per_move": tm * 1e3, "energy": dm.energy(),
                                       "what": "dispmap_ncc.binary_fusion on a synthetic %dx%d pair, %d fronto-parallel "
                                               "proposals: pairwise terms + unaries + QPBO + scatter + energy on the device" % (W, H, len(planes) - 1)}
                # the same moves with the proposals built on the device from their plane: 32 bytes
                # cross PCIe per move instead of 4 x N doubles
                dm.restart()
                pps = [stereo_amd.PlaneProposal([0.0, 0.0, 1.0, -float(d)]) for d in np.linspace(2, K - 3, 9)]
                dm.binary_fusion(pps[0])
                t1 = time.perf_counter()
                for pl in pps[1:]:
                    dm.binary_fusion(pl)
                tm2 = (time.perf_counter() - t1) / (len(pps) - 1)
                extra["end_to_end_device_proposals"] = {"moves_per_s": 1.0 / tm2, "ms_per_move": tm2 * 1e3, "energy": dm.energy(),
                                                        "energy_equal_to_end_to_end": bool(dm.energy() == extra["end_to_end"]["energy"])}
                # hard moves: example_global.m's workload -- dispmap_globalstereo, 14 piecewise-planar proposals, QPBO with
                # weak persistency + Improve (most moves leave nodes unlabelled) -- on the synthetic pair (block-wise random
                # planes, colour-difference edge weights) and on the reference's own Teddy pair with the inputs a MATLAB user
                # has (mean-shift edge weights and SegPln proposals on the reference's own segmentation maps, tests/golden/)
                for key, which in (("hard_moves", "synthetic"), ("hard_moves_teddy", "teddy")):
                    try:
                        extra[key] = hard_moves_leg(which, H, W, im0)
                    except Exception as exc:
                        extra[key] = {"error": "%s: %s" % (type(exc).__name__, exc)}
                out["binary_fusion"] = extra
                try:
                    out["segmentation"] = segmentation_leg()
                except Exception as exc:
                    out["segmentation"] = {"error": "%s: %s" % (type(exc).__name__, exc)}
            except Exception as exc:  # the headline number must not depend on the secondary one
                out["binary_fusion"] = {"error": str(exc)}
    # second leg: one large image tiled across all ranks (strong scaling); every rank takes part
    scale = None
    if not args.no_scale:
        del plan, d_unary, d_alpha, d_pos
        torch.cuda.empty_cache()
        try:
            if world > 1:   # the top-level figure at N > 1: exactly --steps timed iterations after --warmup
                scale = scale_leg(args, rank, local_rank, world, dist, dev, steps=args.steps, warmup=args.warmup)
            else:
                scale = scale_leg(args, rank, local_rank, world, dist, dev)
        except Exception as exc:  # the headline line must not depend on the scaling leg
            scale = {"error": "%s: %s" % (type(exc).__name__, exc), "n_gpus": world}
        if world > 1 and not args.no_scale_n1:
            # the curve's first point in the same job: rank 0 alone, the plain single-GPU plan on the same image
            n1 = None
            if rank == 0:
                try:
                    torch.cuda.empty_cache()
                    r1 = scale_leg(args, 0, local_rank, 1, None, dev, steps=args.steps, warmup=args.warmup)
                    n1 = {k: r1[k] for k in ("value", "unit", "ms_per_iteration", "steps", "warmup", "hbm_frac_per_gpu",
                                             "energy", "lower_bound", "label_sum")}
                    if isinstance(scale, dict) and "value" in scale:   # same image, same iteration count: the strips' answer
                        n1["strips_label_sum_equal"] = bool(scale["label_sum"] == r1["label_sum"])
                        n1["strips_energy_rel_diff"] = abs(scale["energy"] - r1["energy"]) / abs(r1["energy"])
                        n1["strips_lower_bound_rel_diff"] = abs(scale["lower_bound"] - r1["lower_bound"]) / abs(r1["lower_bound"])
                        n1["speedup_over_one_gpu"] = scale["value"] / r1["value"]
                except Exception as exc:
                    n1 = {"error": "%s: %s" % (type(exc).__name__, exc)}
                if isinstance(scale, dict):
                    scale["n1_same_run"] = n1
            D.barrier(dist, dev)
        # the same image and strips under the two labelled options (and both): not the reference's bits
        notes = {"minplus_option": "STEREO_TRWS_MESSAGES_MINPLUS: windowed min-plus messages, no certificate / envelope "
                                   "construction (equal to the reference unless a certificate would have failed)",
                 "index_order_option": "STEREO_TRWS_ORDER_INDEX: not the gateway's node order / labels",
                 "index_order_minplus_option": "both options"}
        for key, io, mp in (("minplus_option", False, True), ("index_order_option", True, False),
                            ("index_order_minplus_option", True, True)):
            if world > 1 and not args.scale_options:
                break
            try:
                torch.cuda.empty_cache()
                alt_scale = scale_leg(args, rank, local_rank, world, dist, dev, index_order=io, minplus=mp)
                if rank == 0 and scale is not None and alt_scale is not None:
                    scale[key] = {k: alt_scale[k] for k in ("value", "unit", "ms_per_iteration", "hbm_frac_per_gpu",
                                                            "energy", "lower_bound")}
                    scale[key]["note"] = notes[key]
                    if mp:  # did plain min-plus give the exact messages' bits on this volume?
                        ref = scale["index_order_option"] if io else scale
                        scale[key]["energy_and_bound_equal_to_exact_messages"] = bool(
                            ref.get("energy") == alt_scale["energy"] and ref.get("lower_bound") == alt_scale["lower_bound"])
            except Exception as exc:
                if rank == 0 and scale is not None:
                    scale[key] = {"error": "%s: %s" % (type(exc).__name__, exc)}
    if rank == 0:
        if scale is not None:
            out["scale"] = scale
        if pre is not None:
            out["preflight"] = pre
        if world > 1:
            # N > 1: the line IS the strong-scaling figure of configs[3]; the one-pair-per-GPU Teddy figure moves to `replicas`
            rep = {k: out[k] for k in ("metric", "value", "unit", "ms_per_step", "steps", "warmup", "data", "config", "roofline",
                                       "final_energy", "final_lower_bound", "serial_envelope_fraction") if k in out}
            rep["scaling"] = "weak"
            rep["note"] = "N INDEPENDENT Teddy problems, one per GPU (no communication): replica throughput, not the scaling curve"
            out["replicas"] = rep
            for k in ("index_order_option", "final_energy", "final_lower_bound", "iterations_done", "serial_envelope_messages",
                      "serial_envelope_fraction"):
                out.pop(k, None)
            sc = scale if isinstance(scale, dict) else {}
            good = "value" in sc
            Hs, Ws, Ks = args.scale_height, args.scale_width, args.scale_labels
            out.update({
                "metric": "TRW-S iterations/sec, ONE %dx%dx%d-label image tiled over %d GPUs (configs[3])" % (Ws, Hs, Ks, world),
                "value": sc.get("value") if good else None, "unit": "iterations/s",
                "ms_per_step": sc.get("ms_per_iteration") if good else None,
                "scaling": "strong", "data": "synthetic",
                "config": {"workload": sc.get("workload", "configs[3]"), "nodes": Hs * Ws, "directed_edges": 4 * Hs * Ws - 2 * (Hs + Ws),
                           "message_mode": "exact (reference envelope)", "parallelism": "%d row strips, one per GPU" % world},
                "preflight_ok": bool(pre and pre.get("ok")),
            })
            if good:
                fr = sc["hbm_frac_per_gpu"]
                out["roofline"] = {"bound": "hbm", "achieved": fr[0] * HBM_PEAK_GBS, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                   "frac": fr[0], "traffic": None, "traffic_source": None,
                                   "kernel": "trws_wide_kernel (one persistent launch per sweep and GPU)",
                                   "per_gpu": True, "frac_min_max_over_gpus": fr,
                                   "bytes_per_launch": sc.get("bytes_per_launch_per_gpu"),
                                   "avg_launch_us": sc.get("sweep_launch_ms_rank0", 0.0) * 1e3, "launches_per_step": 2.0}
            else:
                out["scale_error"] = sc.get("error", "scale leg skipped (--no-scale)")
        if world > 1 and not args.no_gateway_leg:
            # the gateway sharding by itself over the node's devices: a process of its own (the ranks' work is done; whatever
            # happens there -- this path has only ever run on logical strips of one device -- the line above stands)
            import subprocess
            env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT", "GROUP_RANK",
                                                                     "LOCAL_WORLD_SIZE", "ROLE_RANK", "ROLE_WORLD_SIZE", "TORCHELASTIC_RUN_ID")}
            try:
                pr = subprocess.run([sys.executable, os.path.abspath(__file__), "--gateway-leg", str(world)], env=env, capture_output=True, text=True, timeout=420)
                lines = [l for l in pr.stdout.strip().splitlines() if l.startswith("{")]
                out["gateway"] = json.loads(lines[-1]) if lines else {"error": "exit %d: %s" % (pr.returncode, pr.stderr.strip()[-400:])}
            except Exception as exc:
                out["gateway"] = {"error": "%s: %s" % (type(exc).__name__, exc)}
        print(json.dumps(out))
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
