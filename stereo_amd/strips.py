"""Row-strip TRW-S: one image tiled across the GPUs of a node (stereo_trws_plan_create_strip & co).

No reference counterpart -- the reference is one serial sweep (cpp/trw-s/minimize.cpp:36-95) in the
order of cpp/trw-s/ordering.cpp:42-152.  Strip g visits the nodes of its band of rows in exactly
that order's dependency DAG; a boundary node's messages, flag and label are written straight into
the neighbouring strip's arrays by the visiting workgroup (peer stores over xGMI when the
neighbour is another GPU).  Labels are bit-identical to a single plan; energy and lower bound are
per-strip partial sums added in strip order (the only cross-strip reduction: two doubles).

Two drivers over the same C ABI:
  * ``TrwsStrips``      all strips in THIS process (logical strips on one GPU -- the parity tests --
                        or one process driving several GPUs with peer access);
  * ``TrwsStripRank``   one strip per process / GPU (``bench.py --gpus N``): HIP IPC handles are
                        exchanged through torch.distributed, partial sums reduced with all_gather.
"""
import ctypes as C

import numpy as np

from . import _lib
from ._lib import StereoHipError
from .trws import _ptr, MESSAGES_EXACT

IPC_BYTES = 256  # STEREO_TRWS_IPC_BYTES


def row_strip_owner(H, W, nstrips):
    """owner[node] for `nstrips` bands of rows of an H x W image (node id = col*H + row,
    dispmap_super.m:281-282); band g owns rows [g*H/G, (g+1)*H/G)."""
    if nstrips < 1 or nstrips > H:
        raise StereoHipError("need 1 <= nstrips <= image height")
    rows = np.arange(H, dtype=np.int64)
    band = np.minimum(rows * nstrips // H, nstrips - 1).astype(np.int32)
    return np.ascontiguousarray(np.tile(band, W))


class _StripPlan:
    """Handle on one strip's plan (ctypes)."""

    def __init__(self, kernel, K, N, conn_f, owner, nstrips, strip, message_mode, max_workgroups, share):
        self.K, self.N, self.E = int(K), int(N), int(conn_f.shape[1])
        self.nstrips, self.strip = int(nstrips), int(strip)
        self._h = C.c_void_p()
        self._keep = []
        err = _lib.errbuf()
        rc = _lib.lib().stereo_trws_plan_create_strip(
            C.c_int(int(kernel)), C.c_int(self.K), C.c_int64(self.N), C.c_int64(self.E),
            _ptr(conn_f, C.c_uint32), C.c_int(message_mode),
            _ptr(owner, C.c_int32) if owner is not None else None, C.c_int(self.nstrips),
            C.c_int(self.strip), C.c_int(int(max_workgroups)), share._h if share is not None else None,
            C.byref(self._h), err, C.c_size_t(len(err)))
        _lib.check(rc, err)

    def close(self):
        if self._h:
            _lib.lib().stereo_trws_plan_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def upload(self, unary, alphas, tol, q=None, qprim=None, positions=None):
        f = lambda a: np.asfortranarray(a, dtype=np.float64)
        unary = f(unary)
        alphas = f(np.asarray(alphas, dtype=np.float64).reshape(-1))
        assert unary.shape == (self.K, self.N) and alphas.shape[0] == self.E
        pq = pqp = ppos = None
        if q is not None:
            q = f(q); qprim = f(qprim)
            pq, pqp = _ptr(q), _ptr(qprim)
        else:
            positions = f(np.asarray(positions, dtype=np.float64).reshape(-1))
            ppos = _ptr(positions)
        err = _lib.errbuf()
        rc = _lib.lib().stereo_trws_plan_upload(self._h, _ptr(unary), pq, pqp, ppos, _ptr(alphas),
                                                C.c_double(float(tol)), err, C.c_size_t(len(err)))
        _lib.check(rc, err)

    def bind_device(self, d_unary, d_alphas, tol, d_q=None, d_qprim=None, d_positions=None, keepalive=()):
        self._keep = list(keepalive)
        vp = lambda x: C.c_void_p(int(x)) if x else None
        err = _lib.errbuf()
        rc = _lib.lib().stereo_trws_plan_bind_device(self._h, vp(d_unary), vp(d_q), vp(d_qprim), vp(d_positions),
                                                     vp(d_alphas), C.c_double(float(tol)), err, C.c_size_t(len(err)))
        _lib.check(rc, err)

    def bind_device_strip(self, d_unary, d_alphas, tol, d_q=None, d_qprim=None, d_positions=None, keepalive=()):
        """Like bind_device with arrays that hold the strip's rows only (order: layout())."""
        self._keep = list(keepalive)
        vp = lambda x: C.c_void_p(int(x)) if x else None
        err = _lib.errbuf()
        rc = _lib.lib().stereo_trws_plan_bind_device_strip(self._h, vp(d_unary), vp(d_q), vp(d_qprim), vp(d_positions),
                                                           vp(d_alphas), C.c_double(float(tol)), err, C.c_size_t(len(err)))
        _lib.check(rc, err)

    def layout(self):
        """(nodes, n_own, edges): global ids of the strip's local nodes (own ones first) and edges."""
        nn, no, ne = C.c_int64(), C.c_int64(), C.c_int64()
        _lib.lib().stereo_trws_plan_strip_layout(self._h, C.byref(nn), C.byref(no), C.byref(ne), None, None)
        nodes = np.zeros(nn.value, dtype=np.int32)
        edges = np.zeros(ne.value, dtype=np.int32)
        _lib.lib().stereo_trws_plan_strip_layout(self._h, None, None, None, nodes.ctypes.data_as(C.c_void_p),
                                                 edges.ctypes.data_as(C.c_void_p))
        return nodes, int(no.value), edges

    def reset(self):
        err = _lib.errbuf()
        _lib.check(_lib.lib().stereo_trws_plan_reset(self._h, err, C.c_size_t(len(err))), err)

    def connect(self, which, peer):
        err = _lib.errbuf()
        _lib.check(_lib.lib().stereo_trws_plan_connect(self._h, C.c_int(which), peer._h, err, C.c_size_t(len(err))), err)

    def ipc_export(self):
        buf = C.create_string_buffer(IPC_BYTES)
        err = _lib.errbuf()
        _lib.check(_lib.lib().stereo_trws_plan_ipc_export(self._h, buf, C.c_size_t(IPC_BYTES), err, C.c_size_t(len(err))), err)
        return bytes(buf.raw)

    def ipc_connect(self, which, handles):
        buf = C.create_string_buffer(bytes(handles), IPC_BYTES)
        err = _lib.errbuf()
        _lib.check(_lib.lib().stereo_trws_plan_ipc_connect(self._h, C.c_int(which), buf, err, C.c_size_t(len(err))), err)

    def issue(self, stream=None):
        err = _lib.errbuf()
        _lib.check(_lib.lib().stereo_trws_plan_issue(self._h, C.c_void_p(int(stream)) if stream else None, err,
                                                     C.c_size_t(len(err))), err)

    def collect(self):
        lb, en = C.c_double(), C.c_double()
        err = _lib.errbuf()
        _lib.check(_lib.lib().stereo_trws_plan_collect(self._h, C.byref(lb), C.byref(en), err, C.c_size_t(len(err))), err)
        return lb.value, en.value

    def commit(self, lb, energy):
        err = _lib.errbuf()
        _lib.check(_lib.lib().stereo_trws_plan_commit(self._h, C.c_double(lb), C.c_double(energy), err,
                                                      C.c_size_t(len(err))), err)

    def labels(self):
        lab = np.zeros(self.N)
        en, lb, it = C.c_double(), C.c_double(), C.c_double()
        err = _lib.errbuf()
        _lib.check(_lib.lib().stereo_trws_plan_result(self._h, _ptr(lab), C.byref(en), C.byref(lb), C.byref(it), err,
                                                      C.c_size_t(len(err))), err)
        return lab

    def info(self):
        ns, st, nf, nb, a, b = C.c_int(), C.c_int(), C.c_int64(), C.c_int64(), C.c_int(), C.c_int()
        own = C.c_int64()
        _lib.lib().stereo_trws_plan_strip_info(self._h, C.byref(ns), C.byref(st), C.byref(own), C.byref(nf), C.byref(nb),
                                               C.byref(a), C.byref(b))
        return dict(nstrips=ns.value, strip=st.value, own_nodes=own.value, runs_forward=nf.value,
                    runs_backward=nb.value, needs_previous=bool(a.value), needs_next=bool(b.value))

    def path(self):
        return int(_lib.lib().stereo_trws_plan_path(self._h))

    def serial_messages(self, reset=False):
        n = C.c_int64()
        _lib.lib().stereo_trws_plan_counters(self._h, C.byref(n), C.c_int(int(reset)))
        return n.value

    def stats(self, reset=False):
        ms, n = C.c_double(), C.c_int64()
        _lib.lib().stereo_trws_plan_stats(self._h, C.byref(ms), C.byref(n), C.c_int(int(reset)))
        return ms.value, n.value


def _conn_f(connectivity0):
    c = np.asarray(connectivity0)
    if c.ndim != 2 or c.shape[0] != 2:
        raise StereoHipError("connectivity must be 2 x E")
    return np.asfortranarray(c, dtype=np.uint32)


def _relgap(en, lb):
    """(E - LB) / E as minimize.cpp:105 computes it in C: E == 0 gives inf or NaN, never an exception."""
    with np.errstate(divide="ignore", invalid="ignore"):
        return float(np.float64(en - lb) / np.float64(en))


def _combine(parts):
    """Partial (lb, energy) sums added in strip order: deterministic, independent of arrival order."""
    lb = en = 0.0
    for l, e in parts:
        lb += l
        en += e
    return lb, en


class TrwsStrips:
    """All strips of one problem in this process (same call surface as TrwsPlan where it matters).

    On ONE GPU the strips' persistent launches must all be resident together: every strip gets
    `workgroups_per_strip` workgroups (default: the device's CU count // nstrips)."""

    def __init__(self, kernel, K, N, connectivity0, owner, nstrips, message_mode=MESSAGES_EXACT,
                 workgroups_per_strip=None, devices=None):
        conn = _conn_f(connectivity0)
        owner = np.ascontiguousarray(owner, dtype=np.int32)
        if owner.shape != (N,):
            raise StereoHipError("owner must have one entry per node")
        self.K, self.N, self.E, self.nstrips = int(K), int(N), int(conn.shape[1]), int(nstrips)
        self._owner = owner
        self._devices = list(devices) if devices else None
        if workgroups_per_strip is None:
            # strips that share a device share its CUs: one workgroup per CU is what is certain to be
            # resident, and every strip's launch must be resident together with its neighbours'
            cus = int(_lib.lib().stereo_hip_device_cus()) or 256
            workgroups_per_strip = 0 if devices else max(2, cus // self.nstrips)
        self.plans = []
        for g in range(self.nstrips):
            if devices:
                _lib.check(_lib.lib().stereo_hip_set_device(C.c_int(int(devices[g]))), None)
            self.plans.append(_StripPlan(kernel, K, N, conn, owner if g == 0 else None, nstrips, g, message_mode,
                                         workgroups_per_strip, self.plans[0] if g else None))
        for g in range(self.nstrips):
            if g > 0:
                self.plans[g].connect(0, self.plans[g - 1])
            if g + 1 < self.nstrips:
                self.plans[g].connect(1, self.plans[g + 1])
        self.energy = self.lb = 0.0
        self.iterations = 0

    def close(self):
        for p in self.plans:
            p.close()

    def upload(self, unary, alphas, tol, q=None, qprim=None, positions=None):
        for p in self.plans:
            p.upload(unary, alphas, tol, q=q, qprim=qprim, positions=positions)
        self.iterations = 0

    def bind_device(self, d_unary, d_alphas, tol, d_q=None, d_qprim=None, d_positions=None, keepalive=()):
        for p in self.plans:
            p.bind_device(d_unary, d_alphas, tol, d_q=d_q, d_qprim=d_qprim, d_positions=d_positions, keepalive=keepalive)
        self.iterations = 0

    def reset(self):
        for p in self.plans:
            p.reset()
        self.iterations = 0

    def iterate(self, iters, max_relgap=0.0):
        """Mirrors TrwsPlan.iterate: returns (iterations done in this call, stop test fired)."""
        done = 0
        for _ in range(int(iters)):
            if self._devices:   # one launch per strip, each on its own GPU
                for p in self.plans:
                    p.issue()
            else:               # strips share the device: one fused launch per sweep
                handles = (C.c_void_p * self.nstrips)(*[p._h for p in self.plans])
                err = _lib.errbuf()
                _lib.check(_lib.lib().stereo_trws_plans_issue(handles, C.c_int(self.nstrips), None, err,
                                                              C.c_size_t(len(err))), err)
            lb, en = _combine([p.collect() for p in self.plans])
            for p in self.plans:
                p.commit(lb, en)
            self.lb, self.energy = lb, en
            self.iterations += 1
            done += 1
            if _relgap(en, lb) < max_relgap:  # minimize.cpp:105
                return done, True
        return done, False

    def result(self, want_labels=True):
        lab = None
        if want_labels:
            # every strip holds the labels of its own nodes (and of its neighbours' boundary rows)
            owner = self._owner
            lab = np.zeros(self.N)
            for g, p in enumerate(self.plans):
                lg = p.labels()
                lab[owner == g] = lg[owner == g]
        return lab, self.energy, self.lb, float(self.iterations)

    def path(self):
        return self.plans[0].path()

    def serial_messages(self, reset=False):
        return sum(p.serial_messages(reset) for p in self.plans)


def make_strips(kernel, K, H, W, connectivity0, nstrips, **kw):
    """TrwsStrips over `nstrips` bands of rows of an H x W image."""
    return TrwsStrips(kernel, K, H * W, connectivity0, row_strip_owner(H, W, nstrips), nstrips, **kw)


class TrwsStripRank:
    """One strip per process (rank g of `world` owns band g).  `dist` is torch.distributed with an
    initialised process group; its only uses are the exchange of the IPC handles at start-up and
    the all_gather of two doubles per iteration."""

    def __init__(self, kernel, K, H, W, connectivity0, rank, world, dist, device, message_mode=MESSAGES_EXACT,
                 max_workgroups=0):
        import torch
        conn = _conn_f(connectivity0)
        self.rank, self.world, self.dist, self.device = int(rank), int(world), dist, device
        self.owner = row_strip_owner(H, W, world)
        self.N = H * W
        self.plan = _StripPlan(kernel, K, self.N, conn, self.owner, world, rank, message_mode, max_workgroups, None)
        # exchange the IPC handles of (messages, flags, labels) with both neighbours
        mine = self.plan.ipc_export()
        # (a plain all_gather of IPC_BYTES bytes per rank on the device: the same call under RCCL and
        #  under gloo, no pickling collective)
        mine_t = torch.tensor(list(mine), dtype=torch.uint8, device=device)
        all_t = [torch.zeros(IPC_BYTES, dtype=torch.uint8, device=device) for _ in range(world)]
        dist.all_gather(all_t, mine_t)
        gathered = [bytes(t.cpu().numpy().tobytes()) for t in all_t]
        if rank > 0:
            self.plan.ipc_connect(0, gathered[rank - 1])
        if rank + 1 < world:
            self.plan.ipc_connect(1, gathered[rank + 1])
        dist.barrier()
        self._buf = torch.zeros(2, dtype=torch.float64, device=device)
        self._all = [torch.zeros(2, dtype=torch.float64, device=device) for _ in range(world)]
        self.energy = self.lb = 0.0
        self.iterations = 0

    # New inputs reset the strip's state (messages, flags, labels: stereo_trws_plan_reset) -- arrays the
    # neighbouring ranks write into while THEIR kernels run.  Two barriers keep the reset apart from
    # every rank's sweeps: nobody resets while a neighbour may still be running the old problem, and
    # nobody issues the new problem's first sweep into arrays a slower neighbour has yet to wipe.
    def _quiesced(self, fn, *a, **kw):
        self.dist.barrier()
        try:
            return fn(*a, **kw)
        finally:
            self.dist.barrier()

    def bind_device(self, *a, **kw):
        self._quiesced(self.plan.bind_device, *a, **kw)
        self.iterations = 0

    def bind_device_strip(self, *a, **kw):
        self._quiesced(self.plan.bind_device_strip, *a, **kw)
        self.iterations = 0

    def upload(self, *a, **kw):
        self._quiesced(self.plan.upload, *a, **kw)
        self.iterations = 0

    def reset(self):
        self._quiesced(self.plan.reset)
        self.iterations = 0

    def iterate(self, iters, max_relgap=0.0):
        import torch
        done = 0
        for _ in range(int(iters)):
            self.plan.issue()
            lb, en = self.plan.collect()
            self._buf.copy_(torch.tensor([lb, en], dtype=torch.float64))
            self.dist.all_gather(self._all, self._buf)
            parts = torch.stack(self._all).cpu().numpy()
            lb, en = _combine([(float(r[0]), float(r[1])) for r in parts])  # strip order on every rank
            self.plan.commit(lb, en)
            self.lb, self.energy = lb, en
            self.iterations += 1
            done += 1
            if _relgap(en, lb) < max_relgap:
                return done, True
        return done, False

    def own_labels(self):
        """(node ids owned by this rank, their 1-based labels)."""
        lab = self.plan.labels()
        idx = np.nonzero(self.owner == self.rank)[0]
        return idx, lab[idx]

    def close(self):
        self.plan.close()


def schedule_strips(N, connectivity0, direction, owner, nstrips, max_resident_runs=0):
    """Host-only inspection of the strip-aware chain schedule (stereo_trws_schedule_strips)."""
    c = _conn_f(connectivity0)
    E = c.shape[1]
    owner = np.ascontiguousarray(owner, dtype=np.int32)
    i64 = lambda n: np.zeros(n, np.int64)
    rank_at, run_ptr, ticket_run, pred, dep_ptr, dep_rank = i64(N), i64(N + 1), i64(N), i64(N), i64(N + 1), i64(4 * N)
    run_strip, remote = i64(N), i64(N)
    nruns = C.c_int64(0)
    err = _lib.errbuf()
    P = lambda a: _ptr(a, C.c_int64)
    rc = _lib.lib().stereo_trws_schedule_strips(C.c_int64(N), C.c_int64(E), _ptr(c, C.c_uint32),
                                                C.c_int64(max_resident_runs), C.c_int(direction),
                                                _ptr(owner, C.c_int32), C.c_int(nstrips), P(rank_at), P(run_ptr),
                                                C.byref(nruns), P(ticket_run), P(pred), P(dep_ptr), P(dep_rank),
                                                P(run_strip), P(remote), err, C.c_size_t(len(err)))
    _lib.check(rc, err)
    R = nruns.value
    return dict(rank_at=rank_at, run_ptr=run_ptr[:R + 1], ticket_run=ticket_run[:R], pred_rank=pred,
                dep_ptr=dep_ptr, dep_rank=dep_rank[:dep_ptr[N]], run_strip=run_strip[:R], remote=remote)


def strip_layout_host(N, connectivity0, owner, nstrips, strip, direction):
    """Host-only: what a strip stores and its renumbered descriptors (stereo_trws_strip_layout_host).
    Returns dict(nodes, n_own, edges, desc (n_visits x 64 int32))."""
    c = _conn_f(connectivity0)
    E = c.shape[1]
    owner = np.ascontiguousarray(owner, dtype=np.int32)
    nn, no, ne, nv = C.c_int64(), C.c_int64(), C.c_int64(), C.c_int64()
    err = _lib.errbuf()
    f = _lib.lib().stereo_trws_strip_layout_host
    head = (C.c_int64(N), C.c_int64(E), _ptr(c, C.c_uint32), _ptr(owner, C.c_int32), C.c_int(nstrips), C.c_int(strip),
            C.c_int(direction), C.byref(nn), C.byref(no), C.byref(ne), C.byref(nv))
    _lib.check(f(*head, None, None, None, err, C.c_size_t(len(err))), err)
    nodes, edges = np.zeros(nn.value, np.int32), np.zeros(ne.value, np.int32)
    desc = np.zeros((nv.value, 64), np.int32)
    _lib.check(f(*head, _ptr(nodes, C.c_int32), _ptr(edges, C.c_int32), _ptr(desc, C.c_int32), err, C.c_size_t(len(err))), err)
    return dict(nodes=nodes, n_own=int(no.value), edges=edges, desc=desc)


def dataflow_reference(sched, analysis_in):
    """Serial model of one sweep: value[r] = mix(r, values of every incoming neighbour), in the
    order of the schedule positions.  `analysis_in[r]` = ranks of the incoming neighbours of rank r
    in this sweep direction.  What a strip-parallel execution must reproduce."""
    N = len(sched["rank_at"])
    val = np.zeros(N, dtype=np.uint64)
    done = np.zeros(N, dtype=bool)
    # any topological order gives the same values; use dependency order via repeated passes over runs
    order = _topological(sched, analysis_in)
    for r in order:
        val[r] = _mix(r, [val[x] for x in analysis_in[r]])
        done[r] = True
    return val


def _mix(r, vals):
    M = (1 << 61) - 1
    acc = (int(r) * 2654435761 + 12345) % M
    for v in vals:
        acc = (acc * 1000003 + int(v)) % M
    return np.uint64(acc)


def _topological(sched, incoming):
    N = len(sched["rank_at"])
    indeg = np.array([len(incoming[r]) for r in range(N)])
    out = [[] for _ in range(N)]
    for r in range(N):
        for x in incoming[r]:
            out[int(x)].append(r)
    stack = [r for r in range(N) if indeg[r] == 0]
    order = []
    while stack:
        r = stack.pop()
        order.append(r)
        for y in out[r]:
            indeg[y] -= 1
            if indeg[y] == 0:
                stack.append(y)
    if len(order) != N:
        raise StereoHipError("dependency cycle")
    return order


def simulate_strip(sched, strip, incoming, send, recv, workgroups=4):
    """Host-side model of ONE strip's sweep launch, with the hand-off protocol of the kernels:
    `workgroups` resident workgroups draw this strip's runs in ticket order and walk them node by
    node; a node waits for the flags of its foreign dependencies (`dep_rank`) -- those of another
    strip arrive through `recv()` -> (rank, value), blocking; after a visit whose descriptor has
    the notify bit of a neighbour set, `send(neighbour_strip, rank, value)` is called (message
    first, flag last: here one packet).  Values follow dataflow_reference, taking the predecessor
    in the run from the 'LDS hand-over' and everything else from flags.  Returns {rank: value} of
    the strip's nodes; raises if the strip cannot make progress although nothing is in flight."""
    rank_at, run_ptr, ticket_run = sched["rank_at"], sched["run_ptr"], sched["ticket_run"]
    dep_ptr, dep_rank, run_strip, remote = sched["dep_ptr"], sched["dep_rank"], sched["run_strip"], sched["remote"]
    mine = [int(k) for k in ticket_run if run_strip[int(k)] == strip]
    owner_of_rank = {}
    for k in range(len(run_strip)):
        for p in range(int(run_ptr[k]), int(run_ptr[k + 1])):
            owner_of_rank[int(rank_at[p])] = int(run_strip[k])
    val = {}            # every value this strip knows (own visits + received)
    flag = set()        # ranks whose completion flag is visible here
    expected = set()    # foreign ranks of other strips this strip waits for
    for k in mine:
        for p in range(int(run_ptr[k]), int(run_ptr[k + 1])):
            r = int(rank_at[p])
            for x in dep_rank[dep_ptr[r]:dep_ptr[r + 1]]:
                if owner_of_rank[int(x)] != strip:
                    expected.add(int(x))
    held, nxt, cur = [], 0, {}
    result = {}
    while True:
        while len(held) < workgroups and nxt < len(mine):
            k = mine[nxt]; nxt += 1
            held.append(k); cur[k] = int(run_ptr[k])
        if not held:
            break
        progressed = False
        for k in list(held):
            while cur[k] < run_ptr[k + 1]:
                r = int(rank_at[cur[k]])
                deps = [int(x) for x in dep_rank[dep_ptr[r]:dep_ptr[r + 1]]]
                if any(x not in flag for x in deps):
                    break
                pred = int(rank_at[cur[k] - 1]) if cur[k] > run_ptr[k] else None
                for x in incoming[r]:      # everything a visit reads must be here by now
                    if int(x) not in val:
                        raise StereoHipError("rank %d visited before the message of rank %d arrived "
                                             "(pred %s, deps %s)" % (r, int(x), pred, deps))
                v = _mix(r, [val[int(x)] for x in incoming[r]])
                val[r] = v; result[r] = v; flag.add(r)
                rem = int(remote[r])
                if rem & (1 << 16):
                    send(strip - 1, r, v)
                if rem & (1 << 17):
                    send(strip + 1, r, v)
                cur[k] += 1
                progressed = True
            if cur[k] >= run_ptr[k + 1]:
                held.remove(k)
        if progressed:
            continue
        if not (expected - flag):
            raise StereoHipError("strip %d is stuck with nothing left to receive" % strip)
        r, v = recv()
        val[int(r)] = v; flag.add(int(r))
    return result
