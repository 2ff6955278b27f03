"""Mirror of the reference's ``trws.m`` wrapper and a handle on the
device-resident plan API.

``trws(kernel, unary, connectivity, q, qprim, alphas, tol, options)`` has the
argument meaning of trws.m:2-33: ``unary`` is K x N, ``connectivity`` 2 x E and
ONE based (the wrapper subtracts 1 like trws.m:33), ``q`` / ``qprim`` K x E,
``alphas`` E x 1, ``options`` a dict with ``maxiter`` (default 1000) and
``max_relgap`` (default 0) as in trws_mex.cpp:40-41.  Returns
``(solution, energy, lower_bound, iterations)`` with 1-based labels.
"""
import ctypes as C

import numpy as np

from . import _lib
from ._lib import StereoHipError

MESSAGES_EXACT = 0
MESSAGES_MINPLUS = 1
ORDER_INDEX = 0x100   # OR into message_mode: node index order instead of SetAutomaticOrdering (stereo_hip.h)


def _f(a):
    """MATLAB-shaped array -> column-major float64 buffer."""
    return np.asfortranarray(a, dtype=np.float64)


def _conn0(connectivity):
    c = np.asarray(connectivity)
    if c.ndim != 2 or c.shape[0] != 2:
        raise StereoHipError("connectivity must be 2 x E")  # trws_mex.cpp:43
    if c.size and c.min() <= 0:
        raise AssertionError("connectivity must be one based (trws.m:5)")
    return np.asfortranarray(c.astype(np.int64) - 1, dtype=np.uint32)


def _ptr(a, t=C.c_double):
    return a.ctypes.data_as(C.POINTER(t))


def trws(kernel, unary, connectivity, q, qprim, alphas, tol, options=None):
    options = dict(options or {})
    maxiter = float(options.pop("maxiter", 1000))
    max_relgap = float(options.pop("max_relgap", 0))
    if options:
        raise StereoHipError("unknown option(s): %s" % ", ".join(sorted(options)))
    unary = _f(unary)
    q = _f(q)
    qprim = _f(qprim)
    alphas = _f(np.asarray(alphas, dtype=np.float64).reshape(-1))
    if np.isnan(q).any():
        raise StereoHipError("q contains NaN")        # trws.m:9-11
    if np.isnan(qprim).any():
        raise StereoHipError("qprim contains NaN")    # trws.m:13-15
    conn = _conn0(connectivity)
    K, N = unary.shape
    E = conn.shape[1]
    # trws_mex.cpp:43-52
    if not (q.shape == (K, E) and qprim.shape == (K, E)):
        raise StereoHipError("q / qprim must be K x E")
    if alphas.shape[0] != E:
        raise StereoHipError("alphas must be E x 1")
    if np.size(tol) != 1:
        raise StereoHipError("tol must be a scalar")
    kernel = int(np.int32(kernel))
    lab = np.zeros(N)
    en, lb, it = C.c_double(), C.c_double(), C.c_double()
    err = _lib.errbuf()
    rc = _lib.lib().stereo_trws(C.c_int(kernel), _ptr(unary), _ptr(conn, C.c_uint32), _ptr(q),
                                _ptr(qprim), _ptr(alphas), C.c_double(float(np.reshape(tol, -1)[0])),
                                C.c_double(maxiter), C.c_double(max_relgap), C.c_int(K),
                                C.c_int64(N), C.c_int64(E), _ptr(lab), C.byref(en), C.byref(lb),
                                C.byref(it), err, C.c_size_t(len(err)))
    _lib.check(rc, err)
    return lab, en.value, lb.value, it.value


class TrwsPlan:
    """Device-resident TRW-S solver for one connectivity (stereo_trws_plan_*)."""

    def __init__(self, kernel, K, N, connectivity0, message_mode=MESSAGES_EXACT):
        """connectivity0: 2 x E ZERO based pairs (MATLAB layout, as passed to trws_mex)."""
        c = np.asarray(connectivity0)
        if c.ndim != 2 or c.shape[0] != 2:
            raise StereoHipError("connectivity must be 2 x E")
        self._conn = np.asfortranarray(c, dtype=np.uint32)
        self.K, self.N, self.E = int(K), int(N), int(self._conn.shape[1])
        self._h = C.c_void_p()
        self._keep = []
        err = _lib.errbuf()
        rc = _lib.lib().stereo_trws_plan_create(C.c_int(int(kernel)), C.c_int(self.K),
                                                C.c_int64(self.N), C.c_int64(self.E),
                                                _ptr(self._conn, C.c_uint32), C.c_int(message_mode),
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
        """unary K x N, q/qprim K x E (MATLAB shapes) or a shared `positions` K-vector."""
        unary = _f(unary)
        alphas = _f(np.asarray(alphas, dtype=np.float64).reshape(-1))
        assert unary.shape == (self.K, self.N) and alphas.shape[0] == self.E
        pq = pqp = ppos = None
        if q is not None:
            q = _f(q); qprim = _f(qprim)
            assert q.shape == (self.K, self.E) and qprim.shape == (self.K, self.E)
            pq, pqp = _ptr(q), _ptr(qprim)
        else:
            positions = _f(np.asarray(positions, dtype=np.float64).reshape(-1))
            assert positions.shape[0] == self.K
            ppos = _ptr(positions)
        err = _lib.errbuf()
        rc = _lib.lib().stereo_trws_plan_upload(self._h, _ptr(unary), pq, pqp, ppos, _ptr(alphas),
                                                C.c_double(float(tol)), err, C.c_size_t(len(err)))
        _lib.check(rc, err)

    def bind_device(self, d_unary, d_alphas, tol, d_q=None, d_qprim=None, d_positions=None,
                    keepalive=()):
        """Device pointers (ints), e.g. torch_tensor.data_ptr(); tensors must stay alive."""
        self._keep = list(keepalive)
        vp = lambda x: C.c_void_p(int(x)) if x else None
        err = _lib.errbuf()
        rc = _lib.lib().stereo_trws_plan_bind_device(self._h, vp(d_unary), vp(d_q), vp(d_qprim),
                                                     vp(d_positions), vp(d_alphas),
                                                     C.c_double(float(tol)), err,
                                                     C.c_size_t(len(err)))
        _lib.check(rc, err)

    def reset(self):
        err = _lib.errbuf()
        _lib.check(_lib.lib().stereo_trws_plan_reset(self._h, err, C.c_size_t(len(err))), err)

    def iterate(self, iters, max_relgap=0.0, stream=None):
        done, stopped = C.c_int(), C.c_int()
        err = _lib.errbuf()
        rc = _lib.lib().stereo_trws_plan_iterate(self._h, C.c_int(int(iters)),
                                                 C.c_double(float(max_relgap)),
                                                 C.c_void_p(int(stream)) if stream else None,
                                                 C.byref(done), C.byref(stopped), err,
                                                 C.c_size_t(len(err)))
        _lib.check(rc, err)
        return done.value, bool(stopped.value)

    def result(self, want_labels=True):
        lab = np.zeros(self.N) if want_labels else None
        en, lb, it = C.c_double(), C.c_double(), C.c_double()
        err = _lib.errbuf()
        rc = _lib.lib().stereo_trws_plan_result(self._h, _ptr(lab) if want_labels else None,
                                                C.byref(en), C.byref(lb), C.byref(it), err,
                                                C.c_size_t(len(err)))
        _lib.check(rc, err)
        return lab, en.value, lb.value, it.value

    def info(self):
        rank = np.zeros(self.N, np.int64)
        lv, mx = C.c_int64(), C.c_int64()
        err = _lib.errbuf()
        rc = _lib.lib().stereo_trws_plan_info(self._h, _ptr(rank, C.c_int64), C.byref(lv),
                                              C.byref(mx), err, C.c_size_t(len(err)))
        _lib.check(rc, err)
        return dict(rank=rank, levels=lv.value, max_level_nodes=mx.value)

    def path(self):
        """1 generic persistent, 2 pipelined (K <= 64), 3 wide pipelined, 4 pipelined with two
        labels per lane (64 < K <= 128)."""
        return int(_lib.lib().stereo_trws_plan_path(self._h))

    def serial_messages(self, reset=False):
        n = C.c_int64()
        _lib.lib().stereo_trws_plan_counters(self._h, C.byref(n), C.c_int(int(reset)))
        return n.value

    def spec_stats(self):
        """Speculative schedule of the border chain (stereo_trws_plan_spec_stats): dict(active, second_walks, commits,
        runner_visits)."""
        out = (C.c_int64 * 4)()
        _lib.lib().stereo_trws_plan_spec_stats(self._h, out)
        return dict(active=bool(out[0]), second_walks=int(out[1]), commits=int(out[2]), runner_visits=int(out[3]))

    def stats(self, reset=False):
        ms, n = C.c_double(), C.c_int64()
        _lib.lib().stereo_trws_plan_stats(self._h, C.byref(ms), C.byref(n), C.c_int(int(reset)))
        return ms.value, n.value


def analyze(N, connectivity0):
    """Host-only graph analysis (stereo_trws_analyze); connectivity zero based."""
    c = np.asarray(connectivity0)
    if c.ndim != 2 or c.shape[0] != 2:
        raise StereoHipError("connectivity must be 2 x E")
    c = np.asfortranarray(c, dtype=np.uint32)
    E = c.shape[1]
    i64 = lambda n: np.zeros(n, np.int64)
    out = dict(rank=i64(N), tail=i64(E), head=i64(E), dir=np.zeros(E, np.int32),
               fwd_ptr=i64(N + 1), fwd_idx=i64(E), bwd_ptr=i64(N + 1), bwd_idx=i64(E),
               level=i64(N))
    err = _lib.errbuf()
    P = lambda a, t=C.c_int64: _ptr(a, t)
    rc = _lib.lib().stereo_trws_analyze(C.c_int64(N), C.c_int64(E), _ptr(c, C.c_uint32),
                                        P(out["rank"]), P(out["tail"]), P(out["head"]),
                                        P(out["dir"], C.c_int32), P(out["fwd_ptr"]),
                                        P(out["fwd_idx"]), P(out["bwd_ptr"]), P(out["bwd_idx"]),
                                        P(out["level"]), err, C.c_size_t(len(err)))
    _lib.check(rc, err)
    return out


def schedule(N, connectivity0, direction, max_resident_runs=0):
    """Host-only inspection of the chain schedule of the descriptor-driven sweep kernels
    (stereo_trws_schedule); connectivity zero based.  Returns dict(rank_at, run_ptr,
    ticket_run, pred_rank, dep_ptr, dep_rank)."""
    c = np.asarray(connectivity0)
    if c.ndim != 2 or c.shape[0] != 2:
        raise StereoHipError("connectivity must be 2 x E")
    c = np.asfortranarray(c, dtype=np.uint32)
    E = c.shape[1]
    i64 = lambda n: np.zeros(n, np.int64)
    rank_at, run_ptr, ticket_run, pred, dep_ptr, dep_rank = i64(N), i64(N + 1), i64(N), i64(N), i64(N + 1), i64(4 * N)
    nruns = C.c_int64(0)
    err = _lib.errbuf()
    P = lambda a: _ptr(a, C.c_int64)
    rc = _lib.lib().stereo_trws_schedule(C.c_int64(N), C.c_int64(E), _ptr(c, C.c_uint32),
                                         C.c_int64(max_resident_runs), C.c_int(direction), P(rank_at),
                                         P(run_ptr), C.byref(nruns), P(ticket_run), P(pred), P(dep_ptr),
                                         P(dep_rank), err, C.c_size_t(len(err)))
    _lib.check(rc, err)
    R = nruns.value
    return dict(rank_at=rank_at, run_ptr=run_ptr[:R + 1], ticket_run=ticket_run[:R], pred_rank=pred,
                dep_ptr=dep_ptr, dep_rank=dep_rank[:dep_ptr[N]])


def simulate_schedule(sched, workgroups, visit=1.0, handover=0.0):
    """Discrete simulation of the dataflow sweep on `workgroups` resident workgroups that take
    run tickets in order: a visit costs `visit`, a message from another run arrives `handover`
    after its node finished.  Returns (makespan, finished_all).  Deadlock -> finished_all False."""
    import heapq
    rank_at, run_ptr, ticket_run = sched["rank_at"], sched["run_ptr"], sched["ticket_run"]
    dep_ptr, dep_rank = sched["dep_ptr"], sched["dep_rank"]
    N = len(rank_at)
    R = len(ticket_run)
    done_t = np.full(N, -1.0)          # completion time by rank
    waiting = {}                       # rank -> list of runs blocked on it
    cur = [int(run_ptr[k]) for k in range(R)]   # next schedule position of each run
    run_time = [0.0] * R               # local clock of each held run
    next_ticket = 0
    free_at = []                       # min-heap of times at which a workgroup becomes free
    events = []                        # (time, run) runs ready to try advancing
    for _ in range(min(int(workgroups), R)):
        heapq.heappush(free_at, 0.0)
    finished = 0

    def start_next():
        nonlocal next_ticket
        while free_at and next_ticket < R:
            t0 = heapq.heappop(free_at)
            k = int(ticket_run[next_ticket]); next_ticket += 1
            run_time[k] = t0
            heapq.heappush(events, (t0, k))

    start_next()
    while events:
        t, k = heapq.heappop(events)
        blocked = False
        while cur[k] < run_ptr[k + 1]:
            r = int(rank_at[cur[k]])
            ready = run_time[k]
            for x in dep_rank[dep_ptr[r]:dep_ptr[r + 1]]:
                x = int(x)
                if done_t[x] < 0:
                    waiting.setdefault(x, []).append(k)
                    blocked = True
                    break
                ready = max(ready, done_t[x] + handover)
            if blocked:
                break
            run_time[k] = ready + visit
            done_t[r] = run_time[k]
            for kk in waiting.pop(r, []):
                heapq.heappush(events, (run_time[k], kk))
            cur[k] += 1
        if not blocked:
            finished += 1
            heapq.heappush(free_at, run_time[k])
            start_next()
    return float(done_t.max()), finished == R


def spec_schedule(N, connectivity0, direction):
    """Host-only inspection of the speculative schedule (stereo_trws_spec_schedule); connectivity zero based.
    Returns None if the graph has no run to cut, else dict(run, c0, c1, seg_len, nseg, run_ptr, kind, ticket_run)."""
    c = np.asarray(connectivity0)
    if c.ndim != 2 or c.shape[0] != 2:
        raise StereoHipError("connectivity must be 2 x E")
    c = np.asfortranarray(c, dtype=np.uint32)
    E = c.shape[1]
    i64 = lambda n: np.zeros(n, np.int64)
    info, run_ptr, kind, ticket_run = i64(6), i64(N + 2), i64(N + 1), i64(N + 2)
    nruns = C.c_int64(0)
    err = _lib.errbuf()
    P = lambda a: _ptr(a, C.c_int64)
    rc = _lib.lib().stereo_trws_spec_schedule(C.c_int64(N), C.c_int64(E), _ptr(c, C.c_uint32), C.c_int(direction), P(info),
                                              C.byref(nruns), P(run_ptr), P(kind), P(ticket_run), err, C.c_size_t(len(err)))
    _lib.check(rc, err)
    if not info[0]:
        return None
    R = nruns.value
    return dict(run=int(info[1]), c0=int(info[2]), c1=int(info[3]), seg_len=int(info[4]), nseg=int(info[5]),
                run_ptr=run_ptr[:R + 1], kind=kind[:R], ticket_run=ticket_run[:R + 1])


def simulate_spec_schedule(sched, spec, workgroups, visit=1.0, runner_visit=0.25, handover=0.0):
    """Discrete simulation of a sweep on the speculative schedule with `workgroups` resident workgroups
    that take tickets in order.  sched: schedule() of the same graph and direction (positions, dependencies);
    spec: spec_schedule().  The runner (ticket -1) walks the cut run at `runner_visit` per node -- a node's foreign
    dependencies must be done (a node of the cut run is done when its SEGMENT has committed) -- and publishes at every
    cut; segment s starts behind the runner's cut s, walks its nodes at `visit`, and commits once segment s - 1 has
    (+ `handover`): only then are its nodes done for everybody else.  Returns (makespan, finished_all, commit times)."""
    import heapq
    rank_at, dep_ptr, dep_rank = sched["rank_at"], sched["dep_ptr"], sched["dep_rank"]
    run_ptr, kind, ticket_run = spec["run_ptr"], spec["kind"], spec["ticket_run"]
    c0, c1, L, nseg = spec["c0"], spec["c1"], spec["seg_len"], spec["nseg"]
    N = len(rank_at)
    R = len(kind)
    done_t = np.full(N, -1.0)
    pub_t = np.full(nseg + 1, -1.0); pub_t[0] = 0.0
    commit_t = np.full(nseg, -1.0)
    waiting = {}                                   # event key -> tasks blocked on it
    cur = {k: int(run_ptr[k]) for k in range(R)}   # next schedule position of each run
    cur[-1] = c0
    clock = {}
    walked = {}                                    # segments: end of the walk
    next_ticket = 0
    free_at, events = [], []
    T = len(ticket_run)
    for _ in range(min(int(workgroups), T)):
        heapq.heappush(free_at, 0.0)
    finished = 0

    def start_next():
        nonlocal next_ticket
        while free_at and next_ticket < T:
            t0 = heapq.heappop(free_at)
            k = int(ticket_run[next_ticket]); next_ticket += 1
            clock[k] = t0
            heapq.heappush(events, (t0, k))

    def wait(key, k):
        waiting.setdefault(key, []).append(k)

    def fire(key, t):
        for kk in waiting.pop(key, []):
            heapq.heappush(events, (t, kk))

    def deps_ready(r, k, now):
        for x in dep_rank[dep_ptr[r]:dep_ptr[r + 1]]:
            x = int(x)
            if done_t[x] < 0:
                wait(("rank", x), k)
                return None
            now = max(now, done_t[x] + handover)
        return now

    start_next()
    while events:
        t, k = heapq.heappop(events)
        clock[k] = max(clock[k], t)
        blocked = False
        if k == -1:                                # the runner
            while cur[-1] < c1:
                ready = deps_ready(int(rank_at[cur[-1]]), k, clock[k])
                if ready is None:
                    blocked = True; break
                clock[k] = ready + runner_visit
                cur[-1] += 1
                off = cur[-1] - c0
                if cur[-1] < c1 and off % L == 0 and off // L < nseg:
                    pub_t[off // L] = clock[k]
                    fire(("pub", off // L), clock[k])
        else:
            seg = int(kind[k]) - 1
            if seg > 0 and cur[k] == int(run_ptr[k]):
                if pub_t[seg] < 0:
                    wait(("pub", seg), k); continue
                clock[k] = max(clock[k], pub_t[seg] + handover)
            while cur[k] < run_ptr[k + 1]:
                r = int(rank_at[cur[k]])
                ready = deps_ready(r, k, clock[k])
                if ready is None:
                    blocked = True; break
                clock[k] = ready + visit
                if seg < 0:
                    done_t[r] = clock[k]
                    fire(("rank", r), clock[k])
                cur[k] += 1
            if not blocked and seg >= 0:
                if seg > 0 and commit_t[seg - 1] < 0:
                    wait(("commit", seg - 1), k); continue
                if seg > 0:
                    clock[k] = max(clock[k], commit_t[seg - 1] + handover)
                commit_t[seg] = clock[k]
                for pos in range(int(run_ptr[k]), int(run_ptr[k + 1])):
                    done_t[int(rank_at[pos])] = clock[k]
                    fire(("rank", int(rank_at[pos])), clock[k])
                fire(("commit", seg), clock[k])
        if not blocked:
            finished += 1
            heapq.heappush(free_at, clock[k])
            start_next()
    return float(done_t.max()), finished == T and bool((done_t >= 0).all()), commit_t


def look_ahead_allowed(sched, direction):
    """Bit 12 of descriptor word 2 (trws_graph.cpp), restated from the schedule: by rank, may a loader
    wait for the node's foreign dependencies two visits ahead?  True if every dependency comes before
    the node visited two steps earlier in the same run (one step, or the node itself, at the start of
    a run) in this sweep's order."""
    rank_at, run_ptr = sched["rank_at"], sched["run_ptr"]
    dep_ptr, dep_rank = sched["dep_ptr"], sched["dep_rank"]
    N = len(rank_at)
    before = (lambda x, b: x < b) if direction == 0 else (lambda x, b: x > b)
    ok = np.ones(N, bool)
    for k in range(len(run_ptr) - 1):
        a, b = int(run_ptr[k]), int(run_ptr[k + 1])
        for pos in range(a, b):
            r = int(rank_at[pos])
            bound = int(rank_at[pos - 2]) if pos - 2 >= a else int(rank_at[pos - 1]) if pos - 1 >= a else r
            ok[r] = all(before(int(x), bound) for x in dep_rank[dep_ptr[r]:dep_ptr[r + 1]])
    return ok


def simulate_look_ahead(sched, allowed):
    """Does the sweep kernels' loader protocol terminate when the loader of the data behind flags works
    two visits ahead on the nodes `allowed` (by rank) names?  Every run has its own workgroup.  The
    visit that computes position i of a run can end only when (a) the dependencies of position i + 1
    are visible (its data is staged during visit i at the latest) and (b) those of position i + 2 are,
    if that node is allowed (the loader waits for them during visit i); a node becomes visible to other
    runs when the visit that computed it has ended (the storer works during the next one, whatever
    the loader waits for).  Returns True if every run finishes, False on a deadlock."""
    rank_at, run_ptr = sched["rank_at"], sched["run_ptr"]
    dep_ptr, dep_rank = sched["dep_ptr"], sched["dep_rank"]
    N = len(rank_at)
    R = len(run_ptr) - 1
    visible = np.zeros(N, bool)

    def deps_visible(pos):
        r = int(rank_at[pos])
        return all(visible[int(x)] for x in dep_rank[dep_ptr[r]:dep_ptr[r + 1]])

    ended = [0] * R   # visits ended per run: 1 = the lead-in visit, 1 + j = the visit computing the run's j-th node
    progress = True
    while progress:
        progress = False
        for k in range(R):
            a, b = int(run_ptr[k]), int(run_ptr[k + 1])
            while ended[k] < b - a + 1:
                i = a + ended[k] - 1           # position computed by the visit about to end (a - 1: lead-in, nothing computed)
                if i + 1 < b and not deps_visible(i + 1):
                    break                       # node i + 1 cannot be staged yet
                if i + 2 < b and allowed[int(rank_at[i + 2])] and not deps_visible(i + 2):
                    break                       # the loader waits two visits ahead
                ended[k] += 1
                if i >= a:
                    visible[int(rank_at[i])] = True
                progress = True
    return all(ended[k] == int(run_ptr[k + 1]) - int(run_ptr[k]) + 1 for k in range(R))


def messages(kernel, Di, gamma, msg_in, q_source, q_dest, alpha, lam, certificate=True, window=-1,
             shared_positions=None):
    """M message updates on the device (stereo_trws_messages): arrays are M x K (row = message),
    gamma / alpha length M.  Returns (msg_out M x K, vmin M, used_serial M)."""
    c = lambda a: np.ascontiguousarray(a, dtype=np.float64)
    Di, msg_in, q_source, q_dest = c(Di), c(msg_in), c(q_source), c(q_dest)
    M, K = Di.shape
    gamma, alpha = c(np.broadcast_to(gamma, (M,))), c(np.broadcast_to(alpha, (M,)))
    out, vmin, ser = np.zeros((M, K)), np.zeros(M), np.zeros(M, np.int32)
    sp = c(shared_positions) if shared_positions is not None else None
    err = _lib.errbuf()
    rc = _lib.lib().stereo_trws_messages(C.c_int(int(kernel)), C.c_int(K), C.c_int64(M), _ptr(Di), _ptr(gamma),
                                         _ptr(msg_in), _ptr(q_source), _ptr(q_dest), _ptr(alpha), C.c_double(float(lam)),
                                         C.c_int(int(bool(certificate))), C.c_int(int(window)),
                                         _ptr(sp) if sp is not None else None, _ptr(out), _ptr(vmin),
                                         _ptr(ser, C.c_int32), err, C.c_size_t(len(err)))
    _lib.check(rc, err)
    return out, vmin, ser
