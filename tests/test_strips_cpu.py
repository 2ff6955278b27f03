"""CPU tests of the row-strip decomposition (no GPU): the strip-aware chain schedule out of
trws_graph.cpp, and the hand-off protocol between strips driven by two gloo ranks."""
import os
import subprocess
import sys
import textwrap

import numpy as np
import pytest

from helpers import grid_conn

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _incoming(a, direction):
    """ranks of the incoming neighbours of every rank in one sweep direction, from stereo_trws_analyze."""
    N = len(a["rank"])
    rank = a["rank"]
    inc = [[] for _ in range(N)]
    for e in range(len(a["tail"])):
        t, h = int(rank[a["tail"][e]]), int(rank[a["head"][e]])   # oriented: tail has the lower rank
        if direction == 0:
            inc[h].append(t)
        else:
            inc[t].append(h)
    return inc


@pytest.mark.parametrize("shape,G", [((20, 24), 2), ((20, 24), 4), ((9, 30), 3), ((12, 5), 6), ((40, 7), 5)])
def test_strip_schedule_structure(shape, G):
    from stereo_amd.strips import schedule_strips, row_strip_owner
    from stereo_amd.trws import analyze, schedule, simulate_schedule
    H, W = shape
    N = H * W
    conn = grid_conn(H, W)
    own = row_strip_owner(H, W, G)
    a = analyze(N, conn.T)
    own_by_rank = np.zeros(N, dtype=np.int64)
    own_by_rank[a["rank"]] = own
    for d in (0, 1):
        s = schedule_strips(N, conn.T, d, own, G)
        base = schedule(N, conn.T, d)
        R = len(s["ticket_run"])
        # every node exactly once, every run inside one strip
        assert sorted(s["rank_at"].tolist()) == list(range(N))
        for k in range(R):
            ranks = s["rank_at"][s["run_ptr"][k]:s["run_ptr"][k + 1]]
            assert np.all(own_by_rank[ranks] == s["run_strip"][k])
        # runs are only ever cut (the border chain where it crosses a strip boundary), never merged
        assert R >= len(base["ticket_run"])
        # notify bits: exactly the nodes with an outgoing edge into another strip, towards that strip
        inc = _incoming(a, d)
        need = {}
        for r in range(N):
            for x in inc[r]:
                if own_by_rank[x] != own_by_rank[r]:
                    need.setdefault(x, set()).add(int(own_by_rank[r]))
        for r in range(N):
            rem = int(s["remote"][r])
            got = set()
            if rem & (1 << 16):
                got.add(int(own_by_rank[r]) - 1)
            if rem & (1 << 17):
                got.add(int(own_by_rank[r]) + 1)
            assert got == need.get(r, set()), (d, r)
            # a cross-strip neighbour is never the LDS predecessor: it is a flag dependency
            for x in inc[r]:
                if own_by_rank[x] != own_by_rank[r]:
                    assert x in s["dep_rank"][s["dep_ptr"][r]:s["dep_ptr"][r + 1]]
        # with every run resident the sweep still takes exactly the DAG depth
        depth0, ok0 = simulate_schedule(base, 10 ** 6)
        depth, ok = simulate_schedule(s, 10 ** 6)
        assert ok and ok0 and depth == depth0


@pytest.mark.parametrize("shape,G", [((20, 24), 2), ((9, 30), 3), ((12, 5), 4)])
def test_descriptor_look_ahead_bit_is_the_rule_and_cannot_deadlock(shape, G):
    """Bit 12 of descriptor word 2 as the library writes it (read back through the strips' host-side
    layout) equals the rule restated in stereo_amd.trws.look_ahead_allowed, and the loader protocol
    with it terminates on the strip-aware schedule (halo dependencies included)."""
    from stereo_amd.strips import schedule_strips, strip_layout_host, row_strip_owner
    from stereo_amd.trws import analyze, look_ahead_allowed, simulate_look_ahead
    H, W = shape
    N = H * W
    conn = grid_conn(H, W)
    own = row_strip_owner(H, W, G)
    rank = analyze(N, conn.T)["rank"]
    for d in (0, 1):
        s = schedule_strips(N, conn.T, d, own, G)
        rule = look_ahead_allowed(s, d)
        assert simulate_look_ahead(s, rule)
        seen = 0
        for g in range(G):
            lay = strip_layout_host(N, conn.T, own, G, g, d)
            for v in range(lay["desc"].shape[0]):
                node = int(lay["nodes"][lay["desc"][v, 0]])
                bit = (int(lay["desc"][v, 2]) >> 12) & 1
                assert bit == int(rule[rank[node]]), (d, g, v, node)
                seen += 1
        assert seen == N


def test_strips_must_be_a_chain():
    from stereo_amd import StereoHipError
    from stereo_amd.strips import schedule_strips
    H, W = 6, 5
    conn = grid_conn(H, W)
    own = np.tile(np.array([0, 0, 2, 2, 1, 1], dtype=np.int32), W)   # strip 0 touches strip 2
    with pytest.raises(StereoHipError, match="chain"):
        schedule_strips(H * W, conn.T, 0, own, 3)


def test_single_process_protocol_model():
    """All strips of a sweep in one process, hand-offs through queues: every visit finds what it
    reads, values equal the serial dataflow, also with two workgroups per strip."""
    from collections import deque
    from stereo_amd.strips import schedule_strips, row_strip_owner, simulate_strip, dataflow_reference
    from stereo_amd.trws import analyze
    H, W, G = 14, 11, 3
    N = H * W
    conn = grid_conn(H, W)
    own = row_strip_owner(H, W, G)
    a = analyze(N, conn.T)
    for d in (0, 1):
        s = schedule_strips(N, conn.T, d, own, G)
        inc = _incoming(a, d)
        ref = dataflow_reference(s, inc)
        # cooperative round robin: a strip that has to wait yields to the others
        import threading, queue
        boxes = [queue.Queue() for _ in range(G)]
        out = [None] * G
        errs = []

        def work(g):
            try:
                out[g] = simulate_strip(s, g, inc, lambda to, r, v: boxes[to].put((r, v)),
                                        lambda: boxes[g].get(timeout=30), workgroups=2)
            except Exception as exc:  # noqa: BLE001
                errs.append((g, exc))
        th = [threading.Thread(target=work, args=(g,)) for g in range(G)]
        [t.start() for t in th]
        [t.join(60) for t in th]
        assert not errs, errs
        got = {}
        for g in range(G):
            got.update(out[g])
        assert len(got) == N and all(got[r] == ref[r] for r in range(N))


def _worker_source():
    return textwrap.dedent("""
        import json, os, sys
        sys.path.insert(0, {root!r}); sys.path.insert(0, os.path.join({root!r}, "tests"))
        import numpy as np, torch
        import torch.distributed as dist
        from helpers import grid_conn
        from stereo_amd.strips import schedule_strips, row_strip_owner, simulate_strip, dataflow_reference
        from stereo_amd.trws import analyze
        from test_strips_cpu import _incoming
        dist.init_process_group("gloo")
        rank, world = dist.get_rank(), dist.get_world_size()
        H, W = 22, 17
        N = H * W
        conn = grid_conn(H, W)
        own = row_strip_owner(H, W, world)
        a = analyze(N, conn.T)
        ok = True
        sent = [0]
        for sweep in range(4):                 # forward, backward, forward, backward
            d = sweep % 2
            s = schedule_strips(N, conn.T, d, own, world)
            inc = _incoming(a, d)
            ref = dataflow_reference(s, inc)
            pending = []
            def send(to, r, v):
                t = torch.tensor([int(r), int(v)], dtype=torch.int64)
                pending.append((dist.isend(t, dst=to), t)); sent[0] += 1
            def recv():
                t = torch.zeros(2, dtype=torch.int64)
                dist.recv(t)                   # from whichever neighbour is ready first
                return int(t[0]), np.uint64(int(t[1]))
            got = simulate_strip(s, rank, inc, send, recv, workgroups=3)
            for w, _ in pending:
                w.wait()
            ok = ok and all(got[r] == ref[r] for r in got) and len(got) == int((own == rank).sum())
            dist.barrier()
        res = [None] * world
        dist.all_gather_object(res, (ok, sent[0]))
        if rank == 0:
            print(json.dumps({{"ok": [bool(r[0]) for r in res], "sent": [r[1] for r in res]}}))
        dist.destroy_process_group()
    """).format(root=ROOT)


@pytest.mark.parametrize("world", [2, 3])
def test_gloo_ranks_hand_over_strip_boundaries(world, tmp_path):
    """One strip per gloo rank: boundary values travel rank to rank exactly where the descriptors
    say (notify bits), every visit finds its inputs, nobody deadlocks, and the values equal the
    serial sweep -- the multi-process shape of bench.py --gpus N without a GPU."""
    import json
    script = tmp_path / "worker.py"
    script.write_text(_worker_source())
    port = 29600 + (os.getpid() % 300) + world
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=%d" % world,
           "--master-addr", "127.0.0.1", "--master-port", str(port), str(script)]
    env = dict(os.environ, OMP_NUM_THREADS="1")
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=600, env=env)
    assert out.returncode == 0, out.stdout[-1500:] + out.stderr[-3000:]
    r = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1])
    assert all(r["ok"]) and len(r["ok"]) == world
    # per sweep a boundary row hands 17 columns (+ the two border-chain crossings) each way
    assert all(n > 0 for n in r["sent"])


@pytest.mark.parametrize("shape,G", [((20, 24), 2), ((21, 17), 4), ((9, 30), 3), ((12, 5), 6)])
def test_strip_local_ids_agree_between_neighbours(shape, G):
    """Strip-local storage (what lets 8 GPUs hold a volume none of them could hold alone): every
    strip numbers its nodes, halo and edges itself, and writes boundary messages / flags / labels
    into its neighbours' arrays at THEIR ids (descriptor words 45-54).  Checked on the host: the
    strips partition the visits, every id is inside the strip's arrays, and a peer id names, in the
    neighbour's own numbering, the very edge / node the writer means."""
    from stereo_amd.strips import strip_layout_host, row_strip_owner
    H, W = shape
    N = H * W
    conn = grid_conn(H, W)
    E = conn.shape[0]
    own = row_strip_owner(H, W, G)
    for direction in (0, 1):
        L = [strip_layout_host(N, conn.T, own, G, s, direction) for s in range(G)]
        visited = np.zeros(N, dtype=np.int64)
        for s in range(G):
            nodes, n_own, edges, desc = L[s]["nodes"], L[s]["n_own"], L[s]["edges"], L[s]["desc"]
            assert np.all(own[nodes[:n_own]] == s) and np.all(own[nodes[n_own:]] != s)
            assert len(np.unique(nodes)) == len(nodes) and len(np.unique(edges)) == len(edges)
            touching = (own[conn[:, 0]] == s) | (own[conn[:, 1]] == s)
            assert np.array_equal(edges, np.nonzero(touching)[0])
            assert desc.shape[0] == n_own                      # one visit per own node and sweep
            for D in desc:
                v = D[0]
                assert 0 <= v < n_own and D[1] == v
                gv = nodes[v]
                visited[gv] += 1
                nout, nin, nd = D[2] & 15, (D[2] >> 4) & 15, (D[2] >> 8) & 15
                rem = int(D[43]) & 0xFFFFFFFF
                for k in range(nout + nin):
                    assert 0 <= D[4 + k] < len(edges)
                    ge = edges[D[4 + k]]
                    assert gv in conn[ge]                      # the edge is at this node
                    other = conn[ge, 0] + conn[ge, 1] - gv
                    if k >= nout:                              # incoming: the neighbour whose label the primal reads
                        assert nodes[D[32 + k]] == other
                    elif (rem >> k) & 1:                       # outgoing into a neighbour strip
                        peer = s + (1 if (rem >> (8 + k)) & 1 else -1)
                        assert own[other] == peer
                        assert L[peer]["edges"][D[45 + k]] == ge          # the neighbour's id of the same edge
                    else:
                        assert own[other] == s
                for k in range(nd):
                    assert 0 <= D[20 + k] < len(nodes)         # a flag this strip holds (own node or halo)
                if rem & (1 << 16):
                    assert s > 0 and L[s - 1]["nodes"][D[53]] == gv     # flag / label also raised at strip - 1 ...
                if rem & (1 << 17):
                    assert s + 1 < G and L[s + 1]["nodes"][D[54]] == gv  # ... / strip + 1, at their id of this node
                # a node next to another strip's node is announced there
                nbr_strips = {int(own[conn[e, 0] + conn[e, 1] - gv]) for e in np.nonzero((conn[:, 0] == gv) | (conn[:, 1] == gv))[0]}
                out_strips = {s + (1 if (rem >> (8 + k)) & 1 else -1) for k in range(nout) if (rem >> k) & 1}
                assert out_strips <= nbr_strips - {s}
                assert bool(rem & (1 << 16)) == (s - 1 in out_strips) and bool(rem & (1 << 17)) == (s + 1 in out_strips)
        assert np.all(visited == 1)
