"""Development tool (GPU): configs[1] to its stop (617 iterations), the speculative schedule of the border chain
against the plain schedule, iteration by iteration (energy, bound, labels), and both against tests/golden/full_runs.json.
On the first difference: where the lower-bound terms and the message rows differ.
usage: spec_full_check.py [max iterations=700] [fetch messages from iteration=360]"""
import ctypes as C, json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
import numpy as np
from make_golden_full import config1_inputs
from stereo_amd.trws import TrwsPlan
from stereo_amd import _lib
maxit = int(sys.argv[1]) if len(sys.argv) > 1 else 700
mfrom = int(sys.argv[2]) if len(sys.argv) > 2 else 360
want = json.load(open(os.path.join(ROOT, "tests", "golden", "full_runs.json")))["config1"]
unary, conn, K, tol = config1_inputs()
H, W = 375, 450
N, E = unary.shape[0], conn.shape[0]

def make(env):
    for k in ("STEREO_HIP_TRWS_SPEC", "STEREO_HIP_TRWS_DEBUG"):
        os.environ.pop(k, None)
    os.environ.update(env)
    plan = TrwsPlan(1, K, N, conn.T)
    plan.upload(unary.T, np.ones(E), tol, positions=np.arange(K, dtype=np.float64))
    return plan

def terms(plan):
    n = C.c_int64()
    _lib.lib().stereo_trws_plan_debug_terms(plan._h, None, C.c_int64(0), C.byref(n))
    out = np.zeros(n.value)
    _lib.lib().stereo_trws_plan_debug_terms(plan._h, out.ctypes.data_as(C.POINTER(C.c_double)), C.c_int64(n.value), C.byref(n))
    return out

def messages(plan):
    out = np.zeros((E, K))
    rc = _lib.lib().stereo_trws_plan_debug_messages(plan._h, out.ctypes.data_as(C.POINTER(C.c_double)), C.c_int64(E * K))
    assert rc == 0
    return out

a = make({"STEREO_HIP_TRWS_SPEC": "0"})
b = make({})
rank = a.info()["rank"]
order = np.argsort(rank)
# terms in summation order: rank N - 1 down to 0, 1 + (edges to lower-ranked neighbours) each
nb = np.zeros(N, np.int64)
lo = rank[conn[:, 0]] < rank[conn[:, 1]]
np.add.at(nb, np.where(lo, conn[:, 1], conn[:, 0]), 1)
cnt = 1 + nb[order[::-1]]
start = np.concatenate([[0], np.cumsum(cnt)])
pm = None
first = None
for it in range(1, maxit + 1):
    a.iterate(1, max_relgap=-1e300); b.iterate(1, max_relgap=-1e300)
    la, ea, ba, _ = a.result(want_labels=True); lb_, eb, bb, _ = b.result(want_labels=True)
    same = ea == eb and ba == bb and np.array_equal(la, lb_)
    if not same and first is None:
        first = it
        print("iteration %d: energy %r vs %r, bound %r vs %r, labels differ at %d nodes" % (it, ea, eb, ba, bb, int((la != lb_).sum())), flush=True)
        print("   spec stats", b.spec_stats())
        ta, tb = terms(a), terms(b)
        d = np.flatnonzero(ta != tb)
        print("   lower-bound terms: %d of %d differ; first index %s" % (d.size, ta.size, d[:8]))
        for i in d[:8]:
            k = int(np.searchsorted(start, i, side="right") - 1)
            node = int(order[::-1][k])
            print("      term %d: backward position %d (rank %d), node %d = (row %d, col %d), term %d of the node: %r vs %r" % (
                i, k, N - 1 - k, node, node % H, node // H, i - start[k], ta[i], tb[i]))
        if pm is not None:
            print("   messages BEFORE this iteration's backward sweep (behind the fused forward sweep): %d rows differ" % int((pm[0] != pm[1]).any(axis=1).sum()))
            dr = np.flatnonzero((pm[0] != pm[1]).any(axis=1))
            for e in dr[:10]:
                t, h = conn[e]
                print("      edge %d: %d (row %d col %d rank %d) -> %d (row %d col %d rank %d): max |diff| %g" % (
                    e, t, t % H, t // H, rank[t], h, h % H, h // H, rank[h], np.abs(pm[0][e] - pm[1][e]).max()))
            if dr.size:
                rk = np.minimum(rank[conn[dr, 0]], rank[conn[dr, 1]])
                j = dr[np.argmin(rk)]
                print("      lowest-ranked node with a differing row: rank %d" % rk.min(), conn[j], [(int(n % H), int(n // H)) for n in conn[j]])
        ma, mb = messages(a), messages(b)
        dr = np.flatnonzero((ma != mb).any(axis=1))
        print("   messages after the iteration: %d rows differ, max |diff| %g" % (dr.size, np.abs(ma - mb).max()))
    if it >= mfrom and first is None:
        pm = (messages(a), messages(b))
    if first is not None and it > first + 1: break
    if it == want["iterations"]:
        print("iteration %d: plain energy %r bound %r | golden %r %r | spec %r %r" % (it, ea, ba, want["energy"], want["lower_bound"], eb, bb))
print("first difference at iteration", first, "| spec stats", b.spec_stats())
