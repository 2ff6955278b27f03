"""Development tool (GPU): what the exchange-flag reset at the start of a run is for.  A speculative segment walked a
second time (the runner's row made wrong by a lot: STEREO_HIP_TRWS_DEBUG 65536) must give the plain schedule's bits;
with the reset switched off (131072) the finishing wave of a twin pair may merge the first walk's partial minima.
Prints both verdicts."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from helpers import grid_conn
from stereo_amd.trws import TrwsPlan

def solve(env, unary, conn, tol, iters):
    for k in ("STEREO_HIP_TRWS_SPEC", "STEREO_HIP_TRWS_DEBUG"):
        os.environ.pop(k, None)
    os.environ.update(env)
    N, K = unary.shape
    plan = TrwsPlan(1, K, N, conn.T)
    plan.upload(unary.T, np.ones(conn.shape[0]), tol, positions=np.arange(K, dtype=np.float64))
    out = []
    for _ in range(iters):
        plan.iterate(1, max_relgap=-1e300)
        lab, en, lb, _ = plan.result()
        out.append((lab.copy(), en, lb))
    st = plan.spec_stats()
    plan.close()
    return out, st

rng = np.random.default_rng(1)
H, W, K = 64, 64, 60
conn = grid_conn(H, W)
unary = rng.uniform(0, 40, size=(H * W, K))
plain, _ = solve({"STEREO_HIP_TRWS_SPEC": "0"}, unary, conn, 8.0, 8)
for name, dbg in (("second walks, flags reset", "65536"), ("second walks, flags left standing", str(65536 + 131072))):
    got, st = solve({"STEREO_HIP_TRWS_DEBUG": dbg}, unary, conn, 8.0, 8)
    same = all(np.array_equal(a[0], b[0]) and a[1] == b[1] and a[2] == b[2] for a, b in zip(plain, got))
    print("%s: %s  %s" % (name, "EQUAL" if same else "DIFFERENT", st))
    if not same:
        for i, (a, b) in enumerate(zip(plain, got)):
            print("   iter %d: labels differ at %d nodes, energy %r vs %r, bound %r vs %r" % (i + 1, int((a[0] != b[0]).sum()), a[1], b[1], a[2], b[2]))
