"""Development tool: randomised parity stress of the TRW-S kernels against the CPU oracle
(sizes, label counts, kernels, shared / per-edge positions, integer ties drawn at random)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import stereo_amd
from stereo_amd.trws import TrwsPlan
from helpers import trws_problem
from oracle import pyoracle as po

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
t0, n, bad = time.time(), 0, 0
modes = {}
while time.time() - t0 < budget:
    H, W = int(rng.integers(1, 70)), int(rng.integers(2, 70))
    K = int(rng.choice([2, 3, 5, 8, 16, 31, 60, 64, 65, 100, 200, 256]))
    if os.environ.get("STRESS_TRWS_K"):    # e.g. STRESS_TRWS_K=65,100,128: one kernel family only
        K = int(rng.choice([int(x) for x in os.environ["STRESS_TRWS_K"].split(",")]))
    if K > 64 and H * W > 1500:
        H, W = min(H, 30), min(W, 40)
    kernel = int(rng.choice([1, 1, 1, 2]))
    if os.environ.get("STRESS_TRWS_KERNEL"):
        kernel = int(os.environ["STRESS_TRWS_KERNEL"])
    shared = bool(rng.integers(0, 2))
    integer = bool(rng.integers(0, 4) == 0)
    tol = float(rng.choice([0.0, 1.5, 3.0, 8.0, 40.0]))
    iters = int(rng.integers(1, 6))
    seed = int(rng.integers(0, 1 << 30))
    p = trws_problem(seed, H, W, K, kind="fronto" if shared else "general", integer=integer)
    if rng.integers(0, 3) == 0:
        # out-of-range plane proposals give unaries of 4e7 (dispmap_ncc.m:245): some labels of some pixels
        huge = rng.random(p["unary"].shape) < 0.1
        p["unary"] = np.where(huge, 4e7 + p["unary"], p["unary"])
    if shared:
        pos = np.cumsum(rng.uniform(0.05, 2.0, size=K)) if rng.integers(0, 2) else np.arange(K, dtype=np.float64)
        q = np.tile(pos, (p["conn"].shape[0], 1)); qp = q
    else:
        q, qp = p["q"], p["qprim"]
    # message mode, node order and row strips drawn too (strips: labels bit for bit, scalars to 1e-12)
    minplus = bool(rng.integers(0, 4) == 0)
    ordering = int(rng.integers(0, 4) == 0)
    G = int(rng.choice([1, 1, 2, 3]))
    if G > 1 and (H < 2 * G or (minplus and not (shared and K > 64)) or (K > 64 and not shared) or (kernel == 2 and K > 64)):
        G = 1
    ref = po.trws(kernel, p["unary"], p["conn"], q, qp, p["alphas"], tol, iters, -1e300, mode=0 if minplus else 1,
                  ordering=ordering)
    mode = (1 if minplus else 0) | (0x100 if ordering else 0)
    if G == 1:
        plan = TrwsPlan(kernel, K, H * W, p["conn"].T, message_mode=mode)
    else:
        from stereo_amd.strips import make_strips
        plan = make_strips(kernel, K, H, W, p["conn"].T, G, message_mode=mode)
    if shared:
        plan.upload(p["unary"].T, p["alphas"], tol, positions=pos)
    else:
        plan.upload(p["unary"].T, p["alphas"], tol, q=q.T, qprim=qp.T)
    plan.iterate(iters, max_relgap=-1e300)
    got = plan.result()
    path = plan.path()
    plan.close()
    close = lambda a, b: abs(a - b) <= 1e-12 * max(abs(a), abs(b), 1.0)
    ok = np.array_equal(got[0], ref[0]) and ((got[1] == ref[1] and got[2] == ref[2]) if G == 1 else (close(got[1], ref[1]) and close(got[2], ref[2])))
    modes[(minplus, ordering, G, path)] = modes.get((minplus, ordering, G, path), 0) + 1
    n += 1
    if not ok:
        bad += 1
        print("MISMATCH", dict(seed=seed, H=H, W=W, K=K, kernel=kernel, shared=shared, integer=integer, tol=tol, iters=iters, path=path, minplus=minplus, ordering=ordering, strips=G))
print("stress: %d problems, %d mismatches, %.0f s" % (n, bad, time.time() - t0))
print("(minplus, index order, strips, kernel path) -> problems:", sorted(modes.items()))
sys.exit(1 if bad else 0)
