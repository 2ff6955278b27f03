"""Development tool: randomised parity stress of the QPBO path against the reference's own QPBO
library (oracle/_ref/libref_qpbo.so, where it travelled) or the restated oracle."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import stereo_amd
from stereo_amd.rd import RdPlan
from helpers import fusion_problem, glass_problem
from oracle import pyoracle as po

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
use_ref = po.have_ref_qpbo()
t0, n, bad, unl = time.time(), 0, 0, 0
while time.time() - t0 < budget:
    H, W = int(rng.integers(2, 90)), int(rng.integers(2, 90))
    seed = int(rng.integers(0, 1 << 30))
    kind = int(rng.integers(0, 4))
    if kind == 0:
        p = fusion_problem(seed, H, W)
    elif kind == 1:
        p = fusion_problem(seed, H, W, nonsub_boost=float(rng.uniform(1, 8)))
    elif kind == 2:
        p = glass_problem(seed, H, W, float(rng.uniform(0.5, 4)), bool(rng.integers(0, 2)))
    else:
        p = fusion_problem(seed, H, W, kernel=2, tol=20.0, integer=True)
    args = (p["U0"], p["U1"], p["E00"], p["E01"], p["E10"], p["E11"])
    ref = po.ref_rd(*args, p["conn"]) if use_ref else po.rd(*args, p["conn"])
    plan = RdPlan(H * W, p["conn"].T, grid=(H, W) if rng.integers(0, 2) else None)
    got = plan.solve(*args)
    # Strong persistency is unique and must match bit for bit.  Where the reference resolved nodes by
    # weak persistency between incomparable components its answer depends on its DFS order: there the
    # contract is "same energy" (INTEGRATION.md).
    strong = po.ref_rd(*args, p["conn"], stage=1)[0] if use_ref else ref[0]
    mask = strong >= 0
    ok = (np.array_equal(got[0][mask], ref[0][mask]) and abs(got[2] - ref[2]) <= 1e-9 * max(1.0, abs(ref[2]))
          and got[1] <= ref[1] + 1e-9 * max(1.0, abs(ref[1])))
    n += 1; unl += int((~mask).any())
    if not ok:
        bad += 1
        print("MISMATCH", dict(seed=seed, H=H, W=W, kind=kind, unl_ref=ref[3], unl=got[3]))
print("stress rd: %d problems (%d with unlabelled nodes), %d mismatches, reference library: %s, %.0f s" % (n, unl, bad, use_ref, time.time() - t0))
sys.exit(1 if bad else 0)
