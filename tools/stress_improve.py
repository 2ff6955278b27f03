"""Development tool: randomised parity stress of QPBO Improve (QPBOI from user labels 0,
QPBO_extra.cpp:1151-1233) against the reference library, same libc rand() seed on both sides.
Frustrated ("spin glass") problems of random size so that many nodes stay unlabelled."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import stereo_amd
from helpers import glass_problem
from oracle import pyoracle as po

if not po.have_ref_qpbo():
    print("oracle/_ref/libref_qpbo.so not present"); sys.exit(0)
budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
t0, n, ran, bad = time.time(), 0, 0, 0
while time.time() - t0 < budget:
    H, W = int(rng.integers(2, 40)), int(rng.integers(2, 40))
    seed = int(rng.integers(0, 1 << 30))
    p = glass_problem(seed, H, W, field=float(rng.choice([0.5, 1.5, 3.0, 6.0])), integer=bool(rng.integers(0, 2)))
    args = (p["U0"], p["U1"], p["E00"], p["E01"], p["E10"], p["E11"])
    ref, en_r, lb_r, nu_r = po.ref_rd(*args, p["conn"], improve=True, seed=seed)
    po.ref_qpbo().ref_srand(seed)
    lab, en, lb, nu = stereo_amd.rd(*args, p["conn"].T + 1, {"improve": True})
    n += 1
    ran += int(nu_r > 0)
    # Improve only runs if nodes stay unlabelled after the weak persistencies; its result depends on
    # the permutation alone.  Without it the weakly persistent labels depend on the maximum flow
    # found (QPBO_postprocessing.cpp:10-120 walks the residual graph): equal energy is the bar there.
    ok = nu == nu_r and abs(en - en_r) <= 1e-9 * max(1.0, abs(en_r)) and (nu_r == 0 or np.array_equal(lab, ref))
    if not ok:
        bad += 1
        po.ref_qpbo().ref_srand(seed)
        lab2 = stereo_amd.rd(*args, p["conn"].T + 1, {"improve": True})[0]
        print("MISMATCH", dict(seed=seed, H=H, W=W, nu=nu, nu_r=nu_r, differ=int((lab != ref).sum()),
                               differ_on_retry=int((lab2 != ref).sum())))
print("stress improve: %d problems (%d ran Improve), %d mismatches, %.0f s" % (n, ran, bad, time.time() - t0))
sys.exit(1 if bad else 0)
