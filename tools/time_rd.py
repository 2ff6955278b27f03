"""Development tool: time one Teddy-sized binary fusion on the GPU and in the reference."""
import sys, time, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import stereo_amd
from helpers import fusion_problem
from oracle import pyoracle as po
H, W = int(sys.argv[1]) if len(sys.argv) > 1 else 375, int(sys.argv[2]) if len(sys.argv) > 2 else 450
p = fusion_problem(5, H, W, kernel=1, tol=8.0)
args = (p["U0"], p["U1"], p["E00"], p["E01"], p["E10"], p["E11"])
stereo_amd.rd(*args, p["conn"].T + 1, {})
t = time.time(); lab, en, lb, nu = stereo_amd.rd(*args, p["conn"].T + 1, {}); tg = time.time() - t
t = time.time(); ref = po.ref_rd(*args, p["conn"]); tr = time.time() - t
print("gpu %.4fs ref %.4fs labels equal %s en %.6f/%.6f lb %.6f/%.6f unl %g/%g" % (
    tg, tr, np.array_equal(lab, ref[0]), en, ref[1], lb, ref[2], nu, ref[3]))

from stereo_amd.rd import RdPlan
plan = RdPlan(H * W, p["conn"].T)
plan.solve(*args)
t = time.time()
for _ in range(5):
    labp, enp, lbp, nup = plan.solve(*args)
tp = (time.time() - t) / 5
print("plan %.4fs per move (%.1f moves/s), labels equal %s" % (tp, 1 / tp, np.array_equal(labp, ref[0])))
