"""Development tool: the trws.m boundary with HOST buffers (K x E q/qprim materialised, as
dispmap_super.simultaneous_fusion hands them over) -- PCIe-inclusive rate for DESIGN.md."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import stereo_amd
from bench import synthetic_volume
from helpers import grid_conn
H, W, K, iters = 375, 450, 60, 20
conn = grid_conn(H, W); E = conn.shape[0]
unary = synthetic_volume(H, W, K, 1)
q = np.asfortranarray(np.tile(np.arange(K, dtype=np.float64)[:, None], (1, E)))
al = np.ones(E)
stereo_amd.trws(1, unary.T, conn.T + 1, q, q, al, 8.0, {"maxiter": 1, "max_relgap": -1e300})
t = time.time()
lab, en, lb, it = stereo_amd.trws(1, unary.T, conn.T + 1, q, q, al, 8.0, {"maxiter": iters, "max_relgap": -1e300})
dt = time.time() - t
print("stereo_trws boundary call: %d iterations in %.3f s -> %.1f it/s PCIe+setup inclusive (energy %.6f lb %.6f)" % (it, dt, it / dt, en, lb))
t = time.time()
stereo_amd.trws(1, unary.T, conn.T + 1, q, q, al, 8.0, {"maxiter": 1, "max_relgap": -1e300})
print("of which setup + 1 iteration: %.3f s" % (time.time() - t))
