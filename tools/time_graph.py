#!/usr/bin/env python
"""Times the host-side graph analysis (stereo_trws_analyze) of an H x W grid: STEREO_HIP_GRAPH_VERBOSE=1
prints the phases."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from helpers import grid_conn
from stereo_amd.trws import analyze
H, W = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (2000, 3000)
t = time.time(); conn = grid_conn(H, W); print("connectivity %.1f s" % (time.time() - t), conn.shape)
t = time.time(); analyze(H * W, conn.T); print("analysis %.1f s on %d cores" % (time.time() - t, os.cpu_count()))
