"""Build-container tool (no GPU): times what of the reference compiles here -- its QPBO v1.3 library
(oracle/_ref/libref_qpbo.so, driven as cpp/rd_mex.cpp:55-100 drives it) and its TRW-S type classes
(oracle/_ref/libref_trws_types.so: typeStereoLinear.h's UpdateMessage / AddColumn inside the oracle's sweeps) --
next to the oracle's own restatement, on Teddy-sized inputs, one thread.  Prints a markdown table
(committed to BASELINE.md section 2b).   usage: time_reference_cpu.py [iters=2]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from oracle import pyoracle as po
from helpers import fusion_problem, grid_conn
from bench import synthetic_volume

iters = int(sys.argv[1]) if len(sys.argv) > 1 else 2
H, W, K = 375, 450, 60
rows = []
# --- QPBO: one Teddy-sized binary fusion move, synthetic NCC-like terms (bench.py's secondary figure)
fp = fusion_problem(5, H, W, kernel=1, tol=8.0)
args = (fp["U0"], fp["U1"], fp["E00"], fp["E01"], fp["E10"], fp["E11"], fp["conn"])
if po.have_ref_qpbo():
    t = []
    for _ in range(3):
        t0 = time.perf_counter(); r = po.ref_rd(*args); t.append(time.perf_counter() - t0)
    rows.append(("reference QPBO v1.3 library (`QPBO-v1.3.src/*.cpp`, gateway lines restated), 375x450 move", "%.1f ms per move (best of 3), %d unlabelled" % (min(t) * 1e3, int(r[3]))))
    t = []
    for _ in range(2):
        t0 = time.perf_counter(); r = po.ref_rd(*args, improve=True, seed=1); t.append(time.perf_counter() - t0)
    rows.append(("... with `improve` (`QPBO_extra.cpp:1224-1233`)", "%.1f ms per move" % (min(t) * 1e3)))
t0 = time.perf_counter(); po.rd(*args); rows.append(("oracle/qpbo_oracle.c (restatement), same move", "%.1f ms per move" % ((time.perf_counter() - t0) * 1e3)))
# --- TRW-S: noise volume 450x375x60 (fronto-parallel labels), per-iteration time from the trace
conn = grid_conn(H, W); E = conn.shape[0]
unary = synthetic_volume(H, W, K, seed=1)
q = np.tile(np.arange(K, dtype=np.float64), (E, 1))
ones = np.ones(E)
for label, kw in (("oracle/trws_oracle.c, envelope messages (what bench.py's cpu_baseline times)", dict(mode=1)),
                  ("oracle sweeps calling the REFERENCE's type classes (`typeStereoLinear.h` UpdateMessage / AddColumn)", dict(mode=1, use_ref_types=True))):
    if kw.get("use_ref_types") and not po.have_ref_types():
        continue
    t0 = time.perf_counter()
    r = po.trws(1, unary, conn, q, q, ones, 8.0, maxiter=iters, max_relgap=-1e300, want_trace=True, **kw)
    wall = time.perf_counter() - t0
    secs = float(r[4][-1, 2])
    rows.append((label + ", 450x375x60", "%.2f s per iteration (%d iterations), %.1f s setup; energy %.6f" % (secs / iters, iters, wall - secs, r[1])))
import platform
print("Machine: %s, %d cores visible, python %s, one thread per measurement.\n" % (
    [l.split(":")[1].strip() for l in open("/proc/cpuinfo") if l.startswith("model name")][0], os.cpu_count(), platform.python_version()))
print("| What | Time |\n|---|---|")
for a, b in rows:
    print("| %s | %s |" % (a, b))
