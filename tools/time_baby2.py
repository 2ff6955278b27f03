#!/usr/bin/env python
"""Development tool (MI355X box): BASELINE.json configs[4] (tests/example_inputs.baby2_problem) on a resident plan --
ms per iteration, share of the messages that take the serial construction.   tools/time_baby2.py [iterations]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from example_inputs import baby2_problem
from stereo_amd.trws import TrwsPlan
iters = int(sys.argv[1]) if len(sys.argv) > 1 else 100
p = baby2_problem()
N, K = p["unary"].shape
E = p["conn"].shape[0]
plan = TrwsPlan(p["kernel"], K, N, p["conn"].T)
plan.upload(p["unary"].T, p["alphas"], p["tol"], q=p["q"].T, qprim=p["qprim"].T)
plan.iterate(5, max_relgap=-1e300)
plan.serial_messages(reset=True)
t = time.perf_counter()
plan.iterate(iters, max_relgap=-1e300)
dt = (time.perf_counter() - t) / iters
ser = plan.serial_messages()
_, en, lb, it = plan.result(want_labels=False)
print("baby2 %dx%dx%d: path %d, %.2f ms/iter (%.1f it/s), serial messages %d = %.2f %% of the messages, energy %.4f lb %.4f" % (
    p["W"], p["H"], K, plan.path(), dt * 1e3, 1 / dt, ser, 100.0 * ser / (2.0 * E * iters), en, lb))
eq = (p["q"][:, :, None] == p["q"][:, None, :]).sum(axis=(1, 2)) - K
print("edges with at least one pair of equal positions in q: %.1f %%; mean equal pairs per edge %.1f" % (100.0 * (eq > 0).mean(), eq.mean() / 2))
