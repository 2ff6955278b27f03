#!/usr/bin/env python
"""Config-size golden checksums: the CPU oracle run ONCE, in the build
container, to the reference examples' real stopping rules, on the real image pairs.

    python tests/golden/make_golden_full.py [config1] [config4]

writes tests/golden/full_runs.json: per config sha256 of the inputs the run was fed (so that a
test can tell "my inputs differ" from "my solver differs"), sha256 of the int32 labels, energy,
lower bound and iteration count, plus the wall time of the oracle (a CPU figure for BASELINE.md).

config1 = BASELINE.json configs[1]: NCC volume (oracle/terms.py, dispmap_ncc.m:116-198) of the
          Teddy pair, 60 fronto-parallel labels, unary weight 40, tol 8, kernel 1, stop rule of
          dispmap_super.m:9-10 (maxiter 1000, max_relgap 1e-4).
config4 = BASELINE.json configs[4]: example_simultaneous.m on Baby2 (disp_range [0 85], factor 3,
          maxiter 3000, max_relgap 1e-5, :17-18,50-51): globalstereo unaries, 14 SegPln-style
          proposals + the current assignment -> K = 15, edge weights from the reference's own
          mean-shift segmentation (tests/golden/baby2_segments.npz).

The oracle is test infrastructure (oracle/trws_oracle.c through oracle/pyoracle.py).
"""
import hashlib
import json
import os
import sys
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
OUT = os.path.join(HERE, "full_runs.json")


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def config1_inputs():
    """-> unary (N, K), conn (E, 2), K, tol."""
    from oracle import terms
    from helpers import grid_conn
    g = np.load(os.path.join(HERE, "teddy_pair.npz"))
    im0, im1 = g["im0"].astype(np.float64), g["im1"].astype(np.float64)
    H, W, _ = im0.shape
    K = 60
    ncc = terms.compute_ncc(im0, im1, np.arange(K, dtype=np.float64), 2)         # H x W x K
    unary = 40.0 * (1.0 - ncc)                                                   # dispmap_ncc.m:107-115
    unary = np.ascontiguousarray(unary.transpose(1, 0, 2).reshape(H * W, K))     # node id = col * H + row
    return unary, grid_conn(H, W), K, 8.0


def run_config1():
    from oracle import pyoracle
    unary, conn, K, tol = config1_inputs()
    E = conn.shape[0]
    q = np.tile(np.arange(K, dtype=np.float64), (E, 1))
    t0 = time.time()
    lab, en, lb, it = pyoracle.trws(1, unary, conn, q, q, np.ones(E), tol, maxiter=1000,
                                    max_relgap=1e-4, mode=1)
    return {"what": "configs[1]: Teddy pair NCC volume 450x375x60, TRW-S kernel 1, tol 8, alphas 1, "
                    "maxiter 1000, max_relgap 1e-4 (dispmap_super.m:9-10)",
            "unary_sha256": sha(unary), "labels_sha256": sha(lab.astype(np.int32)),
            "energy": float(en), "lower_bound": float(lb), "iterations": float(it),
            "oracle_seconds": time.time() - t0, "oracle": "oracle/trws_oracle.c, envelope messages, 1 thread"}


def run_config4():
    from oracle import pyoracle
    from example_inputs import baby2_problem
    p = baby2_problem()
    t0 = time.time()
    lab, en, lb, it = pyoracle.trws(p["kernel"], p["unary"], p["conn"], p["q"], p["qprim"], p["alphas"],
                                    p["tol"], maxiter=3000, max_relgap=1e-5, mode=1)
    return {"what": "configs[4]: example_simultaneous.m on Baby2 370x413, K = 15 (14 proposals + current), "
                    "maxiter 3000, max_relgap 1e-5 (example_simultaneous.m:50-51)",
            "unary_sha256": sha(p["unary"]), "q_sha256": sha(p["q"]), "qprim_sha256": sha(p["qprim"]),
            "alphas_sha256": sha(p["alphas"]), "labels_sha256": sha(lab.astype(np.int32)),
            "energy": float(en), "lower_bound": float(lb), "iterations": float(it),
            "oracle_seconds": time.time() - t0, "oracle": "oracle/trws_oracle.c, envelope messages, 1 thread"}


def main():
    which = sys.argv[1:] or ["config1", "config4"]
    for name in which:
        res = {"config1": run_config1, "config4": run_config4}[name]()
        print(name, json.dumps(res), flush=True)
        out = json.load(open(OUT)) if os.path.exists(OUT) else {}     # (re-read: the configs may run side by side)
        out[name] = res
        json.dump(out, open(OUT, "w"), indent=1, sort_keys=True)


if __name__ == "__main__":
    main()
