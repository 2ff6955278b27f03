#!/usr/bin/env python
"""Generates the committed golden vectors.  Needs the build container: it uses
oracle/_ref (the reference's own QPBO library and TRW-S type classes compiled
from /root/reference by oracle/Makefile).  Run from the repo root:

    python tests/golden/make_golden.py

Outputs (small .npz files, inputs + expected outputs, no reference source):
  trws_messages.npz  per-message vectors through the reference's
                     TypeStereoLinear/Quadratic::Edge::UpdateMessage and AddColumn
  trws_runs.npz      full TRW-S runs: the oracle's restated MRFEnergy core driving
                     the REFERENCE message/column functions (the core itself cannot
                     be built from the reference: MRFEnergy.h includes <mex.h>)
  rd_runs.npz        roof-duality fusions through the reference QPBO library
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from oracle import pyoracle as po  # noqa: E402
from helpers import trws_problem  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))


def message_vectors():
    rng = np.random.default_rng(20260928)
    rec = dict(kernel=[], K=[], Di=[], msg=[], q=[], qprim=[], gamma=[], alpha=[], lam=[],
               dir=[], mdir=[], out=[], vmin=[], ksrc=[], col_in=[], col_out=[])
    KMAX = 24
    for trial in range(400):
        K = int(rng.integers(2, KMAX + 1))
        kernel = 1 + trial % 2
        mode = (trial // 2) % 4
        if mode == 0:
            q = rng.normal(size=K) * 10; qp = rng.normal(size=K) * 10
        elif mode == 1:   # integer positions with duplicates
            q = rng.integers(0, 10, K).astype(float); qp = rng.integers(0, 10, K).astype(float)
        elif mode == 2:   # fronto-parallel grid
            q = np.arange(K, dtype=float); qp = q.copy()
        else:             # near-duplicate positions
            q = rng.normal(size=K) * 5; qp = q + rng.normal(size=K) * 1e-9
            q[int(rng.integers(0, K))] = q[0]
        Di = rng.normal(size=K) * 20; msg = rng.normal(size=K) * 5
        if trial % 3 == 0:  # exact ties
            Di = np.round(Di); msg = np.round(msg)
        gamma = [0.25, 1 / 6, 0.125, 0.5][trial % 4]
        alpha = float(rng.choice([0.0, 1.0, 9.0, 108.0, float(rng.random() * 3)]))
        lam = float(rng.choice([8.0, 0.02, 2.0, 64.0]))
        d = int(rng.integers(0, 2)); md = int(rng.integers(0, 2))
        out, vmin = po.update_message(kernel, Di, gamma, msg, q, qp, alpha, lam, d, md, "ref")
        ks = int(rng.integers(0, K))
        col_in = rng.normal(size=K)
        col_out = po.add_column(kernel, q, qp, alpha, lam, ks, col_in, d, md, "ref")
        pad = lambda a: np.concatenate([a, np.zeros(KMAX - K)])
        for k, v in dict(kernel=kernel, K=K, Di=pad(Di), msg=pad(msg), q=pad(q), qprim=pad(qp),
                         gamma=gamma, alpha=alpha, lam=lam, dir=d, mdir=md, out=pad(out),
                         vmin=vmin, ksrc=ks, col_in=pad(col_in), col_out=pad(col_out)).items():
            rec[k].append(v)
    np.savez_compressed(os.path.join(HERE, "trws_messages.npz"),
                        **{k: np.asarray(v) for k, v in rec.items()})


RUNS = [
    # name, seed, H, W, K, kernel, kind, integer, tol, maxiter, relgap
    ("g5x6k4lin", 101, 5, 6, 4, 1, "general", False, 2.0, 4, 1e-4),
    ("g7x9k6quad", 102, 7, 9, 6, 2, "general", False, 2.0, 4, 1e-4),
    ("g12x14k8lin", 103, 12, 14, 8, 1, "general", False, 1.5, 4, 1e-4),
    ("f8x7k5lin", 104, 8, 7, 5, 1, "fronto", False, 2.0, 4, 1e-4),
    ("f8x7k5quad", 105, 8, 7, 5, 2, "fronto", False, 4.0, 4, 1e-4),
    ("i9x11k7lin", 106, 9, 11, 7, 1, "general", True, 2.0, 6, 0.0),
    ("i9x11k7quad", 107, 9, 11, 7, 2, "general", True, 4.0, 6, 0.0),
    ("if10x12k12lin", 108, 10, 12, 12, 1, "fronto", True, 3.0, 8, 0.0),
    ("g16x20k16lin", 109, 16, 20, 16, 1, "general", False, 8.0, 10, 1e-3),
]


def trws_runs():
    out = {}
    for name, seed, H, W, K, kernel, kind, integer, tol, maxiter, relgap in RUNS:
        p = trws_problem(seed, H, W, K, kind=kind, integer=integer)
        lab, en, lb, it = po.trws(kernel, p["unary"], p["conn"], p["q"], p["qprim"], p["alphas"],
                                  tol, maxiter, relgap, mode=1, use_ref_types=True)
        out[name + "_labels"] = lab.astype(np.int32)
        out[name + "_scalars"] = np.array([en, lb, it])
    np.savez_compressed(os.path.join(HERE, "trws_runs.npz"), **out)


if __name__ == "__main__":
    assert po.have_ref_types(), "oracle/_ref missing: run `make -C oracle ref` in the build container"
    message_vectors()
    trws_runs()
    try:
        from make_golden_rd import rd_runs
        rd_runs()
    except ImportError:
        pass
    print("golden vectors written to", HERE)
