"""rd (roof duality) golden vectors from the REFERENCE QPBO library (oracle/_ref).
Called by make_golden.py; inputs are regenerated from seeds by the tests."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from oracle import pyoracle as po  # noqa: E402
from helpers import fusion_problem, glass_problem  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))

RD_RUNS = [
    # name, kind, seed, H, W, params
    ("fuse_lin_12x14", "fusion", 201, 12, 14, dict(kernel=1, tol=8.0)),
    ("fuse_quad_12x14", "fusion", 202, 12, 14, dict(kernel=2, tol=20.0)),
    ("fuse_super_30x40", "fusion", 203, 30, 40, dict(kernel=1, tol=8.0, nonsub_boost=6.0)),
    ("fuse_int_20x25", "fusion", 204, 20, 25, dict(kernel=1, tol=8.0, integer=True)),
    ("fuse_lin_60x80", "fusion", 205, 60, 80, dict(kernel=1, tol=8.0, nonsub_boost=2.0)),
    ("glass_int_8x9", "glass", 206, 8, 9, dict(field=3.0, integer=True)),
    ("glass_real_8x9", "glass", 207, 8, 9, dict(field=3.0, integer=False)),
    ("glass_int_20x24", "glass", 208, 20, 24, dict(field=3.0, integer=True)),
]


def make_problem(kind, seed, H, W, params):
    return (fusion_problem if kind == "fusion" else glass_problem)(seed, H, W, **params)


def rd_runs():
    out = {}
    for name, kind, seed, H, W, params in RD_RUNS:
        p = make_problem(kind, seed, H, W, params)
        args = (p["U0"], p["U1"], p["E00"], p["E01"], p["E10"], p["E11"], p["conn"])
        strong, _, _, nus = po.ref_rd(*args, stage=1)
        weak, en, lb, nu = po.ref_rd(*args)
        imp, en_i, _, _ = po.ref_rd(*args, improve=True, seed=seed)
        out[name + "_strong"] = strong.astype(np.int8)
        out[name + "_weak"] = weak.astype(np.int8)
        out[name + "_improved"] = imp.astype(np.int8)
        out[name + "_scalars"] = np.array([en, lb, nu, en_i])
    np.savez_compressed(os.path.join(HERE, "rd_runs.npz"), **out)


if __name__ == "__main__":
    assert po.have_ref_qpbo()
    rd_runs()
