"""-m gpu: the config-size golden checksums (VERDICT r4 'run the real-pair
full-size configs').  tests/golden/full_runs.json holds what the CPU oracle produced ONCE in the build
container when run to the reference examples' real stopping rules (tests/golden/make_golden_full.py):
sha256 of the labels, energy, lower bound, iteration count -- and sha256 of the inputs, so that
"my inputs differ" (another NumPy / libm) is told apart from "my solver differs".  The HIP path must
reach the same four values: labels hash equal, energy / bound / iterations equal as numbers."""
import hashlib
import json
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def _golden(name):
    path = os.path.join(GOLD, "full_runs.json")
    if not os.path.exists(path):
        pytest.skip("tests/golden/full_runs.json not generated")
    runs = json.load(open(path))
    if name not in runs:
        pytest.skip("tests/golden/full_runs.json has no %s" % name)
    return runs[name]


def test_configs1_teddy_ncc_trws_to_the_stop_of_dispmap_super(hip):
    """configs[1]: NCC volume of the Teddy pair x 60 fronto-parallel labels, TRW-S kernel 1, tol 8, alphas 1,
    maxiter 1000 / max_relgap 1e-4 (dispmap_super.m:9-10): the run ends on the gap test, so the iteration count
    is under test as well.  The unary is the oracle's NumPy NCC volume (the golden run's input, hash-checked)."""
    from make_golden_full import config1_inputs
    from stereo_amd.trws import TrwsPlan
    want = _golden("config1")
    unary, conn, K, tol = config1_inputs()
    assert _sha(unary) == want["unary_sha256"], "this host's NumPy built another NCC volume than the build container's"
    plan = TrwsPlan(1, K, unary.shape[0], conn.T)
    plan.upload(unary.T, np.ones(conn.shape[0]), tol, positions=np.arange(K, dtype=np.float64))
    plan.iterate(1000, max_relgap=1e-4)
    lab, en, lb, it = plan.result()
    assert it == want["iterations"], (it, want["iterations"])
    assert en == want["energy"] and lb == want["lower_bound"], (en, want["energy"], lb, want["lower_bound"])
    assert _sha(lab.astype(np.int32)) == want["labels_sha256"]
    assert 1 < it < 1000 and (en - lb) / en < 1e-4


def test_configs4_baby2_simultaneous_fusion_to_the_examples_stop(hip):
    """configs[4]: example_simultaneous.m on Baby2 (tests/example_inputs.py: the reference's own segmentation
    for the edge weights, the 14 SegPln proposals + the current assignment -> K = 15, general planes), through
    the trws.m boundary (hip.trws -> stereo_trws, K x E host arrays), maxiter 3000 / max_relgap 1e-5
    (example_simultaneous.m:50-51)."""
    from example_inputs import baby2_problem
    want = _golden("config4")
    p = baby2_problem()
    for key in ("unary", "q", "qprim", "alphas"):
        assert _sha(p[key]) == want[key + "_sha256"], "this host built another `%s` than the build container" % key
    lab, en, lb, it = hip.trws(p["kernel"], p["unary"].T, p["conn"].T + 1, p["q"].T, p["qprim"].T, p["alphas"], p["tol"],
                               dict(maxiter=3000, max_relgap=1e-5))
    assert it == want["iterations"], (it, want["iterations"])
    assert en == want["energy"] and lb == want["lower_bound"], (en, want["energy"], lb, want["lower_bound"])
    assert _sha(lab.astype(np.int32)) == want["labels_sha256"]
