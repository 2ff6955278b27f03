"""Seeded adversarial inputs for the message certificate: single message updates whose
cones sit at or within a few delta (1e-9 x magnitude, the certificate's margin) or a few ulps of the
situations where the reference's serial lower-envelope construction and plain min-plus part ways:
tangent cone pairs (u or v keys equal), exact-slope ramps, destinations whose minimum meets the
truncation level, plateaus, near-duplicate positions around the quadratic kernel's 1e-8 rule.
Inputs only -- shared by tests/test_certificate_gpu.py and tools/stress_certificate.py."""
import numpy as np

ALPHAS = (1.0, 0.5, 2.0, 0.7, 1.3, 18.0, 216.0)
LAMBDAS = (2.0, 8.0, 0.02, 5.5)
EPS_STEPS = (0.0, 0.5, -0.5, 1.0, -1.0, 2.0, -2.0)


def _ulps(x, n):
    for _ in range(abs(n)):
        x = np.nextafter(x, np.inf if n > 0 else -np.inf)
    return x


def positions(rng, K, mode):
    if mode == "grid":
        return np.arange(K, dtype=np.float64)
    q = np.sort(rng.uniform(0, K, K))
    q += np.arange(K) * 1e-3                       # keep them distinct
    if mode == "dup":
        # exactly equal positions (a proposal's plane that IS the current plane, dispmap_super.m:158):
        # their order is the reference gateway's std::sort artefact (trws_mex.cpp:84-97)
        for _ in range(max(1, K // 6)):
            i, j = rng.choice(K, 2, replace=False)
            q[j] = q[i]
        return np.sort(q) if rng.random() < 0.5 else q[rng.permutation(K)]
    if mode == "neardup":
        for _ in range(max(1, K // 8)):
            i = int(rng.integers(0, K - 1))
            gap = 1e-8 * float(rng.choice([0.5, 1.0, 2.0, 4.0])) + float(rng.choice([0.0, 1e-17, -1e-17]))
            q[i + 1] = q[i] + max(gap, 1e-12)
        q = np.sort(q)
    return q


def heights(rng, q, alpha, lam, kind):
    """One adversarial H vector for source positions q."""
    K = len(q)
    h = rng.uniform(0, 40, K)
    if rng.random() < 0.3:
        h = np.round(h)                            # integer heights: exact ties
    scale = np.abs(h).max() + 2 * alpha * np.abs(q).max() + alpha * lam
    delta = 1e-9 * scale
    eps = float(rng.choice(EPS_STEPS)) * delta
    n_ulp = int(rng.integers(-2, 3))
    if kind == "tangent":
        for _ in range(int(rng.integers(1, 4))):
            i, j = rng.choice(K, 2, replace=False)
            sgn = float(rng.choice([-1.0, 1.0]))   # j on i's right / left arm: equal v / u keys
            h[j] = _ulps(h[i] + sgn * alpha * (q[j] - q[i]) + eps, n_ulp)
    elif kind == "ramp":
        a = int(rng.integers(0, K - 2)); b = int(rng.integers(a + 2, K + 1))
        sgn = float(rng.choice([-1.0, 1.0]))
        for k in range(a + 1, b):
            h[k] = h[k - 1] + sgn * alpha * (q[k] - q[k - 1])   # accumulated: ulp noise along the ramp
        h[a:b] += eps * rng.integers(0, 2, b - a)
        h[a:b] -= min(0.0, h[a:b].min())           # keep it low enough to matter
        if rng.random() < 0.5:                      # exact ties next to ties that are off by an ulp or two (the masked
            for k in rng.choice(np.arange(a, b), max(1, (b - a) // 4), replace=False):   # columns of an NCC volume)
                h[k] = _ulps(h[k], int(rng.integers(-2, 3)))
    elif kind == "vtrunc":
        s = int(rng.integers(0, K)); h[s] = 0.0; h[np.arange(K) != s] += 1.0
        t = int(rng.integers(0, K))
        # a second source whose cost at destination t lands on vTrunc = min h + alpha lam (+- eps)
        j = int(rng.integers(0, K))
        if j != s:
            h[j] = _ulps(alpha * lam - alpha * abs(q[t] - q[j]) + eps, n_ulp)
            if h[j] < 0:
                h[j] = 0.5
    elif kind == "plateau":
        a = int(rng.integers(0, K - 1)); b = int(rng.integers(a + 1, K + 1))
        h[a:b] = float(rng.uniform(0, 5))
        if rng.random() < 0.5:
            h[int(rng.integers(a, b))] += eps
    elif kind == "highcones":
        m = rng.random(K) < 0.3
        h[m] += 4e7                                # out-of-range proposals (dispmap_ncc.m:245)
        i, j = rng.choice(K, 2, replace=False)
        h[j] = h[i] + alpha * (q[j] - q[i]) + eps
    return h


KINDS = ("tangent", "ramp", "vtrunc", "plateau", "highcones")


def batch(seed, kernel, K, M):
    """M adversarial message updates: dict of M x K arrays + per-message alpha, one lambda."""
    rng = np.random.default_rng(seed)
    lam = float(rng.choice(LAMBDAS))
    if kernel == 2:
        lam = lam * lam
    mode = str(rng.choice(["grid", "real", "neardup", "dup"]))
    out = dict(Di=np.zeros((M, K)), msg=np.zeros((M, K)), gamma=np.ones(M), qs=np.zeros((M, K)), qd=np.zeros((M, K)),
               alpha=np.zeros(M), lam=lam, mode=mode)
    shared = positions(rng, K, mode)
    for m in range(M):
        alpha = float(rng.choice(ALPHAS))
        qs = shared if mode == "grid" else positions(rng, K, mode)
        qd = qs if (mode == "grid" or rng.random() < 0.5) else positions(rng, K, "real")
        h = heights(rng, qs, alpha, lam if kernel == 1 else np.sqrt(lam), str(rng.choice(KINDS)))
        if rng.random() < 0.25:                    # through gamma * Di - msg as the sweeps form it
            g = float(rng.choice([0.25, 0.5, 1.0 / 6, 0.125]))
            msg = rng.uniform(0, 3, K)
            out["gamma"][m] = g; out["msg"][m] = msg; out["Di"][m] = (h + msg) / g
        else:
            out["Di"][m] = h
        out["qs"][m] = qs; out["qd"][m] = qd; out["alpha"][m] = alpha
    out["shared"] = shared if mode == "grid" else None
    return out


def check(hip_messages, oracle, seed, kernel, K, M, impl="envelope"):
    """Runs one batch on the device and through the oracle; returns (mismatching rows, serial count)."""
    b = batch(seed, kernel, K, M)
    window = -1
    if b["shared"] is not None:
        lam_d = b["lam"] if kernel == 1 else np.sqrt(b["lam"] * (1 + 1e-9))
        window = int(np.floor(lam_d + 1e-12))
    got, vmin, ser = hip_messages(kernel, b["Di"], b["gamma"], b["msg"], b["qs"], b["qd"], b["alpha"], b["lam"],
                                  certificate=True, window=window if window <= 16 else -1, shared_positions=b["shared"])
    bad = []
    for m in range(M):
        want, v = oracle.update_message(kernel, b["Di"][m], float(b["gamma"][m]), b["msg"][m], b["qd"][m], b["qs"][m],
                                        float(b["alpha"][m]), b["lam"], 0, 0, impl=impl)
        if not (np.array_equal(want, got[m]) and v == vmin[m]):
            bad.append(m)
    return bad, int(ser.sum()), b
