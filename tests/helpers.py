"""Seeded problem generators shared by the tests (inputs only, no solver code)."""
import numpy as np

from oracle import terms


def grid_conn(H, W):
    """(E,2) zero-based directed edges of the reference's 4-neighbourhood
    (dispmap_super.m:279-302)."""
    i1, i2 = terms.construct_neighborhood(H, W)
    return np.stack([i1, i2], 1)


def random_planes(rng, N, K, spread=20.0, slant=0.3):
    """K random disparity planes [a b c d] replicated over N pixels (4, N) each."""
    out = []
    for _ in range(K):
        P = np.zeros((4, N))
        P[0] = rng.normal() * slant
        P[1] = rng.normal() * slant
        P[2] = 1.0
        P[3] = -rng.uniform(0, spread)
        out.append(P)
    return out


def trws_problem(seed, H, W, K, kind="general", integer=False, zero_alpha_frac=0.05,
                 alpha_scale=1.0):
    """Returns dict(unary (N,K), conn (E,2), q (E,K), qprim (E,K), alphas (E,)).
    kind: 'general' random slanted planes (q != qprim); 'fronto' integer grid 0..K-1."""
    rng = np.random.default_rng(seed)
    N = H * W
    conn = grid_conn(H, W)
    E = conn.shape[0]
    if kind == "fronto":
        q = np.tile(np.arange(K, dtype=np.float64), (E, 1))
        qprim = q.copy()
    else:
        pts = terms.get_points(H, W)
        props = random_planes(rng, N, K)
        # per-pixel variation so that planes differ between neighbours
        for P in props:
            P[3] += rng.normal(size=N) * 0.5
        q, qprim = terms.trws_positions(props, conn[:, 0], conn[:, 1], pts)
    unary = rng.uniform(0, 40, size=(N, K))
    if integer:
        unary = np.round(unary / 4)
        if kind != "fronto":
            q = np.round(q)
            qprim = np.round(qprim)
    alphas = rng.uniform(0.5, 2.0, size=E) * alpha_scale
    if integer:
        alphas = np.round(alphas)
        alphas[alphas == 0] = 1
    alphas[rng.random(E) < zero_alpha_frac] = 0.0
    return dict(unary=unary, conn=conn, q=q, qprim=qprim, alphas=alphas)
