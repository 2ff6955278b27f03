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


def fusion_problem(seed, H, W, kernel=1, tol=8.0, integer=False, unary_scale=40.0, weight=1.0,
                   nonsub_boost=0.0):
    """One binary fusion move on an H x W grid: current = noisy fronto-parallel planes,
    proposal = one slanted plane.  Terms follow dispmap_super.m:236-262 (via oracle.terms).
    Returns dict(U0, U1, E00, E01, E10, E11, conn (E,2))."""
    rng = np.random.default_rng(seed)
    N = H * W
    conn = grid_conn(H, W)
    pts = terms.get_points(H, W)
    cur = np.zeros((4, N)); cur[2] = 1.0
    cur[3] = -(np.round(rng.uniform(0, 30, N) / 6) * 6 + rng.normal(size=N) * 0.3)
    prop = random_planes(rng, N, 1, spread=30.0, slant=0.15)[0]
    wts = np.full(conn.shape[0], weight)
    E00, E01, E10, E11 = terms.all_pairwise_costs(kernel, wts, tol, cur, prop, conn[:, 0], conn[:, 1], pts)
    U0 = rng.uniform(0, unary_scale, N)
    U1 = rng.uniform(0, unary_scale, N)
    if nonsub_boost:
        k = rng.random(conn.shape[0]) < 0.2
        E00 = E00 + nonsub_boost * k
        E11 = E11 + nonsub_boost * k
    if integer:
        U0, U1 = np.round(U0 / 4), np.round(U1 / 4)
        E00, E01, E10, E11 = (np.round(x) for x in (E00, E01, E10, E11))
    return dict(U0=U0, U1=U1, E00=E00, E01=E01, E10=E10, E11=E11, conn=conn)


def glass_problem(seed, H, W, field=3.0, integer=True):
    """Frustrated +-J couplings with a weak random field: roof duality leaves part of the
    grid unlabelled (exercises weak persistency and Improve)."""
    rng = np.random.default_rng(seed)
    conn = grid_conn(H, W)
    E, N = conn.shape[0], H * W
    J = rng.normal(size=E) * 2
    U1 = rng.normal(size=N) * field
    if integer:
        J, U1 = np.round(J * 2), np.round(U1 * 2)
    z = np.zeros(E)
    return dict(U0=np.zeros(N), U1=U1, E00=z.copy(), E01=J.copy(), E10=J.copy(), E11=z.copy(), conn=conn)


def piecewise_planar(H, W, cell, rng, d_lo, d_hi, slant=0.3):
    """One proposal, 4 x N (pixel id = col * H + row): a random plane [a b 1 -d] per cell x cell
    block, disparity d0 in [d_lo, d_hi) at the block centre (the build's deterministic stand-in for
    the reference's SegPln proposals, dispmap_globalstereo.m:60-201, which need its segmenters)."""
    rows, cols = np.meshgrid(np.arange(H), np.arange(W), indexing="ij")
    by, bx = rows // cell, cols // cell
    nb = (by.max() + 1, bx.max() + 1)
    a = rng.normal(0, slant, nb)[by, bx]
    b = rng.normal(0, slant, nb)[by, bx]
    d0 = rng.uniform(d_lo, d_hi, nb)[by, bx]
    cx, cy = (bx + 0.5) * cell, (by + 0.5) * cell
    p4 = -(d0 + a * cx + b * cy)
    planes = np.stack([a, b, np.ones_like(a), p4])                     # 4 x H x W
    return np.asfortranarray(planes.transpose(0, 2, 1).reshape(4, H * W))


def piecewise_planar_from_disparity(H, W, cell, rng, dmap, slant=0.05, jitter=0.5):
    """One proposal, 4 x N, that follows a given H x W disparity map: per cell x cell block a plane
    [a b 1 p4] through (block centre, block median of dmap + jitter) with a small random slant, so
    that neighbouring pixels of one block share a plane (q == qprim inside, != across blocks) and
    different proposals win in different parts of the image."""
    rows, cols = np.meshgrid(np.arange(H), np.arange(W), indexing="ij")
    by, bx = rows // cell, cols // cell
    nb = (by.max() + 1, bx.max() + 1)
    med = np.zeros(nb)
    for i in range(nb[0]):
        for j in range(nb[1]):
            med[i, j] = np.median(dmap[i * cell:(i + 1) * cell, j * cell:(j + 1) * cell])
    a = rng.normal(0, slant, nb)[by, bx]
    b = rng.normal(0, slant, nb)[by, bx]
    d0 = (med + rng.normal(0, jitter, nb))[by, bx]
    cx, cy = (bx + 0.5) * cell, (by + 0.5) * cell
    p4 = -(d0 + a * cx + b * cy)
    planes = np.stack([a, b, np.ones_like(a), p4])                     # 4 x H x W
    return np.asfortranarray(planes.transpose(0, 2, 1).reshape(4, H * W))
