"""TEST INFRASTRUCTURE ONLY -- NumPy restatement of the MATLAB array math that
feeds the two solvers (reference: dispmap_super.m, dispmap_ncc.m,
dispmap_globalstereo.m, imrender/vgg/vgg_interp2.cxx).

Parity status: **unpinned by an executable reference** -- there is no MATLAB /
Octave in the image.  Every function follows the cited lines literally
(column-major node ids, 1-based pixel coordinates, conv2(...,'same') zero
padding, interp2 linear, first-max / last-nearest tie rules) and is the
definition the HIP kernels are compared against.

Conventions: images are (H, W, C) float64; node id = col*H + row (0-based,
column major, dispmap_super.m:281-282); points(:, id) = [x = col+1, y = row+1]
(dispmap_super.m:275-278); assignment is (4, N) planes [a b c d];
disparity = -(a*x + b*y + d)/c (dispmap_super.m:318-328).
"""
import numpy as np


# ------------------------------------------------------------------ geometry

def construct_neighborhood(H, W):
    """dispmap_super.m:279-302 -> (ind1, ind2) zero-based int64 arrays of the
    directed edges: vertical down, vertical up, horizontal right, horizontal
    left, each block in column-major order."""
    nodenr = np.arange(H * W, dtype=np.int64).reshape(W, H).T  # nodenr[r, c] = c*H + r
    start = nodenr[:-1, :].T.ravel()    # column-major flatten
    finish = nodenr[1:, :].T.ravel()
    ind1 = [start, finish]
    ind2 = [finish, start]
    start = nodenr[:, :-1].T.ravel()
    finish = nodenr[:, 1:].T.ravel()
    ind1 += [start, finish]
    ind2 += [finish, start]
    return np.concatenate(ind1), np.concatenate(ind2)


def get_points(H, W):
    """dispmap_super.m:275-278: 2 x N, x = column (1-based), y = row (1-based)."""
    cols, rows = np.meshgrid(np.arange(1, W + 1, dtype=np.float64),
                             np.arange(1, H + 1, dtype=np.float64))
    return np.stack([cols.T.ravel(), rows.T.ravel()])  # column-major order


def disparity_from_assignment(assignment, points):
    """dispmap_super.m:318-328.  sum(a(1:2,:).*points) adds a*x then b*y."""
    if np.any(assignment[2] == 0):
        raise ValueError("Infinite disparity")
    s = assignment[0] * points[0] + assignment[1] * points[1]
    return -(s + assignment[3]) / assignment[2]


def pairwise_cost(kernel, weights, p, q, tol):
    """dispmap_super.m:226-235"""
    d = p - q
    if kernel == 1:
        return weights * np.minimum(np.abs(d), tol)
    if kernel == 2:
        return weights * np.minimum(d ** 2, tol)
    raise ValueError("Unkown kernel type")


def all_pairwise_costs(kernel, weights, tol, assignment, proposal, ind1, ind2, points,
                       disp_fn=disparity_from_assignment):
    """dispmap_super.m:236-262 -> E00, E01, E10, E11 (each length E)."""
    p2 = points[:, ind2]
    q = disp_fn(assignment[:, ind2], p2)
    qprim = disp_fn(assignment[:, ind1], p2)
    new_q = disp_fn(proposal[:, ind2], p2)
    new_qprim = disp_fn(proposal[:, ind1], p2)
    E00 = pairwise_cost(kernel, weights, q, qprim, tol)
    E11 = pairwise_cost(kernel, weights, new_q, new_qprim, tol)
    E10 = pairwise_cost(kernel, weights, q, new_qprim, tol)
    E01 = pairwise_cost(kernel, weights, new_q, qprim, tol)
    return E00, E01, E10, E11


def trws_positions(proposals, ind1, ind2, points, disp_fn=disparity_from_assignment):
    """dispmap_super.m:177-183 -> q, qprim as (E, K) arrays (row e = MATLAB column e)."""
    p2 = points[:, ind2]
    q = np.stack([disp_fn(P[:, ind2], p2) for P in proposals], axis=1)
    qprim = np.stack([disp_fn(P[:, ind1], p2) for P in proposals], axis=1)
    return q, qprim


def fronto_parallel(d, N):
    """example_ncc.m:35-41: plane [0 0 1 -d] at every pixel."""
    P = np.zeros((4, N))
    P[2] = 1.0
    P[3] = -float(d)
    return P


# ----------------------------------------------------------------- NCC volume

def _conv2_same_box(a, r, scale=None):
    """conv2(a, ones(2r+1), 'same') with zero padding; summation order is the
    direct double loop over the window (rows inner), matching nothing in
    particular -- MATLAB's conv2 order is unspecified, tolerance 1e-12 relative."""
    H, W = a.shape
    pad = np.zeros((H + 2 * r, W + 2 * r))
    pad[r:r + H, r:r + W] = a
    out = np.zeros((H, W))
    for dx in range(2 * r + 1):
        for dy in range(2 * r + 1):
            out += pad[dy:dy + H, dx:dx + W]
    if scale is not None:
        out = out * scale
    return out


def _shift_image(im1, d):
    """dispmap_ncc.m:145-154: imtr(:, ceil(d+1):W, c) = interp2(im1(:,:,c), X, Y) with
    X = linspace(1, W-d, numel(y_span)); zero elsewhere."""
    H, W, Cn = im1.shape
    c0 = int(np.ceil(d + 1))          # 1-based first filled column
    n = W - c0 + 1
    out = np.zeros((H, W, Cn))
    if n <= 0:
        return out
    X = np.linspace(1.0, W - d, n)    # 1-based sample columns
    x0 = np.floor(X).astype(np.int64)
    x0 = np.clip(x0, 1, W - 1) if W > 1 else x0
    t = X - x0
    for ch in range(Cn):
        a = im1[:, :, ch]
        left = a[:, x0 - 1]
        right = a[:, np.minimum(x0, W - 1)]
        out[:, c0 - 1:, ch] = left * (1 - t) + right * t if np.any(t != 0) else left
    return out


def compute_ncc(im0, im1, disparities, patchsize=2):
    """dispmap_ncc.m:116-198 -> (H, W, D) volume."""
    im0 = np.asarray(im0, np.float64)
    im1 = np.asarray(im1, np.float64)
    H, W, _ = im0.shape
    r = patchsize
    npatch = float((2 * r + 1) ** 2)
    box = lambda a: _conv2_same_box(a, r)
    mean_scale = 1.0 / npatch / 3.0
    R0 = [im0[:, :, c] for c in range(3)]
    box0 = [box(x) for x in R0]
    mean0 = (box0[0] * mean_scale + box0[1] * mean_scale) + box0[2] * mean_scale
    t1 = (box(R0[0] ** 2) + box(R0[1] ** 2)) + box(R0[2] ** 2)
    t2 = (mean0 * box0[0] + mean0 * box0[1]) + mean0 * box0[2]
    t4 = npatch * 3 * mean0 ** 2
    norm0 = np.sqrt((t1 - 2 * t2 + t4).astype(np.complex128))
    out = np.zeros((H, W, len(disparities)))
    for i, d in enumerate(disparities):
        bnd = np.zeros((H, W))
        bnd[:, int(np.floor(d + 1 + 0.5)) - 1:] = 1   # MATLAB round(): half away from zero
        tr = _shift_image(im1, d)
        T = [tr[:, :, c] for c in range(3)]
        boxT = [box(x) for x in T]
        meanT = (boxT[0] * mean_scale + boxT[1] * mean_scale) + boxT[2] * mean_scale
        u1 = (box(T[0] ** 2) + box(T[1] ** 2)) + box(T[2] ** 2)
        u2 = (meanT * boxT[0] + meanT * boxT[1]) + meanT * boxT[2]
        u4 = npatch * 3 * meanT ** 2
        normT = np.sqrt((u1 - 2 * u2 + u4).astype(np.complex128))
        c1 = (box(R0[0] * T[0]) + box(R0[1] * T[1])) + box(R0[2] * T[2])
        c2 = (mean0 * boxT[0] + mean0 * boxT[1]) + mean0 * boxT[2]
        c3 = (meanT * box0[0] + meanT * box0[1]) + meanT * box0[2]
        c4 = npatch * 3 * meanT * mean0
        with np.errstate(divide="ignore", invalid="ignore"):
            ncci = (c1 - c2 - c3 + c4) / norm0 / normT
        bad = ~(np.isfinite(ncci.real) & np.isfinite(ncci.imag))
        ncci[bad] = 0
        ncci[~(bnd >= 1 - 1e-8)] = 0
        out[:, :, i] = ncci.real
    return out


def interpolate_ncc(ncc, disparities, t2, y2, okdepth):
    """dispmap_ncc.m:250-276; t2 is 1-based like MATLAB."""
    d = np.asarray(disparities, np.float64)
    H, W, D = ncc.shape
    t1 = np.where(okdepth, t2 - 1, t2)
    t3 = np.where(okdepth, t2 + 1, t2)
    d1, d2, d3 = d[t1 - 1], d[t2 - 1], d[t3 - 1]
    rows, cols = np.meshgrid(np.arange(H), np.arange(W), indexing="ij")
    y1 = ncc[rows, cols, t1 - 1]
    y3 = ncc[rows, cols, t3 - 1]
    with np.errstate(divide="ignore", invalid="ignore"):
        a = y1 / (d1 - d2) / (d1 - d3)
        b = y2 / (d2 - d1) / (d2 - d3)
        c = y3 / (d3 - d1) / (d3 - d2)
        r = a + b + c
        p = -(a * (d2 + d3) + b * (d1 + d3) + c * (d1 + d2))
        q = a * d2 * d3 + b * d1 * d3 + c * d1 * d2
    return r, p, q, d2


def sample_ncc_from_disp(ncc, disparities, disps):
    """dispmap_ncc.m:222-249: disps (N,) column-major -> nccs (H, W)."""
    d = np.asarray(disparities, np.float64)
    H, W, D = ncc.shape
    x = np.asarray(disps, np.float64).reshape(W, H).T
    t2 = np.ones((H, W), np.int64)
    smallest = np.abs(x - d[0])
    y2 = np.ones((H, W))
    for i in range(D):
        nd = np.abs(x - d[i])
        m = nd <= smallest
        t2[m] = i + 1
        y2[m] = ncc[:, :, i][m]
        smallest[m] = nd[m]
    ok = (t2 < D) & (t2 > 1)
    good = (x <= d.max()) & (x >= d.min())
    r, p, q, _ = interpolate_ncc(ncc, d, t2, y2, ok)
    with np.errstate(invalid="ignore"):
        out = r * x ** 2 + p * x + q
    out[t2 == 1] = ncc[:, :, 0][t2 == 1]
    out[t2 == D] = ncc[:, :, -1][t2 == D]
    out[~good] = -1e6
    return out


def ncc_unary_cost(ncc, disparities, unary_weight, assignment, points):
    """dispmap_ncc.m:107-115 -> (N,) column-major."""
    disps = disparity_from_assignment(assignment, points)
    nccs = sample_ncc_from_disp(ncc, disparities, disps)
    return unary_weight * (1 - nccs.T.ravel())


def best_disp_from_ncc(ncc, disparities):
    """dispmap_ncc.m:208-221 -> (H, W) WTA disparity with parabola refinement."""
    D = ncc.shape[2]
    t2 = np.argmax(ncc, axis=2) + 1          # first max, like MATLAB
    y2 = np.max(ncc, axis=2)
    ok = (t2 < D) & (t2 > 1)
    r, p, q, d2 = interpolate_ncc(ncc, disparities, t2, y2, ok)
    with np.errstate(divide="ignore", invalid="ignore"):
        best = -p / r / 2
    best[~ok] = d2[~ok]
    return best


# ----------------------------------------------------------- globalstereo unary

def vgg_interp2_linear(A, X, Y, oobv):
    """imrender/vgg/vgg_interp2.cxx:245-322, 'linear' branch, 1-based X (col) / Y (row)."""
    A = np.asarray(A, np.float64)
    if A.ndim == 2:
        A = A[:, :, None]
    h, w, col = A.shape
    X = np.asarray(X, np.float64)
    Y = np.asarray(Y, np.float64)
    out = np.full((X.shape[0], col), float(oobv))
    for i in range(X.shape[0]):
        xx, yy = X[i], Y[i]
        if xx >= 1 and yy >= 1:
            if xx < w:
                if yy < h:
                    x = int(xx); y = int(yy); u = xx - x; v = yy - y
                    for c in range(col):
                        a00 = A[y - 1, x - 1, c]; a01 = A[y - 1, x, c]
                        a10 = A[y, x - 1, c]; a11 = A[y, x, c]
                        o = a00 + (a01 - a00) * u
                        o += ((a10 - o) + (a11 - a10) * u) * v
                        out[i, c] = o
                elif yy == h:
                    x = int(xx); u = xx - x
                    for c in range(col):
                        a0 = A[h - 1, x - 1, c]; a1 = A[h - 1, x, c]
                        out[i, c] = a0 + (a1 - a0) * u
            elif xx == w:
                if yy < h:
                    y = int(yy); v = yy - y
                    for c in range(col):
                        a0 = A[y - 1, w - 1, c]; a1 = A[y, w - 1, c]
                        out[i, c] = a0 + (a1 - a0) * v
                elif yy == h:
                    out[i] = A[h - 1, w - 1]
    return out


def globalstereo_rescale(disps, d_min, d_step):
    """dispmap_globalstereo.m:336-345: disparity map in units of the search range."""
    return (disps - d_min) / d_step


def globalstereo_unary_cost(im0, im1, P2, d_min, d_step, col_thresh, assignment, points):
    """dispmap_globalstereo.m:355-375 + ephoto (:405).  im0/im1 (H, W, C) doubles, P2 the
    4 x 3 matrix self.P(:,:,2) (already permuted, :43), assignment (4, N) -> (N,)."""
    im0 = np.asarray(im0, np.float64)
    im1 = np.asarray(im1, np.float64)
    H, W, Cn = im0.shape
    dhat = globalstereo_rescale(disparity_from_assignment(assignment, points), d_min, d_step)
    disp = d_step * (dhat + d_min)                       # sic: exact only for d_min = 0
    X = points[0]
    Y = points[1]
    T = [(X * P2[0, k] + Y * P2[1, k]) + (1.0 * P2[2, k] + disp * P2[3, k]) for k in range(3)]
    Nrm = 1.0 / T[2]
    T0 = T[0] * Nrm
    T1 = T[1] * Nrm
    R = im0.transpose(1, 0, 2).reshape(H * W, Cn)         # column-major pixels
    M = vgg_interp2_linear(im1, T0, T1, -1000.0) - R
    ssd = np.zeros(H * W)
    for c in range(Cn):
        ssd = ssd + M[:, c] ** 2
    return np.log(2.0) - np.log(np.exp(ssd * (-1.0 / (col_thresh * Cn))) + 1.0)


# ----------------------------------------------- globalstereo: construction from the raw arguments
# ojw_default_options('cvpr08') as dispmap_globalstereo uses it (imrender/ojw/ojw_default_options.m:58-80)
GLOBALSTEREO_DEFAULTS = dict(disp_thresh=0.02, col_thresh=30.0, lambda_l=9.0, lambda_h=108.0, connect=4, improve=1)


def globalstereo_setup(P, disp_range, disparity_factor, segment, n_images, kernel, options=None):
    """dispmap_globalstereo.m:29-57 + preprocess (:377-414), from the RAW constructor arguments:
      P2      = self.P(:,:,2) after `permute(P, [2 1 3])` (:43): 4 x 3;
      disps   = disp_range(1)*f : disp_range(2)*f, sorted descending (:48-49) -- a MATLAB colon with
                unit step, lo + (0 : floor(hi - lo));  d_min = disps(end), d_step = disps(1) - d_min (:51-52);
      weights = EW * (num_in / ((connect == 8) + 1)) with EW = lambda_h where the two pixels of a
                neighbourhood edge lie in the same segment, lambda_l otherwise (:397-403), edges in
                construct_neighborhood order;
      kernel 2: weights /= tol, tol = tol^2 (:410-413);  tol = options.disp_thresh (:32);
      improve = options.improve > 0 (:393).
    `segment` (H, W) stands for vgg_segment_ms' label image (:391; the segmenter is out of scope)."""
    opt = dict(GLOBALSTEREO_DEFAULTS)
    opt.update(options or {})
    P = np.asarray(P, np.float64)
    first = P[:, :, 0].reshape(-1, order="F")[[0, 1, 2, 3, 4, 5, 8]]        # P([1:6 9]), column-major linear index
    if np.max(np.abs(first - np.array([1, 0, 0, 0, 1, 0, 1.0]))) > 1e-12:
        raise ValueError("First image must be reference image")
    P2 = np.ascontiguousarray(P[:, :, 1].T)                                 # 4 x 3
    lo, hi = disp_range[0] * disparity_factor, disp_range[1] * disparity_factor
    n = int(np.floor(hi - lo + 1e-10)) + 1                                  # colon: lo, lo+1, ..., <= hi
    disps = np.sort(lo + np.arange(n, dtype=np.float64))[::-1]
    d_min = float(disps[-1])
    d_step = float(disps[0] - d_min)
    seg = np.asarray(segment)
    H, W = seg.shape
    ind1, ind2 = construct_neighborhood(H, W)
    s = seg.T.reshape(-1)                                                   # column-major pixel order
    same = s[ind1] == s[ind2]
    ew = same * float(opt["lambda_h"]) + (~same) * float(opt["lambda_l"])
    ew = ew * (n_images / ((opt["connect"] == 8) + 1))
    tol = float(opt["disp_thresh"])
    if kernel == 2:
        ew = ew / tol
        tol = tol ** 2
    return dict(P2=P2, d_min=d_min, d_step=d_step, weights=ew, tol=tol, improve=bool(opt["improve"] > 0),
                col_thresh=float(opt["col_thresh"]))


# ----------------------------------------------------- dispmap_ncc: local plane proposals
def fit_plane_to_points(points, kernel):
    """dispmap_ncc.m:67-92.  points (3, n) = [x; y; disparity].  Kernel 1: 20 rounds of iteratively
    reweighted least squares, each the last right singular vector of w .* cost_func with
    w = sqrt(|cost_func * v|) of the previous round (ones first); kernel 2: one SVD.
    p(4) = -(p(1:3)' * mean(points, 2)), p = p / p(3).  (MATLAB's svd is outside the reference tree:
    LAPACK's here; a singular vector's sign cancels in p / p(3).)"""
    points = np.asarray(points, np.float64)
    c = points.mean(axis=1, keepdims=True)
    cost = -(points - c).T                                                  # n x 3
    p = np.zeros(4)
    if kernel == 1:
        w = np.ones((cost.shape[0], 1))
        for _ in range(20):
            v = np.linalg.svd(w * cost, full_matrices=False)[2][-1]
            p[:3] = v
            w = np.sqrt(np.abs(cost @ v)).reshape(-1, 1)
    else:
        p[:3] = np.linalg.svd(cost, full_matrices=False)[2][-1]
    p[3] = -(p[:3] @ points.mean(axis=1))
    return p / p[2]


def plane_from_neighbourhood(best_disp, x, y, r, kernel):
    """dispmap_ncc.m:48-66 (generate_new_plane_RANSAC): the plane fitted to the winner-takes-all
    disparities of the pixels STRICTLY closer than r to (x, y) (1-based, x = column).  best_disp (H, W).
    Returns (plane (4,), number of points)."""
    best = np.asarray(best_disp, np.float64)
    H, W = best.shape
    pts = get_points(H, W)
    ids = np.sqrt((pts[0] - x) ** 2 + (pts[1] - y) ** 2) < r
    return fit_plane_to_points(np.vstack([pts[:, ids], best.T.reshape(-1)[ids]]), kernel), int(ids.sum())


def plane_lattice(best_disp, kernel, radius=5, first=10, step=50):
    """example_ncc.m:24-32: `for x = 10:50:W, for y = 10:50:H` one local fit each, in that order."""
    H, W = np.asarray(best_disp).shape
    return [plane_from_neighbourhood(best_disp, x, y, radius, kernel)[0]
            for x in range(first, W + 1, step) for y in range(first, H + 1, step)]


# ------------------------------------------------------- dispmap_globalstereo.segpln (second half)
def _interp2_linear_vec(A, X, Y, oobv):
    """vgg_interp2_linear (above) for many points at once: the same expressions per point, NumPy arrays."""
    A = np.asarray(A, np.float64)
    if A.ndim == 2:
        A = A[:, :, None]
    h, w, col = A.shape
    X = np.asarray(X, np.float64); Y = np.asarray(Y, np.float64)
    out = np.full((X.shape[0], col), float(oobv))
    with np.errstate(invalid="ignore"):
        ok = (X >= 1) & (Y >= 1) & (X <= w) & (Y <= h)          # (NaN coordinates fail every comparison: oobv)
    xi = np.where(ok, X, 1.0).astype(np.int64); yi = np.where(ok, Y, 1.0).astype(np.int64)
    u = np.where(ok, X, 1.0) - xi; v = np.where(ok, Y, 1.0) - yi
    x1 = np.minimum(xi, w - 1); y1 = np.minimum(yi, h - 1)      # index of the "+1" sample; unused (weight 0) on the last column / row
    for c in range(col):
        a00 = A[yi - 1, xi - 1, c]; a01 = A[yi - 1, x1, c]; a10 = A[y1, xi - 1, c]; a11 = A[y1, x1, c]
        inner = a00 + (a01 - a00) * u
        inner = inner + ((a10 - inner) + (a11 - a10) * u) * v
        lastrow = a00 + (a01 - a00) * u                          # yy == h
        lastcol = a00 + (a10 - a00) * v                          # xx == w
        val = np.where(xi == w, np.where(yi == h, a00, lastcol), np.where(yi == h, lastrow, inner))
        out[:, c] = np.where(ok, val, float(oobv))
    return out


def segpln_wta(images, P, disps, col_thresh=30.0, window=2, min_corr=0.07):
    """dispmap_globalstereo.m:60-122: the winner-takes-all disparity map behind the SegPln proposals.
    For every image a and every disparity of `disps` (descending, :49) the reference image's pixels are
    projected (X = WC P(:,1:3,a)', d = disps(b) P(:,4,a), :83-91), the image sampled bilinearly with oobv
    -1000 (:94), ephoto of the colour difference (:98, :405) averaged over the (2 window + 1)^2 box
    ('valid', :99) and accumulated; normalised with the FIRST pixel's out-of-range value (:105-106),
    first maximum over the disparities (:110), disparity of the winner, 0 where the score is below
    `min_corr` (:111-112), mirrored back to full size (:113).  images: list of (H, W, C); P (3, 4, n).
    Returns (H, W) disparities."""
    ims = [np.asarray(im, np.float64) for im in images]
    R = np.clip(np.floor(ims[0] + 0.5), 0, 255)                   # R = uint8(images{1}) (:69): MATLAB rounds half away from zero
    H, W, C = R.shape
    Rvec = R.transpose(1, 0, 2).reshape(H * W, C)
    Xg, Yg = np.meshgrid(np.arange(1, W + 1, dtype=np.float64), np.arange(1, H + 1, dtype=np.float64))
    WC = np.stack([Xg.T.reshape(-1), Yg.T.reshape(-1), np.ones(H * W)], 1)   # column-major pixel order
    P = np.asarray(P, np.float64)
    disps = np.asarray(disps, np.float64)
    wn = 2 * window + 1
    Hv, Wv = H - 2 * window, W - 2 * window
    ephoto = lambda F: np.log(2.0) - np.log(np.exp(np.sum(F ** 2, 1) * (-1.0 / (col_thresh * C))) + 1.0)
    corr = np.zeros((Hv, Wv, len(disps)))
    for a in range(len(ims)):
        # X = WC * P(:,1:3,a)' (:83), every entry as the explicit sum (x P(k,1) + y P(k,2)) + 1 P(k,3)
        X = np.stack([(WC[:, 0] * P[k, 0, a] + WC[:, 1] * P[k, 1, a]) + 1.0 * P[k, 2, a] for k in range(3)], 1)
        P_ = P[:, 3, a]
        for b, dv in enumerate(disps):
            d = dv * P_
            with np.errstate(divide="ignore", invalid="ignore"):
                Z = 1.0 / (X[:, 2] + d[2])
                Y = _interp2_linear_vec(ims[a], (X[:, 0] + d[0]) * Z, (X[:, 1] + d[1]) * Z, -1000.0)
            Y = ephoto(Y - Rvec).reshape(W, H).T                  # (H, W)
            t = np.zeros((Hv, W))
            for i in range(wn):                                   # conv2(filt, filt', ., 'valid'): columns, then rows
                t = t + (1.0 / wn) * Y[i:i + Hv, :]
            o = np.zeros((Hv, Wv))
            for j in range(wn):
                o = o + (1.0 / wn) * t[:, j:j + Wv]
            corr[:, :, b] += o
    X1 = ephoto((-1000.0 - Rvec)[:1])[0] * len(ims)
    corr = (X1 - corr) / X1
    best = np.argmax(corr, axis=2)                                # first maximum
    score = np.max(corr, axis=2)
    out = disps[best]
    out[score < min_corr] = 0.0
    return np.pad(out, window, mode="symmetric")


def _splitmix64(x):
    x = (x + 0x9E3779B97F4A7C15) & 0xFFFFFFFFFFFFFFFF
    z = x
    z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & 0xFFFFFFFFFFFFFFFF
    z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & 0xFFFFFFFFFFFFFFFF
    return z ^ (z >> 31)


def segpln_sample(seed, segment, trial, n):
    """Three distinct indices in [0, n): what `sam = randperm(len); sam = sam(1:3)` (:429-430) draws, from a
    counter-based generator (MATLAB's stream cannot be reproduced: the draw is an INPUT of the parity test,
    and both sides take it from this function): k-th index = splitmix64(seed, segment, trial, attempt) mod n,
    redrawn while it repeats an earlier one."""
    out = []
    attempt = 0
    while len(out) < 3:
        v = _splitmix64((int(seed) * 0x100000001B3 + int(segment) * 0x1000193 + int(trial) * 64 + attempt) & 0xFFFFFFFFFFFFFFFF) % int(n)
        attempt += 1
        if v not in out:
            out.append(int(v))
    return out


def _solve3(a, b):
    """3 x 3 solve by Cramer's rule, every product and sum in this fixed association (the device kernel makes
    the same operations in the same order, no contraction): a (3, 3) rows, b (3,)."""
    a11, a12, a13 = a[0]; a21, a22, a23 = a[1]; a31, a32, a33 = a[2]
    c11 = a22 * a33 - a23 * a32; c12 = a21 * a33 - a23 * a31; c13 = a21 * a32 - a22 * a31
    det = (a11 * c11 - a12 * c12) + a13 * c13
    b1, b2, b3 = b
    d1 = (b1 * c11 - a12 * (b2 * a33 - a23 * b3)) + a13 * (b2 * a32 - a22 * b3)
    d2 = (a11 * (b2 * a33 - a23 * b3) - b1 * c12) + a13 * (a21 * b3 - b2 * a31)
    d3 = (a11 * (a22 * b3 - b2 * a32) - a12 * (a21 * b3 - b2 * a31)) + b1 * c13
    with np.errstate(divide="ignore", invalid="ignore"):
        return np.array([np.float64(d1) / np.float64(det), np.float64(d2) / np.float64(det), np.float64(d3) / np.float64(det)])


def _lstsq3(pts, mask):
    """pts(mask, :) \\ -1 for an n x 3 system, as the normal equations A'A x = A'(-1) solved by _solve3, the six
    moments and three right-hand sides summed as the device's wave does: lane l of 64 adds its points
    l, l + 64, ... in index order, then the lanes are folded 32, 16, 8, 4, 2, 1 apart (lane l += lane l + s).
    (MATLAB's mldivide works by QR and is outside the reference tree: this restatement IS the definition.)"""
    n = pts.shape[0]
    acc = np.zeros((64, 9))
    for l in range(64):
        for i in range(l, n, 64):
            if mask[i]:
                x, y, z = pts[i]
                acc[l] += np.array([x * x, x * y, x * z, y * y, y * z, z * z, -x, -y, -z])
    s = 32
    while s >= 1:
        acc[:s] = acc[:s] + acc[s:2 * s]
        s //= 2
    xx, xy, xz, yy, yz, zz, bx, by, bz = acc[0]
    return _solve3(np.array([[xx, xy, xz], [xy, yy, yz], [xz, yz, zz]]), np.array([bx, by, bz]))


def _segpln_nsamples(ni, n, pf=3, conf=0.95):
    q = 1.0
    for k in range(pf):
        q = q * ((ni - pf + 1 + k) / (n - pf + 1 + k))       # prod([(ni-pf+1):ni] ./ [(ptNum-pf+1):ptNum])
    if (1 - q) < np.finfo(np.float64).eps:
        return 1.0
    with np.errstate(divide="ignore"):
        cnt = np.log(1 - conf) / np.log(1 - q)
    return max(cnt, 1.0)


def segpln_rplane(pts, th, seed, segment, max_sam=500):
    """LO-RANSAC of dispmap_globalstereo.m:417-450 on pts (n, 3): returns the inlier mask."""
    n = pts.shape[0]
    inls = np.zeros(n, bool)
    max_i, no_sam = 3, 0
    max_sam = float(max_sam)
    div = np.array([-1.0, -1.0, -1.0])
    with np.errstate(invalid="ignore", over="ignore"):
        while no_sam < max_sam:
            no_sam += 1
            sam = segpln_sample(seed, segment, no_sam, n)
            N = _solve3(pts[sam], div)
            dist = np.abs(((pts[:, 0] * N[0] + pts[:, 1] * N[1]) + pts[:, 2] * N[2]) + 1.0)
            v = dist < th
            no_i = int(v.sum())
            if max_i < no_i:
                N = _lstsq3(pts, v)
                dist = np.abs(((pts[:, 0] * N[0] + pts[:, 1] * N[1]) + pts[:, 2] * N[2]) + 1.0)
                v = dist < th
                if int(v.sum()) > int(inls.sum()):
                    inls = v
                    max_i = no_i
                    max_sam = min(max_sam, _segpln_nsamples(int(inls.sum()), n))
    return inls


def segpln_planes(wta, segments, seed=0, rt=0.1):
    """dispmap_globalstereo.m:140-197 for ONE segmentation map: wta (H, W) winner-takes-all disparities,
    segments (H, W) labels 1 .. S (0: no segment).  World coordinates [x y 1] / d (:141-145), per segment the
    points with WC(:,3) ~= 0 (:168), LO-RANSAC when more than three (:171-175), the least-squares plane of
    the inliers when more than two (:177-186): proposal columns [N1 N2 1 N3] for the segment's pixels,
    [0 0 1 0] elsewhere (:157-161); NaN / Inf -> 1e-100 (:193-196).  Returns (proposal (4, N), planes (S, 3),
    inlier counts (S,))."""
    wta = np.asarray(wta, np.float64)
    seg = np.asarray(segments).astype(np.int64)
    H, W = wta.shape
    d = wta.T.reshape(-1)
    s = seg.T.reshape(-1)
    pts2 = get_points(H, W)
    with np.errstate(divide="ignore", invalid="ignore"):
        z = 1.0 / d
        WC = np.stack([z * pts2[0], z * pts2[1], z], 1)
    S = int(s.max()) if s.size else 0
    prop = np.zeros((4, H * W)); prop[2] = 1.0
    planes = np.zeros((S, 3)); ninl = np.zeros(S, np.int64)
    for a in range(1, S + 1):
        idx = np.nonzero(s == a)[0]
        if idx.size == 0:
            continue
        N = WC[idx]
        N = N[N[:, 2] != 0]
        keep = np.ones(N.shape[0], bool)
        if N.shape[0] > 3:
            keep = segpln_rplane(N, rt, seed, a)
        ninl[a - 1] = int(keep.sum())
        if int(keep.sum()) > 2:
            with np.errstate(invalid="ignore", over="ignore"):
                N_ = _lstsq3(N, keep)
            planes[a - 1] = N_
            prop[0, idx] = N_[0]; prop[1, idx] = N_[1]; prop[2, idx] = 1.0; prop[3, idx] = N_[2]
    prop[~np.isfinite(prop)] = 1e-100
    return prop, planes, ninl
