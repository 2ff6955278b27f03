"""Host-side mirror of the reference's problem classes (dispmap_super.m,
dispmap_ncc.m, dispmap_globalstereo.m): same method names, argument meaning and
error behaviour, every array operation and both solvers running on the GPU through
the C ABI.  Planes are 4 x N arrays, images H x W x C doubles.

The proposal generators (local plane fits, segpln) and the two segmenters behind
dispmap_globalstereo (stereo_amd/segment.py) run through the same library; figures are
not mirrored.
"""
import numpy as np

from . import terms as T
from ._lib import StereoHipError
from .fusion import FusionContext, PlaneProposal
from .rd import RdPlan, rd  # noqa: F401
from .trws import trws


class dispmap_super:
    """dispmap_super.m.  Subclasses provide unary_cost()."""

    def __init__(self, images, kernel):
        self.images = [np.asarray(im, dtype=np.float64) for im in images]
        self.sz = self.images[0].shape[:2]
        self.maxiter = 1000          # dispmap_super.m:9
        self.host_stack_budget_bytes = 8 << 30   # binary_fuse_until_convergence: largest 4 x N x n host array built for the native call
        self._max_relgap = 1e-4      # dispmap_super.m:10
        self._improve = False        # dispmap_super.m:13
        self._assignment = None
        self.stored_energy = np.inf
        self._kernel = kernel
        self.neighborhood = T.construct_neighborhood(*self.sz)          # zero based 2 x E
        self.points = T.get_points(*self.sz)
        self._smooth_weights = np.ones(self.neighborhood.shape[1])      # dispmap_super.m:35
        self.d_min, self.d_step = 0.0, 0.0                              # rescaling only in globalstereo
        self._rd_plan = None                                            # device-resident QPBO, built lazily
        self._ctx = None              # device-resident object state (stereo_fusion_*), built lazily
        self._ctx_has_assignment = False   # the context holds the current assignment
        self._host_stale = False           # ... and the host copy is older than the context's

    @property
    def smooth_weights(self):
        return self._smooth_weights

    @smooth_weights.setter
    def smooth_weights(self, w):
        self._smooth_weights = np.asarray(w, np.float64).reshape(-1)
        if getattr(self, "_ctx", None) is not None:
            self._invalidate_ctx()

    # ---- properties with the reference's setters (dispmap_super.m:39-56)
    @property
    def max_relgap(self):
        return self._max_relgap

    @max_relgap.setter
    def max_relgap(self, v):
        if v < 0:
            raise StereoHipError("Maximum relative gap must be non-negative")
        self._max_relgap = v

    @property
    def improve(self):
        return self._improve

    @improve.setter
    def improve(self, v):
        self._improve = bool(v)

    @property
    def assignment(self):
        if self._host_stale:          # moves ran on the device: fetch the planes on demand
            self._assignment, _ = self._ctx.get_assignment()
            self._host_stale = False
        return self._assignment

    @assignment.setter
    def assignment(self, a):
        self._assignment = np.asfortranarray(a, dtype=np.float64)
        self._host_stale = False
        self._ctx_has_assignment = False
        self.update_energy()

    # ---- device-resident state
    def _attach_unary(self, ctx):
        """Subclasses hand their unary source to the context; False = no device unary."""
        return False

    def _invalidate_ctx(self):
        """A parameter the context has baked in (tol, weights, kernel, unary source) changed."""
        if self._host_stale:
            self._assignment, _ = self._ctx.get_assignment()
            self._host_stale = False
        self._ctx = None
        self._ctx_has_assignment = False

    def _context(self):
        if self._ctx is None:
            ctx = FusionContext(self.sz[0], self.sz[1], self._kernel, self.tol, self.neighborhood,
                                self.smooth_weights, self.d_min, self.d_step)
            if not self._attach_unary(ctx):
                return None
            self._ctx = ctx
            self._ctx_has_assignment = False
        return self._ctx

    @property
    def smoothness_kernel(self):
        return self._kernel

    @smoothness_kernel.setter
    def smoothness_kernel(self, k):
        self._kernel = k
        self._invalidate_ctx()
        self.update_energy()

    def energy(self):
        return self.stored_energy

    # ---- terms
    def unary_cost(self, assignment):
        raise StereoHipError("Overload unary_cost")

    def disparitymap_from_assignment(self, assignment, points=None):
        pts = self.points if points is None else points
        if np.any(assignment[2] == 0):
            raise StereoHipError("Infinite disparity")
        d = -((assignment[0] * pts[0] + assignment[1] * pts[1]) + assignment[3]) / assignment[2]
        if self.d_step != 0:
            d = (d - self.d_min) / self.d_step
        return d

    def all_pairwise_costs(self, assignment, proposal=None):
        return T.pairwise_terms(self._kernel, self.neighborhood, self.points, assignment, proposal,
                                self.smooth_weights, self.tol, self.d_min, self.d_step)

    def update_energy(self):
        """dispmap_super.m:263-274"""
        if self._assignment is None or not hasattr(self, "tol"):
            self.stored_energy = np.inf
            return
        ctx = self._context()
        if ctx is not None:           # unary, E00 and both sums on the device
            self.stored_energy = ctx.set_assignment(self.assignment)
            self._ctx_has_assignment = True
            return
        U = self.unary_cost(self._assignment)
        P = self.all_pairwise_costs(self._assignment)
        self.stored_energy = float(np.sum(U) + np.sum(P))

    # ---- moves
    def binary_fusion(self, proposal):
        """dispmap_super.m:61-84.  `proposal`: 4 x N planes as in the reference, or a PlaneProposal
        (plane table, built into the 4 x N array on the device)."""
        if isinstance(proposal, PlaneProposal):
            ctx = self._context()
            if ctx is None:
                proposal = proposal.expand(self.sz[0] * self.sz[1])
            else:
                if not self._ctx_has_assignment:
                    ctx.set_assignment(self._assignment)
                    self._ctx_has_assignment = True
                self.stored_energy, e, lb, num_unlabelled = ctx.binary_planes(proposal, self._improve)
                self._host_stale = True
                return e, lb, num_unlabelled
        proposal = np.asfortranarray(proposal, dtype=np.float64)
        if proposal.shape != self._assignment.shape:
            raise StereoHipError("Binary fusion: Proposals is of wrong size")
        ctx = self._context()
        if ctx is not None:
            # device-resident move: only the proposal goes up, four scalars come back
            if not self._ctx_has_assignment:
                ctx.set_assignment(self._assignment)
                self._ctx_has_assignment = True
            self.stored_energy, e, lb, num_unlabelled = ctx.binary(proposal, self._improve)
            self._host_stale = True
            return e, lb, num_unlabelled
        E00, E01, E10, E11 = self.all_pairwise_costs(self._assignment, proposal)
        U0 = self.unary_cost(self._assignment)
        U1 = self.unary_cost(proposal)
        if self._rd_plan is None:      # same solver as rd(...), graph layout kept across moves
            self._rd_plan = RdPlan(self.sz[0] * self.sz[1], self.neighborhood, grid=self.sz)
        labelling, e, lb, num_unlabelled = self._rd_plan.solve(U0, U1, E00, E01, E10, E11, self._improve)
        a = self._assignment.copy(order="F")
        take = labelling == 1
        a[:, take] = proposal[:, take]
        self.assignment = a
        return e, lb, num_unlabelled

    def binary_fuse_until_convergence(self, proposal_cell, rng=None, device_loop=True):
        """dispmap_super.m:85-152, including its quirks (the loop variable is bumped inside the
        body, so ids(2) is fused first).  `rng` supplies the random revisit order (MATLAB's
        randi stream cannot be reproduced): a numpy Generator or an explicit id list.
        device_loop: run the schedule natively on the resident state (one call); False keeps the loop
        here, one binary_fusion call per move (the same moves, the same energies)."""
        if not isinstance(proposal_cell, (list, tuple)):
            raise StereoHipError("Input proposals should be given in cell array.")
        n = len(proposal_cell)
        number_of_random_ids = self.maxiter * 5
        if rng is None or isinstance(rng, np.random.Generator):
            rng = rng or np.random.default_rng()
            rand_ids = rng.integers(1, n + 1, number_of_random_ids)
        else:
            rand_ids = np.asarray(rng, dtype=np.int64)
        ids = np.concatenate([np.arange(1, n + 1), rand_ids])
        ids[:-1][np.diff(ids) == 0] = 0         # ids([diff(ids) == 0]) = 0 zeroes the FIRST of a repeated pair
        ids = ids[(ids >= 1) & (ids <= n)]
        # (a subclass with its own binary_fusion keeps the loop below, which calls it)
        native = device_loop and type(self).binary_fusion is dispmap_super.binary_fusion and hasattr(self, "tol")
        single = n > 0 and all(isinstance(p, PlaneProposal) and p.segments is None for p in proposal_cell)
        if native and not single and 32.0 * self.sz[0] * self.sz[1] * n > self.host_stack_budget_bytes:
            native = False   # the native call takes ONE contiguous 4 x N x n host array: beyond the budget keep the
                             # loop here, one proposal uploaded per move (the same moves, the same energies)
        ctx = self._context() if native else None
        if ctx is not None and n > 0:
            # the whole schedule in one native call on the resident state (stereo_fusion_fuse_until_convergence):
            # same loop, same exact energy comparisons, no interpreter and no PCIe traffic between moves
            N = self.sz[0] * self.sz[1]
            if not self._ctx_has_assignment:
                ctx.set_assignment(self._assignment)
                self._ctx_has_assignment = True
            props = None
            if not single:
                props = [p.expand(N) if isinstance(p, PlaneProposal) else np.asfortranarray(p, dtype=np.float64) for p in proposal_cell]
                for p in props:
                    if p.shape != (4, N):
                        raise StereoHipError("Binary fusion: Proposals is of wrong size")
            try:
                if single:
                    E = ctx.fuse_until_convergence(ids, self.maxiter, planes=np.concatenate([p.planes for p in proposal_cell], axis=1),
                                                   improve=self._improve)
                else:
                    E = ctx.fuse_until_convergence(ids, self.maxiter, proposals=props, improve=self._improve)
            except Exception:
                # a schedule that fails half way (a solver bound, out of memory) has already moved the RESIDENT
                # assignment and energy: the host copies are stale from here on, and the stored energy is whatever
                # the context holds now -- the object stays consistent with the device, then the error goes up
                self._host_stale = True
                try:
                    self.stored_energy = float(ctx.get_assignment()[1])
                except Exception:
                    self._ctx_has_assignment = False     # not even that: re-upload the host assignment before the next move
                    self._host_stale = False
                raise
            self.stored_energy = float(E[-1])
            self._host_stale = True
            self.fusion_energies = [float(e) for e in E]
            return len(E)
        E = [self.energy()]
        visited = np.zeros(n, dtype=bool)
        for it in range(1, self.maxiter + 1):
            if it > number_of_random_ids:
                ids = np.concatenate([ids, ids])
            it1 = it + 1                          # iter = iter + 1 inside the for body
            if it1 > len(ids):
                break
            pid = ids[it1 - 1]
            if visited[pid - 1]:
                continue
            self.binary_fusion(proposal_cell[pid - 1])
            E.append(self.energy())
            if E[-2] != E[-1]:
                visited[:] = False
            else:
                visited[pid - 1] = True
            if visited.all():
                break
        self.fusion_energies = [float(e) for e in E]
        return len(E)

    def simultaneous_fusion(self, proposal_cell):
        """dispmap_super.m:153-198"""
        if not isinstance(proposal_cell, (list, tuple)):
            raise StereoHipError("Input proposals should be given in cell array.")
        ctx = self._context()
        N = self.sz[0] * self.sz[1]
        single = len(proposal_cell) > 0 and all(isinstance(p, PlaneProposal) and p.segments is None for p in proposal_cell)
        if not single:
            proposal_cell = [p.expand(N) if isinstance(p, PlaneProposal) else p for p in proposal_cell]
        for p in proposal_cell:
            if np.shape(p) != (4, N) and not isinstance(p, PlaneProposal):
                raise StereoHipError("Simultaneous fusion: Proposals is of wrong size")
        # (an empty cell: the reference appends the current assignment and runs trws with that one
        #  label, dispmap_super.m:158 -- the stateless path below does the same)
        if ctx is not None and len(proposal_cell) > 0:
            # device-resident: proposals go up, unary / positions / TRW-S / scatter stay in HBM
            if not self._ctx_has_assignment:
                ctx.set_assignment(self._assignment)
                self._ctx_has_assignment = True
            if single:   # K planes instead of K x 4 x N doubles
                tab = np.concatenate([p.planes for p in proposal_cell], axis=1)
                self.stored_energy, e, lb, iterations = ctx.simultaneous_planes(tab, self.maxiter, self._max_relgap)
                self._host_stale = True
                return e, lb, iterations
            self.stored_energy, e, lb, iterations = ctx.simultaneous(proposal_cell, self.maxiter, self._max_relgap)
            self._host_stale = True
            return e, lb, iterations
        props = [np.asfortranarray(p, dtype=np.float64) for p in proposal_cell] + [self.assignment]
        unary = np.stack([self.unary_cost(p) for p in props], axis=0)             # K x N
        q, qprim = T.trws_positions(self.neighborhood, self.points, props, self.d_min, self.d_step)
        L, e, lb, iterations = trws(np.int32(self._kernel), unary, self.neighborhood + 1, q, qprim,
                                    self.smooth_weights.reshape(-1), self.tol,
                                    {"maxiter": self.maxiter, "max_relgap": self._max_relgap})
        a = np.zeros_like(props[0])
        for i, p in enumerate(props):
            a[:, L == i + 1] = p[:, L == i + 1]
        self.assignment = a
        return e, lb, iterations

    def current_dispmap(self):
        return self.disparitymap_from_assignment(self.assignment).reshape(self.sz[1], self.sz[0]).T

    def set_disparity(self, disp):
        a = np.zeros((4, self.sz[0] * self.sz[1]))
        a[2] = 1
        a[3] = -np.asarray(disp, np.float64).T.reshape(-1)   # column-major (:)
        self.assignment = a


class dispmap_ncc(dispmap_super):
    """dispmap_ncc.m"""

    def __init__(self, images, disparities, kernel, unary_weight, tol):
        super().__init__(images, kernel)
        self.disparities = np.asarray(disparities, dtype=np.float64).reshape(-1)
        self._unary_weight = unary_weight
        self._tol = tol
        self.ncc = T.ncc_volume(self.images[0], self.images[1], self.disparities, 2)   # compute_ncc(self, 2)
        self.init_solution()

    @property
    def tol(self):
        return self._tol

    @tol.setter
    def tol(self, v):
        if v < 0:
            raise StereoHipError("Tolerance weight must be positive")
        self._tol = v
        self._invalidate_ctx()
        self.update_energy()

    @property
    def unary_weight(self):
        return self._unary_weight

    @unary_weight.setter
    def unary_weight(self, v):
        if v < 0:
            raise StereoHipError("Unary weight must be positive")
        self._unary_weight = v
        self._invalidate_ctx()
        self.update_energy()

    def unary_cost(self, assignment):
        return T.ncc_unary(self.ncc, self.disparities, self._unary_weight, assignment)

    def _attach_unary(self, ctx):
        ctx.unary_ncc(self.ncc, self.disparities, self._unary_weight)
        return True

    def best_disp_from_ncc(self):
        return T.ncc_best_disp(self.ncc, self.disparities)

    def init_solution(self):
        self.set_disparity(self.best_disp_from_ncc())

    restart = init_solution

    # ---- proposals (dispmap_ncc.m:48-92).  Host side: a few dozen points and 3 x 3 SVDs per
    # proposal; MATLAB's svd is outside the reference tree, so this mirrors the arithmetic with
    # numpy's (parity unpinned -- proposals are inputs of the parity-tested path, not outputs).
    def generate_new_plane_RANSAC(self, x, y, r, on_device=False):
        """dispmap_ncc.m:48-66: plane fitted to the winner-takes-all disparities within radius r
        of pixel (x, y) (1-based, x = column), repeated for every pixel.  on_device: the fit runs in
        the device-resident context (stereo_fusion_fit_plane) and a PlaneProposal comes back instead
        of the 4 x N array -- nothing of size N crosses PCIe for the proposal or the move."""
        if on_device:
            ctx = self._context()
            if ctx is not None:
                plane, _ = ctx.fit_plane(x, y, r)
                return PlaneProposal(plane)
        pts = self.points
        best = np.asarray(self.best_disp_from_ncc()).T.reshape(-1)      # column-major pixel order
        ids = np.sqrt((pts[0] - x) ** 2 + (pts[1] - y) ** 2) < r
        p = self.fit_plane_to_points(np.vstack([pts[:, ids], best[ids]]))
        return np.asfortranarray(np.repeat(p.reshape(4, 1), self.sz[0] * self.sz[1], axis=1))

    def generate_plane_lattice(self, radius=5, first=10, step=50, on_device=True):
        """The proposal lattice of example_ncc.m:24-32: `for x = first:step:W, for y = first:step:H`, one
        local plane fit (generate_new_plane_RANSAC) per point, in that order.  on_device: ALL fits in one
        launch (stereo_fusion_fit_planes), PlaneProposals come back; otherwise the host fits, 4 x N arrays."""
        xs = [x for x in range(first, self.sz[1] + 1, step) for _ in range(first, self.sz[0] + 1, step)]
        ys = [y for _ in range(first, self.sz[1] + 1, step) for y in range(first, self.sz[0] + 1, step)]
        if on_device:
            ctx = self._context()
            if ctx is not None and xs:
                planes, _ = ctx.fit_planes(xs, ys, radius)
                return [PlaneProposal(planes[:, i]) for i in range(planes.shape[1])]
        return [self.generate_new_plane_RANSAC(x, y, radius) for x, y in zip(xs, ys)]

    def fit_plane_to_points(self, points):
        """dispmap_ncc.m:67-92: total least squares (kernel 2) or 20 rounds of iteratively
        reweighted least squares (kernel 1) through the SVD of the centred points."""
        points = np.asarray(points, np.float64)
        c = points.mean(axis=1, keepdims=True)
        cost = -(points - c).T                                             # n x 3
        p = np.zeros(4)
        if self._kernel == 1:
            w = np.ones((cost.shape[0], 1))
            for _ in range(20):
                v = np.linalg.svd(w * cost, full_matrices=False)[2][-1]
                p[:3] = v
                w = np.sqrt(np.abs(cost @ v)).reshape(-1, 1)
        elif self._kernel == 2:
            p[:3] = np.linalg.svd(cost, full_matrices=False)[2][-1]
        p[3] = -(p[:3] @ points[:3].mean(axis=1))
        return p / p[2]


class dispmap_globalstereo(dispmap_super):
    """dispmap_globalstereo.m with the constants of ojw_default_options('cvpr08')
    (imrender/ojw/ojw_default_options.m:58-80) as defaults.  As in preprocess() (:377-403) the edge weights come
    from the mean-shift segmentation of the reference image, vgg_segment_ms(Rorig, seg_params) with seg_params =
    [4 5 0] (ojw_default_options.m:68; stereo_amd/segment.py); `segment` (H x W labels) or `smooth_weights`
    replace it for a caller who has them."""

    def __init__(self, images, P, disp_range, disparity_factor, options=None, segment=None,
                 smooth_weights=None, start_disparity=None, rng=None):
        opt = dict(smoothness_kernel=1, disp_thresh=0.02, col_thresh=30.0, lambda_l=9.0, lambda_h=108.0,
                   connect=4, improve=1, seg_params=(4, 5, 0))
        opt.update(options or {})
        super().__init__(images, opt["smoothness_kernel"])
        self.options = opt
        P = np.asarray(P, dtype=np.float64)
        if np.max(np.abs(P[:, :, 0].T.reshape(-1)[[0, 1, 2, 3, 4, 5, 8]] - np.array([1, 0, 0, 0, 1, 0, 1.0]))) > 1e-12:
            raise StereoHipError("First image must be reference image")
        self.P2 = np.asfortranarray(P[:, :, 1].T)          # self.P = permute(P, [2 1 3]); a = 2
        self._P = np.array(P[:, :, :len(self.images)], dtype=np.float64)   # the constructor's 3 x 4 x n (segpln projects with it, :67)
        # MATLAB colon lo*f : hi*f (dispmap_globalstereo.m:48): lo*f + (0 : floor(hi*f - lo*f)), also
        # for a fractional span (np.arange(lo*f, hi*f + 1) would append one element then)
        lo_f, hi_f = disp_range[0] * disparity_factor, disp_range[1] * disparity_factor
        disps = (lo_f + np.arange(np.floor(hi_f - lo_f + 1e-10) + 1))[::-1]
        self.disps = disps                                 # sorted descending (:49)
        self.d_min = float(disps[-1])
        self.d_step = float(disps[0] - self.d_min)
        self._tol = opt["disp_thresh"]
        self._improve = opt["improve"] > 0
        nin = len(self.images)
        if smooth_weights is None and segment is None:
            from . import segment as _segment                                         # :378-392
            sp = opt["seg_params"]
            segment = _segment.vgg_segment_ms(_segment.to_uint8(self.images[0]), sp[0], sp[1], sp[2])
        self.segment = None if segment is None else np.asarray(segment)
        if smooth_weights is not None:
            self.smooth_weights = np.asarray(smooth_weights, np.float64).reshape(-1)
        elif segment is not None:
            seg = np.asarray(segment).T.reshape(-1)
            same = seg[self.neighborhood[0]] == seg[self.neighborhood[1]]
            ew = same * opt["lambda_h"] + (~same) * opt["lambda_l"]                   # :400-402
            self.smooth_weights = ew * (nin / ((opt["connect"] == 8) + 1))            # :403
        if self._kernel == 2:                                                          # :410-413
            self.smooth_weights = self.smooth_weights / self._tol
            self._tol = self._tol ** 2
        H, W = self.sz
        if start_disparity is None:
            rng = rng or np.random.default_rng()
            start_disparity = rng.random((H, W)) * self.d_step + self.d_min          # :56
        self.start_disparity = np.asarray(start_disparity, np.float64)
        self.init_solution()

    @property
    def tol(self):
        return self._tol

    @tol.setter
    def tol(self, v):
        if v < 0:
            raise StereoHipError("Tolerance weight must be positive")
        self._tol = v
        self._invalidate_ctx()
        self.update_energy()

    def unary_cost(self, assignment):
        return T.globalstereo_unary(self.images[0], self.images[1], self.P2, self.d_min, self.d_step,
                                    self.options["col_thresh"], assignment)

    def _attach_unary(self, ctx):
        ctx.unary_globalstereo(self.images[0], self.images[1], self.P2, self.options["col_thresh"])
        return True

    def init_solution(self):
        self.set_disparity(self.start_disparity)

    # ---- proposals (dispmap_globalstereo.m:60-201)
    def segpln_wta(self):
        """The winner-takes-all disparity map of segpln (:72-113), computed on the device once per object."""
        if getattr(self, "_segpln_wta", None) is None:
            self._segpln_wta = T.segpln_wta(self.images, self._P, self.disps, col_thresh=self.options["col_thresh"],
                                            window=self.options.get("window", 2))
        return self._segpln_wta

    def segpln_segments(self):
        """The 14 segmentation maps of segpln (:116-134), H x W x 14, once per object."""
        if getattr(self, "_segpln_maps", None) is None:
            from . import segment as _segment
            self._segpln_maps = _segment.segpln_segments(self.images[0])
        return self._segpln_maps

    def segpln(self, segment_maps=None, seed=0):
        """dispmap_globalstereo.m:60-201: one piecewise-planar proposal (4 x N) per segmentation map -- 14 of them,
        on mean-shift / Felzenszwalb segmentations at the scales `mults` (:122-137; segpln_segments above), or on
        the maps the caller passes (H x W labels 1 .. S, 0 = no segment).  Window matching, LO-RANSAC and the
        plane fits run on the device (stereo_segpln_wta / stereo_segpln_planes); `seed` stands for MATLAB's
        random stream (map b uses seed + b)."""
        if segment_maps is None:
            maps = self.segpln_segments()
            segment_maps = [maps[:, :, b] for b in range(maps.shape[2])]
        wta = self.segpln_wta()
        return [out[0] for out in T.segpln_planes_batch(wta, list(segment_maps), [int(seed) + b for b in range(len(segment_maps))])]

    restart = init_solution
