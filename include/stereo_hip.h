/* stereo_hip.h -- C ABI of libstereo_hip.so (MI355X / gfx950).
 *
 * Drop-in boundary for the 3D-label stereo energy-minimisation path of
 * johannesu/stereo.  Every entry point replaces one reference interface; the
 * citation says which (paths relative to the reference tree).  Plain pointers
 * and sizes only; all functions return 0 on success, non-zero on failure,
 * never throw and never exit().  On failure a message is copied into `err`
 * (if errcap > 0) and is also available from stereo_hip_last_error().
 *
 * Array conventions are MATLAB's at the mex boundary: column-major, i.e. a
 * K x N matrix is N consecutive K-vectors (label fastest).  `conn` is the
 * 2 x E uint32 connectivity, ZERO based as passed to the mex gateways
 * (rd.m:21 / trws.m:33 subtract 1 before the call).
 */
#ifndef STEREO_HIP_H_
#define STEREO_HIP_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* Bumped whenever an entry point is added or a signature changes; the Python binding refuses a
 * library that reports another version (a stale libstereo_hip.so).  3: stereo_hip_device_cus, plan
 * entry points select their plan's device, wall-clock bound on cross-workgroup waits.  4: stereo_fusion_fit_planes,
 * stereo_fusion_fuse_until_convergence.  5: stereo_segpln_wta, stereo_segpln_planes.  6: stereo_segment_* (the segmenters
 * behind dispmap_globalstereo), stereo_trws_plan_debug_terms / _messages, stereo_segpln_planes_batch. */
#define STEREO_HIP_ABI_VERSION 6

/* ---- library ---------------------------------------------------------- */

int stereo_hip_abi_version(void);
/* Number of visible HIP devices (0 if none / runtime failure). */
int stereo_hip_device_count(void);
/* Selects the device used by subsequent calls of this thread (default 0).  A plan stays on the
 * device that was current when it was created: every plan entry point makes that device current for
 * its own duration. */
int stereo_hip_set_device(int device);
/* Compute units of the current device (0 without a device): the number of workgroups of a persistent
 * sweep launch that are certain to be resident together. */
int stereo_hip_device_cus(void);
/* Creates the HIP context of the current device and launches one empty kernel, so that the
 * runtime's one-time initialisation happens now.  Optional; it matters to callers that seed libc
 * rand() for QPBO Improve (QPBO_extra.cpp:13-27 draws its permutation from it): the runtime
 * consumes rand() values during that initialisation.  Returns 0, or non-zero without a device. */
int stereo_hip_warm_up(void);

/* Test hook (host only, no device needed): the node permutation QPBO::Improve walks
 * (QPBO_extra.cpp:13-27, :1224-1232), drawn from libc rand() exactly as the reference draws it --
 * N - 1 values consumed, the generator left where N - 1 calls of rand() leave it.  out: N int32. */
int stereo_hip_improve_permutation(int64_t N, int32_t *out);
/* Last error message of the calling thread ("" if none). */
const char *stereo_hip_last_error(void);

/* ---- TRW-S simultaneous fusion ---------------------------------------- *
 * Replaces cpp/trws_mex.cpp:149-164 mexFunction -> solve_mrf<T> (:27-147):
 *   [labelling, energy, lower_bound, iterations] =
 *       trws_mex(kernel, unary, connectivity-1, q, qprim, alphas, tol, options)
 * kernel      1 = truncated linear  (TypeStereoLinear),
 *             2 = truncated quadratic (TypeStereoQuadratic); anything else fails
 *             with "Unsupported kernel" (trws_mex.cpp:162).
 * unary       K x N   q, qprim  K x E   alphas  E   tol = lambda (trws_mex.cpp:37)
 * maxiter, max_relgap: the two fields of the options struct (trws_mex.cpp:40-41;
 *             defaults 1000 and 0 are applied by the caller / gateway).
 * labelling   N doubles, ONE based (trws_mex.cpp:137).
 */
/* Repeated calls with the same (kernel, K, connectivity) -- one per simultaneous_fusion of an object,
 * dispmap_super.m:153-198 -- reuse the plan of the last problem seen: graph analysis, descriptors and
 * device buffers are kept, a call costs the upload of its inputs, the iterations and the labels back.
 * If every column of q and qprim is bitwise the same vector (fronto-parallel proposals, :177-183) only
 * that vector is uploaded and the shared-position kernels run; same bits as the K x E form.
 * STEREO_HIP_TRWS_CACHE=0: a fresh plan per call and the K x E arrays always.  The cached plan keeps its
 * device memory (messages: 8 K E bytes) until the next different problem or stereo_trws_cache_clear(). */
int stereo_trws(int kernel, const double *unary, const uint32_t *conn, const double *q,
                const double *qprim, const double *alphas, double tol, double maxiter,
                double max_relgap, int K, int64_t N, int64_t E, double *labelling,
                double *energy, double *lower_bound, double *iterations, char *err,
                size_t errcap);
void stereo_trws_cache_clear(void);


/* Device-resident form of the same solver: graph analysis (SetAutomaticOrdering,
 * ordering.cpp:7-157; CompleteGraphConstruction, MRFEnergy.cpp:137-229) is done
 * once per connectivity, inputs live in HBM across calls, iterations can be
 * issued without host round trips.  Used by the mex gateway for repeated
 * simultaneous_fusion calls on one image (dispmap_super.m:153-198) and by
 * bench.py (inputs resident before the timed region).
 */
typedef struct stereo_trws_plan stereo_trws_plan;

/* flags */
#define STEREO_TRWS_MESSAGES_EXACT 0   /* reference lower-envelope semantics (default) */
#define STEREO_TRWS_MESSAGES_MINPLUS 1 /* plain min-plus messages, no certificate / envelope construction:
                                         the reference's bits unless a message's certificate would have
                                         failed (exact or near ties), the brute-force O(K^2) result always.
                                         With 64 < K <= 256 and shared ascending positions it is a branch of
                                         trws_wide_kernel (1.4x the exact messages at 3000x2000x256) */
/* OR-ed into message_mode: visit the nodes in index order -- MRFEnergy's order when the gateway
 * does NOT call SetAutomaticOrdering (trws_mex.cpp:121; nodes keep the order of AddNode,
 * MRFEnergy.cpp:37-76).  On an image grid the dependency DAG then has H + W - 1 anti-diagonal
 * levels and no serial border chain (the gateway's order: ~2 (H + W)): a faster, equally valid
 * TRW-S schedule whose labels / energies are NOT the gateway's.  Explicit opt-in only. */
#define STEREO_TRWS_ORDER_INDEX 0x100

int stereo_trws_plan_create(int kernel, int K, int64_t N, int64_t E, const uint32_t *conn,
                            int message_mode, stereo_trws_plan **plan, char *err,
                            size_t errcap);
void stereo_trws_plan_destroy(stereo_trws_plan *plan);

/* Host -> HBM upload of the per-call inputs (same meaning as stereo_trws).
 * q == NULL && qprim == NULL selects "shared positions": every edge uses the
 * K-vector `positions` for both q(:,e) and qprim(:,e) (fronto-parallel labels,
 * dispmap_super.m:177-183 with planes [0 0 1 -d]); nothing K x E is stored. */
int stereo_trws_plan_upload(stereo_trws_plan *plan, const double *unary, const double *q,
                            const double *qprim, const double *positions,
                            const double *alphas, double tol, char *err, size_t errcap);
/* Same, but the arrays are already DEVICE pointers (e.g. torch tensors, or the
 * output of the cost-volume kernels); they are borrowed, not copied. */
int stereo_trws_plan_bind_device(stereo_trws_plan *plan, const double *d_unary,
                                 const double *d_q, const double *d_qprim,
                                 const double *d_positions, const double *d_alphas,
                                 double tol, char *err, size_t errcap);
/* Both calls start a NEW minimisation: if the plan has iterated before, they imply
 * stereo_trws_plan_reset (an iterate call runs the forward sweep of the following iteration ahead
 * of time, so messages computed from the old inputs cannot be continued with new ones). */

/* Zero all messages (MRFEnergy.cpp:115-133) and the iteration counter. */
int stereo_trws_plan_reset(stereo_trws_plan *plan, char *err, size_t errcap);
/* Runs up to `iters` further iterations of Minimize_TRW_S (minimize.cpp:31-113):
 * forward sweep, backward sweep (+ lower bound), primal labelling + energy, stop
 * test `(E-LB)/E < max_relgap`.  Returns in *done_iters the number executed in
 * this call, *stopped != 0 if the relative-gap test fired.  `stream` is a
 * hipStream_t (NULL = default stream); the call returns after the last
 * iteration's scalars have reached the host. */
int stereo_trws_plan_iterate(stereo_trws_plan *plan, int iters, double max_relgap,
                             void *stream, int *done_iters, int *stopped, char *err,
                             size_t errcap);
/* Result of the most recent iteration (minimize.cpp:104: the LAST primal, not the
 * best).  labelling: N doubles, one based; may be NULL. */
int stereo_trws_plan_result(stereo_trws_plan *plan, double *labelling, double *energy,
                            double *lower_bound, double *iterations, char *err,
                            size_t errcap);
/* Diagnostics: graph analysis results (all may be NULL). rank: N int64;
 * levels: number of dependency levels of the node order. */
int stereo_trws_plan_info(stereo_trws_plan *plan, int64_t *rank, int64_t *levels,
                          int64_t *max_level_nodes, char *err, size_t errcap);
/* Timing hooks for bench.py: total device time (ms, hipEvent on the plan's
 * stream) and launch count of the sweep kernels since the last reset_stats. */
int stereo_trws_plan_stats(stereo_trws_plan *plan, double *sweep_ms, int64_t *sweep_launches,
                           int reset);

/* Diagnostics: number of message updates (since the last reset) for which the fast
 * min-plus path could not certify equality with the reference's serial envelope
 * construction and the serial construction was run instead. */
int stereo_trws_plan_counters(stereo_trws_plan *plan, int64_t *serial_messages, int reset);

/* The speculative schedule of the long serial run (the image grid's border chain): a runner computes the
 * run's recurrence ahead with plain min-plus, segments of the run recompute every visit with the certified routine side
 * by side, compare what they started from with what the segment in front produced, and walk again if it differs --
 * results are the sequential sweep's either way.  out[0] = 1 if the plan's next sweeps use it (graph, kernel family,
 * positions permitting; STEREO_HIP_TRWS_SPEC=0 turns it off at plan creation), out[1] = segments walked twice,
 * out[2] = segments committed, out[3] = visits of the runner, since the plan was created.  No reference counterpart. */
int stereo_trws_plan_spec_stats(stereo_trws_plan *plan, int64_t out[4]);

/* Diagnostics / test hook: M independent message updates Edge::UpdateMessage
 * (typeStereoLinear.h:329-487, typeStereoQuadratic.h:329-501) on the device, one wavefront per
 * message, through the very routine the pipelined sweep kernel uses (certified min-plus fast
 * path, second look, serial lower-envelope construction).  Row m of the M x K arrays (K <= 64):
 * H = gamma[m] * Di - msg_in (:383-387), source positions q_source, destination positions q_dest
 * (the caller has resolved dir == m_dir, :343-357).  certificate = 0 forces the serial
 * construction.  shared_positions != NULL tells the routine that every row uses these K strictly
 * ascending positions for sources and destinations (`window` = index distance beyond which a
 * source costs >= vTrunc, or -1).  used_serial[m] (may be NULL) = 1 if the serial construction ran. */
int stereo_trws_messages(int kernel, int K, int64_t M, const double *Di, const double *gamma,
                         const double *msg_in, const double *q_source, const double *q_dest,
                         const double *alpha, double lambda, int certificate, int window,
                         const double *shared_positions, double *msg_out, double *vmin,
                         int32_t *used_serial, char *err, size_t errcap);

/* Diagnostics: which sweep implementation the plan's current inputs select.
 * 1 generic persistent kernel, 2 pipelined kernel (K <= 64),
 * 3 wide pipelined kernel (64 < K <= 256, shared strictly ascending positions),
 * 4 two-labels-per-lane pipelined kernel (64 < K <= 128, linear kernel, any positions).
 * All give identical results.  Negative on a NULL plan. */
int stereo_trws_plan_path(stereo_trws_plan *plan);

/* ---- row strips: one image tiled across the GPUs of a node ------------------------------ *
 * No reference counterpart (the reference is one serial sweep, minimize.cpp:36-95); what is
 * sharded is exactly that sweep under the order of ordering.cpp:42-152, and the labels are the
 * ones a single plan returns, bit for bit.  `owner[i]` names the strip that visits node i; strips
 * form a chain (an edge joins nodes of one strip or of strips g, g+1) -- for an image, bands of
 * rows.  One plan per strip, normally one process and one GPU each.  A strip stores only what
 * it touches -- its own nodes, the nodes one edge away (halo) and the edges with an own endpoint,
 * under strip-local ids -- so the messages and unaries of a large problem are divided among the
 * GPUs (1/G each plus a boundary row).  A boundary node's new messages, its completion flag and
 * its label are written by the visiting workgroup straight into the neighbouring strip's arrays,
 * at the neighbour's ids (peer stores over xGMI + flag; nothing is staged through the host and
 * there is no collective on the data path).  stereo_trws_plan_upload / _bind_device still take
 * the arrays of the WHOLE problem and keep the strip's rows (a gather into arrays owned by the
 * plan, so the full ones may be freed afterwards); stereo_trws_plan_bind_device_strip takes
 * arrays that are strip-local already, in the order stereo_trws_plan_strip_layout reports.
 * stereo_trws_plan_result fills the labels of the strip's nodes (1 elsewhere).  Energy and lower bound come back as per-strip
 * partial sums (each in the reference's summation order restricted to the strip); the caller adds
 * them in strip order -- the only cross-strip reduction, two doubles per iteration -- so they
 * agree with the single-plan values to rounding (1e-12 relative), not bit for bit.
 *
 * max_workgroups > 0 caps the strip's launch (strips that share one GPU must all be resident
 * together).  share_analysis_with: another strip's plan of the same problem in this process
 * (owner may then be NULL), so the host-side analysis is done once. */
int stereo_trws_plan_create_strip(int kernel, int K, int64_t N, int64_t E, const uint32_t *conn,
                                  int message_mode, const int32_t *owner, int nstrips, int strip,
                                  int max_workgroups, stereo_trws_plan *share_analysis_with,
                                  stereo_trws_plan **plan, char *err, size_t errcap);
/* Neighbour in the same process (which: 0 = strip - 1, 1 = strip + 1); enables peer access if the
 * two plans live on different devices. */
int stereo_trws_plan_connect(stereo_trws_plan *plan, int which, stereo_trws_plan *peer, char *err,
                             size_t errcap);
/* Neighbour in another process: export writes STEREO_TRWS_IPC_BYTES bytes of HIP IPC handles
 * (messages, flags, labels) that the neighbour passes to ipc_connect after exchanging them over
 * any channel (bench.py: torch.distributed all_gather). */
#define STEREO_TRWS_IPC_BYTES 256
int stereo_trws_plan_ipc_export(stereo_trws_plan *plan, void *handles, size_t cap, char *err,
                                size_t errcap);
int stereo_trws_plan_ipc_connect(stereo_trws_plan *plan, int which, const void *handles, char *err,
                                 size_t errcap);
/* One iteration of a strip in three steps, so that all strips of a process can be in flight
 * together and the caller can reduce the partial sums across processes:
 *   issue   launches forward sweep (first iteration only), backward sweep and the primal pass
 *           fused with the next forward sweep, without waiting (stream NULL = the plan's own);
 *   collect waits and returns this strip's partial lower bound and energy;
 *   commit  records the reduced totals (what stereo_trws_plan_result reports) and counts the
 *           iteration.  The stop test (E-LB)/E < max_relgap of minimize.cpp:105 is the caller's. */
int stereo_trws_plan_issue(stereo_trws_plan *plan, void *stream, char *err, size_t errcap);
/* The same for n <= 16 strips that share ONE device (logical strips; a process that owns several
 * bands): each sweep is a single launch whose workgroups are divided among the strips, so all of
 * them are resident together (separate launches on separate streams may be serialised by the
 * runtime, and strips wait for each other in both directions).  Collect / commit per plan. */
int stereo_trws_plans_issue(stereo_trws_plan *const *plans, int n, void *stream, char *err, size_t errcap);
int stereo_trws_plan_collect(stereo_trws_plan *plan, double *lower_bound_part, double *energy_part,
                             char *err, size_t errcap);
int stereo_trws_plan_commit(stereo_trws_plan *plan, double lower_bound, double energy, char *err,
                            size_t errcap);
/* What a strip stores.  nodes (n_nodes entries): global id of local node i -- the n_own own nodes
 * in ascending order, then the halo; edges (n_edges): global id of local edge e, ascending.  Any
 * output may be NULL (ask for the counts first).  A plan of the whole problem reports identity. */
int stereo_trws_plan_strip_layout(stereo_trws_plan *plan, int64_t *n_nodes, int64_t *n_own,
                                  int64_t *n_edges, int32_t *nodes, int32_t *edges);
/* stereo_trws_plan_bind_device with strip-local arrays: d_unary K x n_nodes, d_alphas n_edges,
 * d_q / d_qprim K x n_edges (or shared d_positions), rows in the order of strip_layout.  The
 * arrays stay the caller's. */
int stereo_trws_plan_bind_device_strip(stereo_trws_plan *plan, const double *d_unary,
                                       const double *d_q, const double *d_qprim,
                                       const double *d_positions, const double *d_alphas,
                                       double tol, char *err, size_t errcap);
/* The same without a device: what strip `strip` of the problem stores, and the descriptors of its
 * own visits in sweep `direction` (0 forward, 1 backward) after renumbering -- n_visits x 64 words,
 * layout in stereo_amd/csrc/trws_graph.h (words 0 node, 4-11 edges, 20-23 dependencies, 32-39
 * neighbours, 43 remote bits, 45-52 / 53-54 the neighbour strips' ids).  Sizes first (arrays
 * NULL), then the arrays.  For tests of the multi-GPU decomposition on a host without GPUs. */
int stereo_trws_strip_layout_host(int64_t N, int64_t E, const uint32_t *conn, const int32_t *owner,
                                  int nstrips, int strip, int direction, int64_t *n_nodes,
                                  int64_t *n_own, int64_t *n_edges, int64_t *n_visits, int32_t *nodes,
                                  int32_t *edges, int32_t *desc, char *err, size_t errcap);
/* Diagnostics (any output may be NULL). */
int stereo_trws_plan_strip_info(stereo_trws_plan *plan, int *nstrips, int *strip, int64_t *own_nodes,
                                int64_t *runs_forward, int64_t *runs_backward, int *needs_previous,
                                int *needs_next);
/* Development aid: completion flags (N, by rank: epoch of the last completed visit) and
 * [ticket, abort] of the plan's last launch. */
int stereo_trws_plan_debug_flags(stereo_trws_plan *plan, int32_t *done, int32_t *ctl);
/* Development aids: the lower-bound terms of the plan's last backward sweep in the order the host sums them (rank N - 1
 * down to 0: the node's own term, minimize.cpp:79-83, then one per message it sent, :85-91), and the message rows as
 * they lie in HBM (E x K doubles, edge-major).  No reference counterpart. */
int stereo_trws_plan_debug_terms(stereo_trws_plan *plan, double *lb_terms, int64_t cap, int64_t *n_lb);
int stereo_trws_plan_debug_messages(stereo_trws_plan *plan, double *out, int64_t count);
/* The gateway on several devices.  With STEREO_HIP_GPUS=G (2 .. 16) in the environment stereo_trws -- what trws_mex
 * reaches, cpp/trws_mex.cpp:149-164 -- cuts the problem into G row strips when the graph is the image grid of
 * dispmap_super.m:279-302 (nodes col * H + row, 4-neighbourhood, H >= 2 G; K <= 128, or <= 256 with one positions vector
 * in every column of q and qprim): strip g on device g when the process sees G devices (peer access over xGMI), all
 * strips on the current device otherwise (logical strips, one fused launch per sweep).  Same labels, energy, bound and
 * iteration count as on one device.  This call tells how many strips the calling thread's last stereo_trws ran on
 * (1: the single-device plan).  No reference counterpart. */
int stereo_trws_gateway_strips(void);

/* Host only (no device): the speculative schedule of the graph's one long serial run, if it has one (DESIGN.md 4.5;
 * stereo_trws_plan_spec_stats).  info[0..5] = exists, the run of stereo_trws_schedule that is cut, its first schedule
 * position, one past its last, visits per segment, segments.  The arrays (may be NULL) describe the schedule with that
 * run replaced by its segments: run_ptr (*nruns + 1 schedule positions), kind (*nruns: 0, or 1 + segment index),
 * ticket_run (*nruns + 1 tickets; -1 = the runner's ticket, the first).  No reference counterpart. */
int stereo_trws_spec_schedule(int64_t N, int64_t E, const uint32_t *connectivity0, int direction, int64_t *info,
                              int64_t *nruns, int64_t *run_ptr, int64_t *kind, int64_t *ticket_run, char *err,
                              size_t errcap);

/* stereo_trws_schedule with strips: additionally run_strip (N provided, *nruns used) = strip of
 * every run, remote (N, by rank) = descriptor word 43 (bits 0-7: outgoing message k is written
 * to a neighbour, 8-15: which one, 16 / 17: flag and label also raised at strip - 1 / + 1). */
int stereo_trws_schedule_strips(int64_t N, int64_t E, const uint32_t *conn, int64_t max_resident_runs,
                                int direction, const int32_t *owner, int nstrips, int64_t *rank_at,
                                int64_t *run_ptr, int64_t *nruns, int64_t *ticket_run,
                                int64_t *pred_rank, int64_t *dep_ptr, int64_t *dep_rank,
                                int64_t *run_strip, int64_t *remote, char *err, size_t errcap);

/* Host-only graph analysis behind stereo_trws_plan_create (no device needed):
 * node order of SetAutomaticOrdering (ordering.cpp:7-157), edge orientation and
 * per-node forward/backward edge lists of CompleteGraphConstruction
 * (MRFEnergy.cpp:137-229), and the dependency level of every node.
 * rank: N; tail, head: E (after orientation); mdir: E (Swap parity);
 * fwd_ptr, bwd_ptr: N+1 CSR offsets indexed by NODE ID; fwd_idx, bwd_idx: E
 * edge ids in list order; level: N (indexed by node id).  Any output may be NULL. */
int stereo_trws_analyze(int64_t N, int64_t E, const uint32_t *conn, int64_t *rank,
                        int64_t *tail, int64_t *head, int32_t *mdir, int64_t *fwd_ptr,
                        int64_t *fwd_idx, int64_t *bwd_ptr, int64_t *bwd_idx, int64_t *level,
                        char *err, size_t errcap);

/* Host-only inspection of the dataflow schedule the descriptor-driven sweep kernels walk
 * (trws_graph.h "chain schedule"; no reference counterpart -- the reference is serial,
 * minimize.cpp:36-95).  direction 0 = forward sweep, 1 = backward.  max_resident_runs as
 * passed by stereo_trws_plan_create (0 = unlimited).  Outputs (any may be NULL):
 * rank_at (N): rank visited at schedule position p; run_ptr (N+1 entries provided, *nruns+1
 * used): offsets of the runs in schedule positions; ticket_run (N provided, *nruns used):
 * run picked up with ticket t; pred_rank (N, by rank): rank visited just before in the same
 * run whose messages are handed over in LDS, or -1; dep_ptr (N+1) / dep_rank (4N provided):
 * foreign ranks a node waits for (completion flags).  Returns non-zero if the graph is not
 * eligible for those kernels. */
int stereo_trws_schedule(int64_t N, int64_t E, const uint32_t *conn, int64_t max_resident_runs,
                         int direction, int64_t *rank_at, int64_t *run_ptr, int64_t *nruns,
                         int64_t *ticket_run, int64_t *pred_rank, int64_t *dep_ptr,
                         int64_t *dep_rank, char *err, size_t errcap);

/* ---- QPBO roof-duality binary fusion ---------------------------------- *
 * Replaces cpp/rd_mex.cpp:14-100 mexFunction:
 *   [labelling, energy, lower_bound, num_unlabelled] =
 *       rd_mex(U0, U1, E00, E01, E10, E11, connectivity-1, options)
 * U0,U1 N; E00..E11 E; improve = options.improve (rd_mex.cpp:34, default false).
 * labelling N doubles in {-1,0,1}; num_unlabelled is counted BEFORE Improve
 * (rd_mex.cpp:83-88).
 */
/* Repeated calls with the same connectivity (a fusion schedule: one call per proposal) reuse the plan of
 * the last connectivity seen -- from the second call on a move costs the upload of the six term arrays
 * and the solve (3.7 ms at 450 x 375 instead of 56 ms; STEREO_HIP_RD_CACHE=0 rebuilds per call). */
int stereo_rd(const double *U0, const double *U1, const double *E00, const double *E01,
              const double *E10, const double *E11, const uint32_t *conn, int64_t N,
              int64_t E, int improve, double *labelling, double *energy,
              double *lower_bound, double *num_unlabelled, char *err, size_t errcap);
/* Frees the plan stereo_rd keeps for the last connectivity (its device buffers stay allocated until
 * the next different problem otherwise). */
void stereo_rd_cache_clear(void);

/* Device-resident form of stereo_rd for repeated fusion moves on one connectivity
 * (dispmap_super.m:61-84 is called once per proposal): edge grouping, the doubled-graph
 * slot layout and all device buffers are set up once; a move uploads (or binds) the six term
 * arrays, rebuilds capacities on the device and solves. */
typedef struct stereo_rd_plan stereo_rd_plan;
int stereo_rd_plan_create(int64_t N, int64_t E, const uint32_t *conn, stereo_rd_plan **plan, char *err,
                          size_t errcap);
void stereo_rd_plan_destroy(stereo_rd_plan *plan);
/* Device pointer to the labels of the last solve: N int8 in {-1, 0, 1} (valid until the next solve). */
const int8_t *stereo_rd_plan_device_labels(stereo_rd_plan *plan);
/* Optional hint: the N nodes are the pixels of an H x W image numbered col*H + row
 * (dispmap_super.m:281-282).  Only the internal work partition changes (image patches instead of
 * index ranges per workgroup: fewer grid-wide barriers), never a result. */
int stereo_rd_plan_set_grid(stereo_rd_plan *plan, int H, int W, char *err, size_t errcap);
int stereo_rd_plan_solve(stereo_rd_plan *plan, const double *U0, const double *U1, const double *E00,
                         const double *E01, const double *E10, const double *E11, int improve,
                         double *labelling, double *energy, double *lower_bound,
                         double *num_unlabelled, char *err, size_t errcap);
/* the same with the six term arrays already in HBM (e.g. written by the term-builder kernels);
 * labelling may be NULL: the labels then stay on the device (stereo_rd_plan_device_labels) */
int stereo_rd_plan_solve_device(stereo_rd_plan *plan, const double *d_U0, const double *d_U1,
                                const double *d_E00, const double *d_E01, const double *d_E10,
                                const double *d_E11, int improve, double *labelling, double *energy,
                                double *lower_bound, double *num_unlabelled, char *err, size_t errcap);

/* ---- term builders and cost volume ------------------------------------- *
 * Device versions of the MATLAB array math between the images and the two
 * solvers.  Pixels are numbered column-major, id = col*H + row (dispmap_super.m:281-282);
 * points(:,id) = [col+1; row+1] (:275-278); planes are 4-vectors [a b c d],
 * disparity = -(a*x + b*y + d)/c (:318-328).  d_step != 0 applies the
 * dispmap_globalstereo rescaling (d - d_min)/d_step (dispmap_globalstereo.m:336-345).
 */

/* dispmap_super.m:236-262 all_pairwise_costs (+ :226-235 pairwise_cost).
 * conn 2 x E zero based [ind1; ind2]; points 2 x N; assignment / proposal 4 x N;
 * proposal == NULL computes E00 only (update_energy, :263-274).  Planes with c == 0
 * fail with "Infinite disparity" (:323-325). */
int stereo_pairwise_terms(int kernel, int64_t N, int64_t E, const uint32_t *conn, const double *points,
                          const double *assignment, const double *proposal, const double *weights,
                          double tol, double d_min, double d_step, double *E00, double *E01,
                          double *E10, double *E11, char *err, size_t errcap);

/* dispmap_super.m:177-183: q(k,e), qprim(k,e) of K proposals (4 x N x K) -> K x E each. */
int stereo_trws_positions(int64_t N, int64_t E, int K, const uint32_t *conn, const double *points,
                          const double *proposals, double d_min, double d_step, double *q,
                          double *qprim, char *err, size_t errcap);

/* dispmap_ncc.m:116-198 compute_ncc.  im0 = reference image, im1 = the image that is
 * shifted; both H x W x 3 column major doubles.  layout 0: H x W x D (MATLAB, self.ncc);
 * layout 1: D x (H*W), label fastest -- what TRW-S consumes as `unary` after
 * unary_weight*(1 - ncc). */
int stereo_ncc_volume(const double *im0, const double *im1, int H, int W, const double *disparities,
                      int D, int patchsize, int layout, double *ncc, char *err, size_t errcap);

/* dispmap_ncc.m:107-115 unary_cost = unary_weight*(1 - sample_ncc_from_disp(...)) (:222-276). */
int stereo_ncc_unary(const double *ncc, int H, int W, int D, int layout, const double *disparities,
                     double unary_weight, const double *assignment, double *U, char *err,
                     size_t errcap);

/* dispmap_ncc.m:208-221 best_disp_from_ncc (winner takes all + parabola vertex), H x W. */
int stereo_ncc_best_disp(const double *ncc, int H, int W, int D, int layout, const double *disparities,
                         double *best, char *err, size_t errcap);

/* dispmap_globalstereo.m:355-375 unary_cost with ephoto (:405) and the 'linear' branch of
 * vgg_interp2 (imrender/vgg/vgg_interp2.cxx:245-322, oobv = -1000).  P2 = self.P(:,:,2),
 * 4 x 3 column major (already permuted, :43). */
int stereo_globalstereo_unary(const double *im0, const double *im1, int H, int W, int C,
                              const double *P2, double d_min, double d_step, double col_thresh,
                              const double *assignment, double *U, char *err, size_t errcap);

/* ---- device-resident fusion moves --------------------------------------- *
 * The state of one dispmap object kept in HBM across moves: connectivity, weights, points,
 * the current assignment (dispmap_super.m properties :5-24), its unary and the unary source of
 * the subclass.  A binary fusion move (dispmap_super.m:61-84 binary_fusion followed by
 * :263-274 update_energy) then uploads only the proposal (4 x N) and returns four scalars:
 * pairwise terms, both unaries, QPBO (stereo_rd_plan), the scatter of the accepted planes and
 * the new energy all run on the device.  d_step != 0: dispmap_globalstereo rescaling. */
typedef struct stereo_fusion stereo_fusion;
int stereo_fusion_create(int H, int W, int kernel, double tol, int64_t E, const uint32_t *conn,
                         const double *weights, double d_min, double d_step, stereo_fusion **ctx,
                         char *err, size_t errcap);
void stereo_fusion_destroy(stereo_fusion *ctx);
/* unary source: dispmap_ncc.m:107-115 (ncc: H x W x D, MATLAB layout) ... */
int stereo_fusion_unary_ncc(stereo_fusion *ctx, const double *ncc, int D, const double *disparities,
                            double unary_weight, char *err, size_t errcap);
/* ... or dispmap_globalstereo.m:355-375 (images H x W x C, P2 4 x 3 as in stereo_globalstereo_unary) */
int stereo_fusion_unary_globalstereo(stereo_fusion *ctx, const double *im0, const double *im1, int C,
                                     const double *P2, double col_thresh, char *err, size_t errcap);
/* set.assignment + update_energy (dispmap_super.m:57-60, :263-274); energy may be NULL */
int stereo_fusion_set_assignment(stereo_fusion *ctx, const double *assignment, double *energy, char *err,
                                 size_t errcap);
/* current assignment (4 x N, may be NULL) and stored energy (may be NULL) */
int stereo_fusion_get_assignment(stereo_fusion *ctx, double *assignment, double *energy, char *err,
                                 size_t errcap);
/* one binary fusion move; energy = stored energy after the move, the other three are the outputs
 * of rd.m for this move (any may be NULL) */
int stereo_fusion_binary(stereo_fusion *ctx, const double *proposal, int improve, double *energy,
                         double *rd_energy, double *lower_bound, double *num_unlabelled, char *err,
                         size_t errcap);
/* Proposals built on the device.  The proposals of the reference's examples are one
 * plane for every pixel (example_ncc.m:24-41: plane fits and fronto-parallel planes, repmat'ed at
 * dispmap_ncc.m:65) or one plane per image segment (dispmap_globalstereo.m:154-192): `planes` is
 * 4 x S, `segments` N ids in [0, S) (may be NULL when S == 1, or to reuse the ids of the last call).
 * Same move as stereo_fusion_binary; 32 S bytes cross PCIe instead of 32 N. */
int stereo_fusion_binary_planes(stereo_fusion *ctx, const double *planes, int S, const int32_t *segments,
                                int improve, double *energy, double *rd_energy, double *lower_bound,
                                double *num_unlabelled, char *err, size_t errcap);
/* dispmap_ncc.m:48-92 generate_new_plane_RANSAC / fit_plane_to_points on the device: plane [a b 1 d]
 * through the winner-takes-all disparities (:208-221) within radius r of pixel (x, y) (one based,
 * x = column); kernel 1: 20 rounds of IRLS, kernel 2: total least squares.  The reference's
 * svd(...) V(:, end) is computed as the smallest eigenvector of the 3 x 3 normal matrix (MATLAB's svd
 * lies outside the reference tree: agreement to rounding, not bit for bit).  npoints (may be NULL):
 * pixels inside the radius. */
int stereo_fusion_fit_plane(stereo_fusion *ctx, double x, double y, double r, double *plane,
                            double *npoints, char *err, size_t errcap);
/* binary_fuse_until_convergence (dispmap_super.m:85-152) in one call on the resident state: n
 * proposals (either `proposals`, 4 x N x n column major, or `planes`, 4 x n single planes built on the
 * device; the other NULL), the 1-based revisit schedule `ids` exactly as the reference's loop indexes
 * it (ids(iter + 1) in trip iter; the caller concatenates 1:n with its random draws and applies the
 * reference's diff filter), at most `maxiter` trips.  n_energies = length(E), what the reference
 * returns; energies (cap entries, may be NULL) = E itself. */
int stereo_fusion_fuse_until_convergence(stereo_fusion *ctx, const double *proposals, const double *planes, int n,
                                         const int64_t *ids, int64_t n_ids, int64_t maxiter, int improve,
                                         double *n_energies, double *energies, int64_t cap, char *err, size_t errcap);
/* The same fit for n centres in ONE launch (xs, ys: 1-based pixel coordinates; planes: 4 x n column
 * major; npoints: n or NULL) -- the lattice `for x = 10:50:W, for y = 10:50:H` of example_ncc.m:24-32. */
int stereo_fusion_fit_planes(stereo_fusion *ctx, const double *xs, const double *ys, int n, double r,
                             double *planes, double *npoints, char *err, size_t errcap);
/* stereo_fusion_simultaneous with K single-plane proposals (planes 4 x K) built on the device */
int stereo_fusion_simultaneous_planes(stereo_fusion *ctx, const double *planes, int K, double maxiter,
                                      double max_relgap, double *energy, double *trws_energy,
                                      double *lower_bound, double *iterations, char *err, size_t errcap);
/* one simultaneous fusion (dispmap_super.m:153-198): the K proposals (4 x N x K) and the current
 * assignment (appended as label K+1, :160) compete per pixel; unary K x N, q / qprim K x E, TRW-S
 * (stereo_trws_plan, options maxiter / max_relgap as in trws.m) and the scatter of the winning
 * planes run on the device.  energy = stored energy afterwards; the other three are trws.m's. */
int stereo_fusion_simultaneous(stereo_fusion *ctx, const double *proposals, int K, double maxiter,
                               double max_relgap, double *energy, double *trws_energy,
                               double *lower_bound, double *iterations, char *err, size_t errcap);

/* ---- SegPln proposals (dispmap_globalstereo.m:60-201) -------------------------------------
 * The winner-takes-all disparity map by window matching (:72-113): for every image and every disparity of
 * `disps` (the class's self.disps: descending, :49) the reference image's pixels are projected, the image is
 * sampled bilinearly (vgg_interp2, oobv -1000), ephoto of the colour difference (:405) is averaged over the
 * (2 window + 1)^2 box ('valid') and summed over the images; normalised score, first maximum, disparities with
 * a score below min_corr (0.07, :112) set to 0, mirrored back to H x W.  images: n_images blocks of H x W x C
 * doubles (MATLAB layout), the first one the reference image; P: 3 x 4 x n_images column major (the argument
 * of the class constructor, :67).  wta: H x W column major. */
int stereo_segpln_wta(const double *images, int n_images, int H, int W, int C, const double *P, const double *disps,
                      int nd, double col_thresh, int window, double min_corr, double *wta, char *err, size_t errcap);
/* One proposal of segpln (:140-197) from a segmentation the CALLER supplies (labels 1 .. S, 0 = none; the
 * mean-shift / Felzenszwalb segmenters of :124-137 are out of scope): per segment the LO-RANSAC of :417-450
 * (threshold rt = 0.1 at :163, at most max_samples = 500 trials, confidence 0.95) over the world coordinates
 * [x y 1] / d, then the least-squares plane of the inliers; proposal (4 x N) = [N1 N2 1 N3] on the segment's
 * pixels, [0 0 1 0] where nothing was fitted, NaN / Inf -> 1e-100.  The reference draws its triples with
 * randperm; here trial t of segment s takes the three distinct indices splitmix64(seed, s, t, attempt) mod n
 * (oracle/terms.py:segpln_sample -- the draw is an input of the parity test), the 3 x 3 systems are solved by
 * Cramer's rule and the least-squares planes through the normal equations in a fixed summation order (MATLAB's
 * mldivide is outside the reference tree; oracle/terms.py:segpln_planes is the definition, matched bit for
 * bit).  planes (3 x S) / inliers (S) may be NULL; so may proposal when planes is given (the caller expands the planes
 * over the segments itself, or hands planes + segments to stereo_fusion_binary_planes). */
int stereo_segpln_planes(const double *wta, const int32_t *segments, int H, int W, double rt, uint64_t seed,
                         int max_samples, double *proposal, int S, double *planes, int32_t *inliers, char *err,
                         size_t errcap);

/* The same for M maps in one call (segpln fits fourteen, dispmap_globalstereo.m:122-197): segments[m], seeds[m], S[m],
 * proposals[m] (or NULL), planes[m] (or NULL), inliers[m] (or NULL) are map m's arguments of stereo_segpln_planes and
 * the results are those of M such calls, bit for bit -- the segments of all maps run side by side on the device (one
 * launch per workgroup size; pixels grouped and proposals delivered on helper threads), which is what the call is for. */
int stereo_segpln_planes_batch(const double *wta, const int32_t *const *segments, int M, int H, int W, double rt,
                               const uint64_t *seeds, int max_samples, double *const *proposals, const int *S,
                               double *const *planes, int32_t *const *inliers, char *err, size_t errcap);


/* ---- Image segmentation: what dispmap_globalstereo takes its edge weights (:391-403) and the maps of
 * its 14 SegPln proposals (:121-134) from.  A: H x W x 3 uint8, MATLAB column-major; out: H x W uint32, column-major.
 *
 * stereo_segment_ms = vgg_segment_ms(A, h_s, h_r, min_sz) (imrender/vgg/vgg_segment_ms.cxx:18-87 ->
 * msImageProcessor::Segment(sigmaS, sigmaR, minRegion, HIGH_SPEEDUP), seg_ms/msImageProcessor.cpp:703-808): mean-shift
 * filter in LUV (one thread per pixel on the device), then on the host connected components of the filtered image,
 * transitive closure of the region graph and pruning of regions below min_sz pixels; labels from 1 in the reference's
 * numbering.  The scalars are cast as the gateway casts them ((int) h_s, (float) h_r, (int) min_sz).  The optional
 * fifth gateway argument (a weight map, :55-68) has no caller in the reference and is not offered.  The reference reads
 * its speed-up threshold uninitialised (msImageProcessor.h:796); this entry point computes what the reference's own
 * build and the fixtures made with it compute: the upper half of a stack address taken for a float, 4.59e-41 -- in
 * effect "the colours are equal" (csrc/segment_host.h: kSpeedThreshold).
 *
 * stereo_segment_gb = vgg_segment_gb(A, sigma, k, min_sz, compress) (imrender/vgg/vgg_segment_gb.cxx:21-87 ->
 * seg_gb/segment-image.h:181-247): Gaussian smoothing and the weights of the 8-neighbourhood's edges on the device,
 * edges sorted by weight (std::sort, as in the reference: the order of equal weights is the library's), Kruskal with
 * the threshold k / size, components below min_sz joined; the labels are the union-find roots, or -- compress != 0 --
 * 1, 2, ... by first appearance in column-major order.  The library never writes its last row and column: they stay 0
 * (compressed: the label value 0 gets).
 *
 * The staged entry points are the same computation cut at the device / host boundaries: _ms_luv (host: RGB -> LUV,
 * L x 3 floats, pixel y * W + x); _ms_own (device: every pixel's own mode, same layout, and `events`, L bytes: 1 where
 * another pixel's colour came within the reference's speed-up threshold of the pixel's trajectory -- in flat regions);
 * _ms_finish (host: the pixels with events walked again in scan order with the reference's basin-of-attraction
 * bookkeeping, msImageProcessor.cpp:4015-4022,4085-4127,4256-4270 -> the reference's filtered image; *walked = how many);
 * _ms_filter = _ms_own + _ms_finish; _ms_regions (host: labels from a filtered image); _gb_weights (device: 4 floats per
 * pixel y * W + x: to (x+1,y), (x,y+1), (x+1,y+1), (x+1,y-1); 0 where the edge leaves the image), _gb_regions (host).
 * The host stages need no device. */
int stereo_segment_ms(const uint8_t *A, int H, int W, double h_s, double h_r, double min_sz, uint32_t *out, char *err,
                      size_t errcap);
int stereo_segment_gb(const uint8_t *A, int H, int W, double sigma, double k, double min_sz, int compress, uint32_t *out,
                      char *err, size_t errcap);
int stereo_segment_ms_luv(const uint8_t *A, int H, int W, float *luv, char *err, size_t errcap);
int stereo_segment_ms_own(const uint8_t *A, int H, int W, double h_s, double h_r, float *own, uint8_t *events, char *err,
                          size_t errcap);
int stereo_segment_ms_finish(const uint8_t *A, int H, int W, double h_s, double h_r, const float *own, const uint8_t *events,
                             float *filtered, int64_t *walked, char *err, size_t errcap);
int stereo_segment_ms_filter(const uint8_t *A, int H, int W, double h_s, double h_r, float *filtered, char *err,
                             size_t errcap);
int stereo_segment_ms_regions(const float *filtered, int H, int W, double h_r, double min_sz, uint32_t *out, char *err,
                              size_t errcap);
int stereo_segment_gb_weights(const uint8_t *A, int H, int W, double sigma, float *weights, char *err, size_t errcap);
int stereo_segment_gb_regions(const float *weights, int H, int W, double k, double min_sz, int compress, uint32_t *out,
                              char *err, size_t errcap);

#ifdef __cplusplus
}
#endif
#endif /* STEREO_HIP_H_ */
