// TRW-S simultaneous fusion on MI355X (gfx950): kernels, plan object, C ABI.
//
// Replaces the reference's trws_mex gateway + MRFEnergy core + TypeStereo*
// message update (cpp/trws_mex.cpp, cpp/trw-s/{minimize,ordering,MRFEnergy}.cpp,
// cpp/trw-s/typeStereo{Linear,Quadratic}.h).  Built with -ffp-contract=off: the
// reference runs SSE2 doubles without FMA and every value below is computed
// with the same association of + - * / so results are bit identical.
//
// Layout in HBM (all label-fastest, exactly MATLAB's K x N / K x E column major):
//   unary [N][K]   messages [E][K]   q,qprim [E][K] (or one shared positions[K])
//   perm_q, perm_qp [E][K] uint16: ascending sort permutation of q(:,e), qprim(:,e)
// Work decomposition: the reference node order induces a dependency DAG.  One persistent
// launch per sweep walks it as a dataflow: workgroups draw "runs" (a grid row, the border
// chain) from a ticket counter and hand messages over in LDS inside a run, through HBM +
// completion flags between runs.  Four kernel families, identical results
// (stereo_trws_plan_path), one translation unit each over the common device header trws_dev.h:
//   trws_pipe.hip     K <= 64, role-specialised waves, both smoothness kernels
//   trws_pipe2.hip    64 < K <= 128, two labels per lane, linear kernel, per-edge positions
//   trws_wide.hip     64 < K <= 256, shared strictly ascending positions, linear kernel
//   trws_generic.hip  everything else (any graph, K <= 512, min-plus message mode)
// The three descriptor-driven families walk the chain schedule of trws_graph.h; messages take a
// certified min-plus fast path and fall back to the reference's serial envelope
// construction when the certificate fails.  This file: plan object, host logic, C ABI.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <limits>
#include <memory>
#include <mutex>
#include <string>
#include <thread>
#include <utility>
#include <vector>

#include "../../include/stereo_hip.h"
#include "common.h"
#include "trws_graph.h"
#include "trws_dev.h"
#include "trws_launch.h"

namespace stereo {

constexpr int kCtlWords = 8;  // d_ctl: ticket, abort flag, four words of give-up report, two spare

std::string &last_error() {
  static thread_local std::string s;
  return s;
}

namespace {

// Ascending sort permutation of each K-vector (ties: lower index first), one
// wave per vector, bitonic network in LDS.  Replaces the per-edge std::sort of
// trws_mex.cpp:84-119 (which re-sorts after every push_back).
__global__ __launch_bounds__(kWave) void argsort_kernel(const double *vals, uint16_t *perm, int K,
                                                        int P, int64_t count) {
  extern __shared__ __attribute__((aligned(16))) double lds[];
  double *v = lds;
  int *id = (int *)(lds + P);
  const int lane = threadIdx.x;
  for (int64_t a = blockIdx.x; a < count; a += gridDim.x) {
    const double *src = vals + (size_t)a * K;
    for (int i = lane; i < P; i += kWave) {
      v[i] = i < K ? src[i] : __builtin_huge_val();
      id[i] = i < K ? i : (0x10000 + i);
    }
    __syncthreads();
    for (int size = 2; size <= P; size <<= 1) {
      for (int stride = size >> 1; stride > 0; stride >>= 1) {
        for (int t = lane; t < P / 2; t += kWave) {
          const int lo = (t / stride) * (stride * 2) + (t % stride);
          const int hi = lo + stride;
          const bool up = ((lo & size) == 0);
          const double a0 = v[lo], a1 = v[hi];
          const int i0 = id[lo], i1 = id[hi];
          const bool gt = (a0 > a1) || (a0 == a1 && i0 > i1);
          if (gt == up) { v[lo] = a1; v[hi] = a0; id[lo] = i1; id[hi] = i0; }
        }
        __syncthreads();
      }
    }
    uint16_t *dstp = perm + (size_t)a * K;
    for (int i = lane; i < K; i += kWave) dstp[i] = (uint16_t)id[i];
    __syncthreads();
  }
}

// Rows whose ascending order holds two equal values (the order of equal positions needs the
// reference gateway's own sort sequence, see gateway_order below); one thread per row.
__global__ __launch_bounds__(kBlock) void equal_values_kernel(const double *vals, const uint16_t *perm, int K,
                                                             int64_t count, uint8_t *flag) {
  const int64_t a = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  if (a >= count) return;
  const double *v = vals + (size_t)a * K;
  const uint16_t *pm = perm + (size_t)a * K;
  bool eq = false;
  double prev = v[pm[0]];
  for (int k = 1; k < K; ++k) { const double x = v[pm[k]]; eq = eq || x == prev; prev = x; }
  flag[a] = eq ? 1 : 0;
}
__global__ __launch_bounds__(kBlock) void gather_rows_kernel(const double *vals, const int64_t *rows, int64_t n, int K,
                                                            double *out) {
  const int64_t t = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  if (t < n * K) out[t] = vals[(size_t)rows[t / K] * K + t % K];
}
__global__ __launch_bounds__(kBlock) void scatter_perm_kernel(const uint16_t *in, const int64_t *rows, int64_t n, int K,
                                                             uint16_t *perm) {
  const int64_t t = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  if (t < n * K) perm[(size_t)rows[t / K] * K + t % K] = in[t];
}

// ---- single message updates (diagnostic entry point stereo_trws_messages) ------------------
// One wave per message through message_regs -- the routine the pipelined sweep kernel computes
// its messages with (certified fast path, second look, serial construction), table in LDS as
// there -- so that the certificate can be attacked with hand-placed near-tangent cones.
template <int KERNEL, bool SHAREDPOS>
__global__ __launch_bounds__(kWave) void trws_messages_kernel(DevParams p, int K, int64_t M, const double *Di,
                                                             const double *gamma, const double *msg_in,
                                                             const double *qsrc, const double *qdst,
                                                             const double *alpha, const uint16_t *perm, int window,
                                                             double *msg_out, double *vmin, int32_t *serial,
                                                             unsigned long long *counters) {
  __shared__ __attribute__((aligned(16))) double tab[kPipeTab];
  const int lane = threadIdx.x;
  const bool act = lane < K;
  if (lane < 2 * kPipePad) {
    double *e = tab + 4 * (lane < kPipePad ? lane : kWave + lane);
    e[0] = __builtin_huge_val(); e[1] = 0; e[2] = 0; e[3] = 0;
  }
  __syncthreads();
  for (int64_t m = blockIdx.x; m < M; m += gridDim.x) {
    const size_t o = (size_t)m * K + lane;
    const double h = act ? gamma[m] * Di[o] - msg_in[o] : __builtin_huge_val();
    const double qs = act ? qsrc[o] : 0.0, qt = act ? qdst[o] : 0.0;
    p.fallbacks = counters + blockIdx.x;  // (one counter per workgroup: its messages run one after the other)
    unsigned long long before = 0;
    if (lane == 0) before = *p.fallbacks;
    before = __shfl(before, 0, kWave);
    double out = 0;
    const double v = message_regs<KERNEL, SHAREDPOS>(p, K, alpha[m], h, qs, qt, perm + (size_t)m * K, out, lane,
                                                     tab + 4 * kPipePad, window);
    __threadfence();
    if (act) msg_out[o] = out;
    if (lane == 0) { vmin[m] = v; serial[m] = (int32_t)(*p.fallbacks - before); }
  }
}

}  // namespace
}  // namespace stereo

// --------------------------------------------------------------------- plan

using namespace stereo;

struct stereo_trws_plan {
  int kernel = 1, K = 0, Kp = 0, mode = 0, device = 0;
  int64_t N = 0, E = 0;
  std::shared_ptr<const TrwsGraph> graph;  // host-side analysis; shared with the cache of the last connectivity
  // device copies of the graph
  DevBuf<int32_t> d_tail, d_order, d_fptr, d_fidx, d_bptr, d_bidx, d_lbn, d_lbe, d_x;
  DevBuf<uint8_t> d_mdir;
  DevBuf<double> d_gamma, d_msg, d_lbterms, d_eterms;
  // persistent sweep schedule
  DevBuf<int32_t> d_run_order[2], d_chain_run_ptr[2], d_chain_run_order[2];
  DevBuf<int32_t> d_run_ptr[2], d_dep_ptr[2], d_dep_rank[2], d_done, d_ctl;  // d_ctl: [ticket, abort, give-up report x 4]
  DevBuf<int8_t> d_in_slot[2];
  DevBuf<int32_t> d_desc[2];
  bool fast = false;
  bool wide = false;  // 64 < K <= 256 with shared strictly ascending positions: trws_wide_kernel
  bool fast2 = false; // 64 < K <= 128, any positions, both smoothness kernels: trws_pipe2_kernel (when not wide)
  bool wide_allowed = false;
  bool pos_ascending = false;  // shared positions finite and strictly ascending
  double pos_first = 0, pos_last = 0, pos_gap = 0;
  int window = 0;
  double uniform_step = 0;
  DevBuf<unsigned long long> d_fallbacks, d_prof, d_timeline;
  bool certificate = true;
  int epoch = 0;
  long long spin_ticks = 0;  // wall-clock bound of a wait for another workgroup (100 MHz ticks)
  bool fwd_pending = false;  // the forward sweep of the next iteration has already run
  int grid_blocks = 0;
  // inputs (owned unless bound)
  DevBuf<double> o_unary, o_q, o_qprim, o_pos, o_alpha;
  DevBuf<uint16_t> d_perm_q, d_perm_qp, d_perm_pos;
  const double *unary = nullptr, *q = nullptr, *qprim = nullptr, *pos = nullptr, *alpha = nullptr;
  double lambda = 0;
  bool have_inputs = false;
  PinnedBuf<double> h_lb, h_en;
  PinnedBuf<int32_t> h_x, h_ctl;
  hipStream_t issue_stream = nullptr;
  stereo_trws_plan *timed_by = nullptr;  // first plan of the group launch this plan was issued in
  double energy = 0, lb = 0;
  int64_t iterations = 0;
  hipEvent_t ev0 = nullptr, ev1 = nullptr;
  // the lower-bound terms of an iteration go to the host on their own stream while the next
  // launch (forward sweep + primal) runs, and are summed there meanwhile
  hipStream_t copy_stream = nullptr;
  hipEvent_t ev_bwd = nullptr, ev_lb = nullptr;
  bool lb_in_flight = false;
  double sweep_ms = 0;
  int64_t sweep_launches = 0;
  bool time_sweeps = false;
  // row strips (one plan per strip; see DevParams)
  int nstrips = 1, strip = 0;
  DevBuf<int32_t> d_tickets[2];
  int ntickets[2] = {0, 0};
  int64_t n_lb = 0, n_en = 0;  // lower-bound / energy terms this plan writes (strip-local with strips)
  // what the arrays on the device are sized for: the whole problem, or with strips the strip's own
  // nodes + halo and the edges with an own endpoint (StripLayout, trws_graph.h)
  int64_t Nl = 0, El = 0;
  std::unique_ptr<StripLayout> layout;
  DevBuf<int64_t> d_lnodes, d_ledges;  // local -> global ids, for gathering the strip's inputs
  double *peer_msg[2] = {nullptr, nullptr};
  int32_t *peer_done[2] = {nullptr, nullptr}, *peer_x[2] = {nullptr, nullptr};
  bool need_peer[2] = {false, false};
  void *ipc_mapped[2][3] = {{nullptr, nullptr, nullptr}, {nullptr, nullptr, nullptr}};
  DevBuf<DevParams> d_group;         // parameters of the strips launched together with this one (first plan of a group)
  PinnedBuf<DevParams> h_group;
  hipStream_t own_stream = nullptr;  // strips launch concurrently: never on the NULL stream
  bool issued = false;
  int cus = 256;
  // speculative schedule of the long serial run (trws_graph.h: Sweep::Spec; trws_spec.h)
  DevBuf<int32_t> d_spec_run_ptr[2], d_spec_run_order[2], d_spec_kind[2], d_spec_x;
  DevBuf<double> d_spec_rows, d_spec_undo;
  DevBuf<unsigned long long> d_spec_stat;
  DevBuf<DevParams> d_self;
  PinnedBuf<DevParams> h_self;
  bool self_sent = false;
  bool spec_allowed = false;   // the graph has such a run in both directions and STEREO_HIP_TRWS_SPEC is not 0
  bool spec_window = false;    // the positions are uniformly spaced over the window rounded up to four (finish_inputs)
  ~stereo_trws_plan() {
    for (int w = 0; w < 2; ++w)
      for (int k = 0; k < 3; ++k)
        if (ipc_mapped[w][k]) (void)hipIpcCloseMemHandle(ipc_mapped[w][k]);
    if (own_stream) (void)hipStreamDestroy(own_stream);
    if (ev0) (void)hipEventDestroy(ev0);
    if (ev1) (void)hipEventDestroy(ev1);
    if (ev_bwd) (void)hipEventDestroy(ev_bwd);
    if (ev_lb) (void)hipEventDestroy(ev_lb);
    if (copy_stream) (void)hipStreamDestroy(copy_stream);
  }
};

namespace {

size_t persistent_lds_bytes(int Kp) { return generic_lds_bytes(Kp); }

// The speculative schedule runs where its runner's arithmetic is what message_regs returns under a passed certificate:
// trws_pipe_kernel, linear kernel, certified messages, shared strictly ascending positions, uniformly spaced over the
// truncation window (rounded up to a multiple of four entries, at most eight: the runner keeps the window in registers).
bool spec_active(const stereo_trws_plan *P) {
  // (and a handful of resident workgroups: the runner, the segment that commits, the segments in between)
  // (trws_pipe_kernel, or trws_wide_kernel with an even label count: its vector loaders, trws_wspec.h)
  const bool kernel_has_it = (P->fast && !P->wide && !P->fast2) || (P->wide && (P->K & 1) == 0 && P->mode == STEREO_TRWS_MESSAGES_EXACT);
  return P->spec_allowed && P->grid_blocks >= 8 && kernel_has_it && P->nstrips == 1 && P->kernel == 1 && P->certificate && P->pos != nullptr &&
         P->pos_ascending && P->window <= 8 && P->uniform_step != 0 && P->spec_window;
}

DevParams make_params(stereo_trws_plan *P, bool allow_spec = true) {
  DevParams p{};
  p.K = P->K; p.Kp = P->Kp; p.kernel = P->kernel; p.lambda = P->lambda;
  p.unary = P->unary; p.msg = P->d_msg.p; p.q = P->q; p.qprim = P->qprim; p.pos = P->pos;
  p.perm_q = P->d_perm_q.p; p.perm_qp = P->d_perm_qp.p; p.perm_pos = P->d_perm_pos.p;
  p.alpha = P->alpha; p.mdir = P->d_mdir.p; p.tail = P->d_tail.p; p.order = P->d_order.p;
  p.fptr = P->d_fptr.p; p.fidx = P->d_fidx.p; p.bptr = P->d_bptr.p; p.bidx = P->d_bidx.p;
  p.gamma = P->d_gamma.p; p.lb_pos_node = P->d_lbn.p; p.lb_pos_edge = P->d_lbe.p;
  p.lbterms = P->d_lbterms.p; p.eterms = P->d_eterms.p; p.x = P->d_x.p;
  // the descriptor-driven kernels walk the chain schedule (trws_graph.h), the generic ones the
  // rank-contiguous runs
  const bool chain = P->graph->fast_ok && (P->wide || P->fast2 || P->fast);
  for (int d = 0; d < 2; ++d) {
    if (chain) {
      p.run_ptr[d] = P->d_chain_run_ptr[d].p; p.nruns[d] = (int)P->graph->sweep[d].chain_run_ptr.size() - 1;
      p.run_order[d] = P->d_chain_run_order[d].p;
    } else {
      p.run_ptr[d] = P->d_run_ptr[d].p; p.nruns[d] = (int)P->graph->sweep[d].run_ptr.size() - 1;
      p.run_order[d] = P->d_run_order[d].p;
    }
    p.dep_ptr[d] = P->d_dep_ptr[d].p; p.dep_rank[d] = P->d_dep_rank[d].p;
    p.in_slot[d] = P->d_in_slot[d].p;
  }
  for (int d = 0; d < 2; ++d) {
    if (P->nstrips > 1) {  // the strip's own runs, already in ticket order
      p.run_ptr[d] = P->d_chain_run_ptr[d].p; p.nruns[d] = P->ntickets[d]; p.run_order[d] = nullptr;
    }
    p.ntickets[d] = p.nruns[d];
  }
  p.peer_msg0 = P->peer_msg[0]; p.peer_msg1 = P->peer_msg[1];
  p.peer_done0 = P->peer_done[0]; p.peer_done1 = P->peer_done[1];
  p.peer_x0 = P->peer_x[0]; p.peer_x1 = P->peer_x[1];
  p.done = P->d_done.p; p.ticket = P->d_ctl.p; p.abort_flag = P->d_ctl.p + 1; p.N = (int)P->Nl;
  p.spin_ticks = P->spin_ticks;
  p.n_own = P->layout ? (int)P->layout->n_own : (int)P->Nl;
  p.fallbacks = P->d_fallbacks.p; p.certificate = P->certificate ? 1 : 0;
  // MINPLUS in the wide-label regime: the wide kernel's plain min-plus branch
  p.lean = (P->wide && P->mode == STEREO_TRWS_MESSAGES_MINPLUS) ? 1 : 0;
  p.prof = P->d_prof.p;
  p.timeline = P->d_timeline.p;
  p.desc[0] = P->d_desc[0].p; p.desc[1] = P->d_desc[1].p;
  p.prof_run = -1;
  p.window = P->window;
  p.uniform_step = P->uniform_step;
  p.pos_first = P->pos_first; p.pos_last = P->pos_last;
  p.debug = 0;
  if (const char *dbg = std::getenv("STEREO_HIP_TRWS_DEBUG")) p.debug = std::atoi(dbg);
  p.tl_stride = p.nruns[0];
  p.self = P->d_self.p;
  if (allow_spec && spec_active(P)) {
    // the chain schedule with the long run cut into segments + the runner's ticket
    for (int d = 0; d < 2; ++d) {
      const TrwsGraph::Sweep::Spec &sp = P->graph->sweep[d].spec;
      p.run_ptr[d] = P->d_spec_run_ptr[d].p; p.nruns[d] = (int)sp.kind.size();
      p.run_order[d] = P->d_spec_run_order[d].p; p.ntickets[d] = (int)sp.run_order.size();
      p.spec_kind[d] = P->d_spec_kind[d].p; p.spec_c0[d] = sp.c0; p.spec_c1[d] = sp.c1;
    }
    const TrwsGraph::Sweep::Spec &sp = P->graph->sweep[0].spec;
    p.spec_len = sp.seg_len; p.spec_nseg = sp.nseg; p.spec_max_len = sp.max_len;
    p.spec_rows = P->d_spec_rows.p; p.spec_x = P->d_spec_x.p; p.spec_undo = P->d_spec_undo.p; p.spec_stat = P->d_spec_stat.p;
    p.tl_stride = std::max(p.nruns[0], p.nruns[1]);
  }
  p.win_ok = (P->pos_ascending && P->window <= 16 && !(p.debug & 256)) ? 1 : 0;
  p.pos_gap = P->pos_gap;
  if (const char *pr = std::getenv("STEREO_HIP_TRWS_PROF_RUN")) p.prof_run = std::atoi(pr);
  return p;
}

// One persistent launch: 0 = forward, 1 = backward, 2 = forward + primal of the
// previous iteration, 3 = primal only.
void launch_persistent(stereo_trws_plan *P, const DevParams &p, int what, hipStream_t s) {
  const int epoch = ++P->epoch;
  STEREO_HIP_CHECK(hipMemsetAsync(P->d_ctl.p, 0, sizeof(int32_t), s));  // ticket = 0
  if (P->wide) launch_wide(P->kernel, what, std::min(P->grid_blocks, P->cus), s, p, epoch);
  else if (P->fast2) launch_pipe2(P->kernel, P->pos != nullptr, what, std::min(P->grid_blocks, P->cus), s, p, epoch);
  else if (P->fast) {
    int blocks = P->grid_blocks;
    static const char *be = std::getenv("STEREO_HIP_TRWS_BLOCKS");   // (development: workgroups of a pipelined sweep launch)
    if (be && std::atoi(be) > 0) blocks = std::min(blocks, std::atoi(be));
    launch_pipe(P->kernel, P->pos != nullptr, what, blocks, s, p, epoch);
  }
  else launch_generic(P->kernel, P->mode, what, P->grid_blocks, persistent_lds_bytes(P->Kp), s, p, epoch);
  if (what != 3) P->sweep_launches += 1;
}

void persistent_iteration(stereo_trws_plan *P, const DevParams &p, hipStream_t s) {
  if (P->time_sweeps) STEREO_HIP_CHECK(hipEventRecord(P->ev0, s));
  if (!P->fwd_pending) launch_persistent(P, p, 0, s);
  launch_persistent(P, p, 1, s);
  // the backward sweep's lower-bound terms travel while the next launch runs
  STEREO_HIP_CHECK(hipEventRecord(P->ev_bwd, s));
  STEREO_HIP_CHECK(hipStreamWaitEvent(P->copy_stream, P->ev_bwd, 0));
  STEREO_HIP_CHECK(hipMemcpyAsync(P->h_lb.p, P->d_lbterms.p, sizeof(double) * P->n_lb, hipMemcpyDeviceToHost,
                                  P->copy_stream));
  STEREO_HIP_CHECK(hipEventRecord(P->ev_lb, P->copy_stream));
  P->lb_in_flight = true;
  // forward sweep of the NEXT iteration fused with this iteration's primal
  launch_persistent(P, p, 2, s);
  P->fwd_pending = true;
  if (P->time_sweeps) STEREO_HIP_CHECK(hipEventRecord(P->ev1, s));
}

void run_argsort(const double *vals, uint16_t *perm, int K, int64_t count, hipStream_t s) {
  int Pw = 2;
  while (Pw < K) Pw <<= 1;
  const size_t lds = (size_t)Pw * (sizeof(double) + sizeof(int));
  const int64_t grid = std::min<int64_t>(count, 256 * 32);
  hipLaunchKernelGGL(argsort_kernel, dim3((unsigned)grid), dim3(kWave), lds, s, vals, perm, K, Pw, count);
  STEREO_HIP_CHECK(hipGetLastError());
}

// Zero messages (MRFEnergy.cpp:115-133), labels, flags and every piece of iteration state.
void reset_state(stereo_trws_plan *P) {
  STEREO_HIP_CHECK(hipMemset(P->d_msg.p, 0, sizeof(double) * (size_t)P->El * P->K));
  STEREO_HIP_CHECK(hipMemset(P->d_x.p, 0, sizeof(int32_t) * P->Nl));
  STEREO_HIP_CHECK(hipMemset(P->d_done.p, 0, sizeof(int32_t) * P->d_done.n));
  STEREO_HIP_CHECK(hipMemset(P->d_ctl.p, 0, sizeof(int32_t) * kCtlWords));
  STEREO_HIP_CHECK(hipDeviceSynchronize());
  P->iterations = 0; P->energy = 0; P->lb = 0; P->epoch = 0; P->fwd_pending = false;
  P->lb_in_flight = false; P->issued = false;
}

// The order in which the reference's gateway hands EQUAL positions to the message code.
// trws_mex.cpp:84-97 pushes one (value, index) pair at a time and calls std::sort on the whole
// vector after every push, comparing values only (:16-20).  std::sort is not stable: up to 16
// elements it is an insertion sort (equal values stay in index order -- what argsort_kernel
// produces), beyond that its introsort may swap equal values.  Equal positions are no corner
// case: simultaneous_fusion appends the current assignment as a label (dispmap_super.m:158), so
// wherever a proposal's plane is the current plane two labels coincide exactly.  For such vectors
// the same sequence of calls is made here, with the std::sort of the toolchain in use -- what a
// reference built with that toolchain does.
void gateway_order(const double *v, int K, uint16_t *perm) {
  typedef std::pair<double, int> Pair;
  struct Cmp {
    bool operator()(const Pair &a, const Pair &b) const { return a.first < b.first; }
  };
  std::vector<Pair> pr;
  pr.reserve(K);
  for (int j = 0; j < K; ++j) {
    pr.push_back(Pair(v[j], j));
    std::sort(pr.begin(), pr.end(), Cmp());
  }
  for (int j = 0; j < K; ++j) perm[j] = (uint16_t)pr[j].second;
}

// After argsort_kernel: rows with equal values get the gateway's order (K > 16 only, see above).
void fix_equal_positions(const double *d_vals, uint16_t *d_perm, int K, int64_t count) {
  if (K <= 16 || count <= 0) return;
  DevBuf<uint8_t> d_flag;
  d_flag.alloc(count);
  hipLaunchKernelGGL(equal_values_kernel, dim3((unsigned)((count + kBlock - 1) / kBlock)), dim3(kBlock), 0, 0, d_vals,
                     d_perm, K, count, d_flag.p);
  STEREO_HIP_CHECK(hipGetLastError());
  std::vector<uint8_t> flag(count);
  STEREO_HIP_CHECK(hipMemcpy(flag.data(), d_flag.p, count, hipMemcpyDeviceToHost));
  std::vector<int64_t> rows;
  for (int64_t a = 0; a < count; ++a)
    if (flag[a]) rows.push_back(a);
  const int64_t n = (int64_t)rows.size();
  if (n == 0) return;
  DevBuf<int64_t> d_rows;
  DevBuf<double> d_g;
  DevBuf<uint16_t> d_p;
  d_rows.upload(rows.data(), n);
  d_g.alloc((size_t)n * K); d_p.alloc((size_t)n * K);
  const unsigned gb = (unsigned)(((int64_t)n * K + kBlock - 1) / kBlock);
  hipLaunchKernelGGL(gather_rows_kernel, dim3(gb), dim3(kBlock), 0, 0, d_vals, d_rows.p, n, K, d_g.p);
  STEREO_HIP_CHECK(hipGetLastError());
  std::vector<double> g((size_t)n * K);
  STEREO_HIP_CHECK(hipMemcpy(g.data(), d_g.p, sizeof(double) * n * K, hipMemcpyDeviceToHost));
  std::vector<uint16_t> pm((size_t)n * K);
  const int64_t T = std::max<int64_t>(1, std::min<int64_t>({(int64_t)std::thread::hardware_concurrency() / 2, 64, n / 256 + 1}));
  std::vector<std::thread> pool;
  auto work = [&](int64_t a, int64_t b) { for (int64_t i = a; i < b; ++i) gateway_order(&g[(size_t)i * K], K, &pm[(size_t)i * K]); };
  for (int64_t t = 1; t < T; ++t) pool.emplace_back(work, n * t / T, n * (t + 1) / T);
  work(0, n / T);
  for (auto &th : pool) th.join();
  STEREO_HIP_CHECK(hipMemcpy(d_p.p, pm.data(), sizeof(uint16_t) * n * K, hipMemcpyHostToDevice));
  hipLaunchKernelGGL(scatter_perm_kernel, dim3(gb), dim3(kBlock), 0, 0, d_p.p, d_rows.p, n, K, d_perm);
  STEREO_HIP_CHECK(hipGetLastError());
  STEREO_HIP_CHECK(hipDeviceSynchronize());
}

void finish_inputs(stereo_trws_plan *P) {
  // New inputs start a new minimisation: the forward sweep of the next iteration has usually run
  // already with the OLD inputs (persistent_iteration fuses it with the primal pass), so the
  // messages on the device belong to no state the reference could be in with the new ones.
  if (P->iterations > 0 || P->fwd_pending) reset_state(P);
  if (P->pos) {
    P->d_perm_pos.alloc(P->K);
    run_argsort(P->pos, P->d_perm_pos.p, P->K, 1, nullptr);
    fix_equal_positions(P->pos, P->d_perm_pos.p, P->K, 1);
    P->d_perm_q.release(); P->d_perm_qp.release();
  } else {
    P->d_perm_q.alloc((size_t)P->El * P->K);
    P->d_perm_qp.alloc((size_t)P->El * P->K);
    run_argsort(P->q, P->d_perm_q.p, P->K, P->El, nullptr);
    run_argsort(P->qprim, P->d_perm_qp.p, P->K, P->El, nullptr);
    fix_equal_positions(P->q, P->d_perm_q.p, P->K, P->El);
    fix_equal_positions(P->qprim, P->d_perm_qp.p, P->K, P->El);
  }
  STEREO_HIP_CHECK(hipDeviceSynchronize());
  // shared positions that are finite and strictly ascending: truncation window in index steps
  // (windowed min-plus of the pipelined kernel's flat-h path; the wide-label kernel requires it)
  P->wide = false; P->uniform_step = 0; P->pos_ascending = false; P->window = 0; P->spec_window = false;
  if (P->pos && P->lambda >= 0) {
    std::vector<double> hp(P->K);
    STEREO_HIP_CHECK(hipMemcpy(hp.data(), P->pos, sizeof(double) * P->K, hipMemcpyDeviceToHost));
    bool asc = std::isfinite(hp[0]);
    for (int k = 1; k < P->K && asc; ++k) asc = std::isfinite(hp[k]) && hp[k] > hp[k - 1];
    if (asc) {
      // a source farther than lambda from a destination (squared distance for kernel 2)
      // costs >= vTrunc, so min-plus only needs the sources within +-window indices
      int w = 0;
      for (int k = 0, lo = 0; k < P->K; ++k) {
        for (;; ++lo) {
          const double d = hp[k] - hp[lo];
          if ((P->kernel == 1 ? d : d * d) <= (P->kernel == 1 ? P->lambda : P->lambda * (1 + 1e-9))) break;
        }
        w = std::max(w, k - lo);
      }
      P->window = w;
      P->pos_ascending = true;
      P->pos_first = hp[0]; P->pos_last = hp[P->K - 1];
      P->pos_gap = std::numeric_limits<double>::infinity();
      for (int k = 1; k < P->K; ++k) P->pos_gap = std::min(P->pos_gap, hp[k] - hp[k - 1]);
      P->wide = P->wide_allowed;
      // exact arithmetic progression inside the window?  (then alpha |t - q| = alpha |d step| bit for bit)
      P->uniform_step = 0;
      if (w <= 16 && P->K > 1) {
        const double step = hp[1] - hp[0];
        bool uni = step > 0;
        for (int d = 1; d <= w && uni; ++d)
          for (int k = 0; k + d < P->K && uni; ++k) uni = (hp[k + d] - hp[k]) == (double)d * step;
        if (uni) P->uniform_step = step;
        // the runner of the speculative schedule (trws_spec.h) walks the window in groups of four entries: the
        // spacing must hold for those too, and what lies beyond the window must cost >= vTrunc as an index distance
        const int wr = (w + 3) & ~3;
        bool spw = uni && P->kernel == 1;
        for (int d = w + 1; d <= wr && spw; ++d) {
          spw = (double)d * step > P->lambda;
          for (int k = 0; k + d < P->K && spw; ++k) spw = (hp[k + d] - hp[k]) == (double)d * step;
        }
        P->spec_window = spw;
      }
    }
  }
  if (P->nstrips > 1 && !(P->fast || P->fast2 || P->wide))
    throw HipError{"stereo_trws: row strips with these inputs would need the generic kernel, which has no strip support "
                   "(K > 128 or the MINPLUS mode need shared strictly ascending positions)"};
  P->have_inputs = true;
}

}  // namespace

extern "C" {

int stereo_hip_abi_version(void) { return STEREO_HIP_ABI_VERSION; }

int stereo_hip_device_count(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) return 0;
  return n;
}

__global__ void warm_up_kernel() {}

int stereo_hip_warm_up(void) {
  if (stereo_hip_device_count() < 1) return 1;
  if (hipFree(nullptr) != hipSuccess) return 1;
  hipLaunchKernelGGL(warm_up_kernel, dim3(1), dim3(64), 0, 0);
  if (hipDeviceSynchronize() != hipSuccess) return 1;
  // every runtime service the QPBO path uses (cooperative launch, occupancy query, function
  // attributes, the Improve kernels) once, on a frustrated triangle that stays unlabelled
  const double U[3] = {0, 0, 0}, same[3] = {1, 1, 1}, diff[3] = {0, 0, 0};
  const uint32_t conn[6] = {0, 1, 1, 2, 2, 0};
  double lab[3], en = 0, lb = 0, nu = 0;
  char err[256];
  return stereo_rd(U, U, same, diff, diff, same, conn, 3, 3, 1, lab, &en, &lb, &nu, err, sizeof(err));
}

int stereo_hip_device_cus(void) {
  int dev = 0, cus = 0;
  if (hipGetDevice(&dev) != hipSuccess) return 0;
  if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) return 0;
  return cus;
}

int stereo_hip_set_device(int device) {
  if (hipSetDevice(device) != hipSuccess) {
    last_error() = "hipSetDevice failed";
    return 1;
  }
  return 0;
}

const char *stereo_hip_last_error(void) { return last_error().c_str(); }

static int plan_create_impl(int kernel, int K, int64_t N, int64_t E, const uint32_t *conn, int message_mode,
                            const int32_t *owner, int nstrips, int strip, int max_blocks,
                            stereo_trws_plan *share, bool strip_api, stereo_trws_plan **plan, char *err, size_t errcap) {
  if (!plan) return fail("stereo_trws_plan_create: plan is NULL", err, errcap);
  *plan = nullptr;
  if (nstrips < 1 || strip < 0 || strip >= nstrips) return fail("stereo_trws_plan_create: strip out of range", err, errcap);
  if (nstrips > 1 && !owner && !share) return fail("stereo_trws_plan_create: strips need an owner per node", err, errcap);
  if (kernel != 1 && kernel != 2) return fail("Unsupported kernel", err, errcap);
  if (K < 1 || K > 8 * kWave) return fail("stereo_trws: K must be in [1, 512]", err, errcap);
  const int ordering = (message_mode & STEREO_TRWS_ORDER_INDEX) ? 1 : 0;
  message_mode &= ~STEREO_TRWS_ORDER_INDEX;
  if (message_mode != STEREO_TRWS_MESSAGES_EXACT && message_mode != STEREO_TRWS_MESSAGES_MINPLUS)
    return fail("stereo_trws: unknown message mode", err, errcap);
  if (stereo_hip_device_count() < 1)
    return fail("stereo_trws: no HIP device available (the HIP path has no CPU fallback)", err, errcap);
  try {
    std::unique_ptr<stereo_trws_plan> P(new stereo_trws_plan);
    P->kernel = kernel; P->K = K; P->Kp = (K + 1) & ~1; P->mode = message_mode; P->N = N; P->E = E;
    P->nstrips = nstrips; P->strip = strip;
    std::string gerr;
    // Workgroups that stay resident: runs beyond that are cut / dispensed by dependency level.  The
    // bound comes from the device in use (a partitioned or masked MI355X exposes fewer CUs): one
    // workgroup per CU is what is certain to be resident, LDS decides how many more fit.
    STEREO_HIP_CHECK(hipGetDevice(&P->device));
    {
      int cus = 0;
      STEREO_HIP_CHECK(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, P->device));
      P->cus = std::max(cus, 1);
    }
    // (kernel 1 in either message mode -- MINPLUS runs it lean --, kernel 2 with exact messages)
    const bool wide_candidate = (kernel == 1 || message_mode == STEREO_TRWS_MESSAGES_EXACT) && K > kWave && K <= 256;
    const int64_t per_cu = std::min<int64_t>(std::max<int64_t>(1, (int64_t)(160 * 1024) / (int64_t)persistent_lds_bytes(P->Kp)), 4);
    const int64_t capacity = wide_candidate ? P->cus : P->cus * per_cu;
    // The analysis depends on the connectivity only (ordering, lists, schedules: 0.2-0.6 s at Teddy
    // size); consecutive plans for the same image grid -- every trws() call of a fusion loop --
    // share the last one.
    if (share) {
      // the strips of one process share one analysis (it is the same on every strip)
      if (!share->graph || share->N != N || share->E != E || share->graph->nstrips != nstrips)
        return fail("stereo_trws_plan_create: the plan to share the graph analysis with belongs to another problem", err, errcap);
      P->graph = share->graph;
    } else {
      static std::mutex cache_mutex;
      static struct { int64_t N = -1, E = -1, capacity = -1, cus = -1; int nstrips = 1, ordering = 0, seg = 0; std::vector<uint32_t> conn; std::vector<int32_t> owner;
                      std::shared_ptr<const TrwsGraph> g; } cache;
      std::lock_guard<std::mutex> lock(cache_mutex);
      const bool hit = cache.g && cache.N == N && cache.E == E && cache.capacity == capacity && cache.cus == P->cus &&
                       cache.nstrips == nstrips && cache.ordering == ordering && cache.seg == spec_segment_length() &&
                       std::memcmp(cache.conn.data(), conn, sizeof(uint32_t) * 2 * (size_t)E) == 0 &&
                       (nstrips == 1 || std::memcmp(cache.owner.data(), owner, sizeof(int32_t) * (size_t)N) == 0);
      if (hit) {
        P->graph = cache.g;
      } else {
        auto fresh = std::make_shared<TrwsGraph>();
        if (!build_trws_graph(N, E, conn, *fresh, gerr, capacity, nstrips > 1 ? owner : nullptr, nstrips, P->cus, ordering)) return fail(gerr, err, errcap);
        P->graph = fresh;
        if (N <= (1 << 23)) {  // (3000 x 2000: 3 GB of descriptors stay in host memory until the next connectivity)
          cache.N = N; cache.E = E; cache.capacity = capacity; cache.cus = P->cus; cache.nstrips = nstrips; cache.ordering = ordering;
          cache.seg = spec_segment_length();
          cache.conn.assign(conn, conn + 2 * (size_t)E); cache.g = fresh;
          if (nstrips > 1) cache.owner.assign(owner, owner + N); else cache.owner.clear();
        } else {
          cache.g.reset(); cache.conn.clear(); cache.owner.clear(); cache.N = -1;
        }
      }
    }
    const TrwsGraph &g = *P->graph;
    P->Nl = N; P->El = E;
    if (nstrips > 1) {
      if (!g.fast_ok)
        return fail("stereo_trws: row strips need a graph the pipelined kernels take (<= 8 edges per node)", err, errcap);
      P->layout.reset(new StripLayout);
      if (!build_strip_layout(g, strip, *P->layout, gerr)) return fail(gerr, err, errcap);
      const StripLayout &L = *P->layout;
      P->Nl = (int64_t)L.nodes.size(); P->El = (int64_t)L.edges.size();
      std::vector<int64_t> ids(L.nodes.begin(), L.nodes.end());
      P->d_lnodes.upload(ids.data(), ids.size());
      ids.assign(L.edges.begin(), L.edges.end());
      P->d_ledges.upload(ids.data(), ids.size());
      for (int d = 0; d < 2; ++d) {
        P->d_desc[d].upload(L.desc[d].data(), L.desc[d].size());
        P->d_chain_run_ptr[d].upload(L.run_ptr[d].data(), L.run_ptr[d].size());
        P->ntickets[d] = (int)L.run_ptr[d].size() - 1;
        P->need_peer[d] = L.need_peer[d];
      }
      // (the generic kernels' index arrays are not needed: a strip runs a descriptor-driven kernel)
      P->layout->desc[0] = std::vector<int32_t>(); P->layout->desc[1] = std::vector<int32_t>();
    } else {
    P->d_tail.upload(g.tail.data(), g.tail.size());
    P->d_order.upload(g.order.data(), g.order.size());
    P->d_fptr.upload(g.fptr.data(), g.fptr.size());
    P->d_fidx.upload(g.fidx.data(), g.fidx.size());
    P->d_bptr.upload(g.bptr.data(), g.bptr.size());
    P->d_bidx.upload(g.bidx.data(), g.bidx.size());
    P->d_lbn.upload(g.lb_pos_node.data(), g.lb_pos_node.size());
    P->d_lbe.upload(g.lb_pos_edge.data(), g.lb_pos_edge.size());
    P->d_mdir.upload(g.mdir.data(), g.mdir.size());
    P->d_gamma.upload(g.gamma.data(), g.gamma.size());
    for (int d = 0; d < 2; ++d) {
      const TrwsGraph::Sweep &S = g.sweep[d];
      P->d_run_ptr[d].upload(S.run_ptr.data(), S.run_ptr.size());
      if (!S.run_order.empty()) P->d_run_order[d].upload(S.run_order.data(), S.run_order.size());
      P->d_dep_ptr[d].upload(S.dep_ptr.data(), S.dep_ptr.size());
      P->d_dep_rank[d].upload(S.dep_rank.data(), S.dep_rank.size());
      P->d_in_slot[d].upload(S.in_slot.data(), S.in_slot.size());
      if (g.fast_ok) {
        P->d_desc[d].upload(S.desc.data(), S.desc.size());
        P->d_chain_run_ptr[d].upload(S.chain_run_ptr.data(), S.chain_run_ptr.size());
        if (!S.chain_run_order.empty()) P->d_chain_run_order[d].upload(S.chain_run_order.data(), S.chain_run_order.size());
      }
    }
    }
    {
      const TrwsGraph::Sweep::Spec &s0 = g.sweep[0].spec, &s1 = g.sweep[1].spec;
      bool on = g.fast_ok && nstrips == 1 && s0.ok && s1.ok && s0.nseg == s1.nseg && s0.seg_len == s1.seg_len &&
                (K <= kWave || (K <= 256 && (K & 1) == 0 && kernel == 1));   // (trws_pipe_kernel; trws_wide_kernel with its vector loaders)
      if (const char *e = std::getenv("STEREO_HIP_TRWS_SPEC")) on = on && std::atoi(e) != 0;
      P->spec_allowed = on;
      if (on) {
        for (int d = 0; d < 2; ++d) {
          const TrwsGraph::Sweep::Spec &sp = g.sweep[d].spec;
          P->d_spec_run_ptr[d].upload(sp.run_ptr.data(), sp.run_ptr.size());
          P->d_spec_run_order[d].upload(sp.run_order.data(), sp.run_order.size());
          P->d_spec_kind[d].upload(sp.kind.data(), sp.kind.size());
        }
        const size_t ml = (size_t)std::max(s0.max_len, s1.max_len);
        P->d_spec_rows.alloc((size_t)s0.nseg * 8 * K);
        P->d_spec_undo.alloc((size_t)s0.nseg * ml * 4 * K);
        P->d_spec_x.alloc(s0.nseg);
        P->d_spec_stat.alloc(32);
        STEREO_HIP_CHECK(hipMemset(P->d_spec_stat.p, 0, 32 * sizeof(unsigned long long)));
        STEREO_HIP_CHECK(hipMemset(P->d_spec_rows.p, 0, sizeof(double) * (size_t)s0.nseg * 8 * K));
        STEREO_HIP_CHECK(hipMemset(P->d_spec_x.p, 0, sizeof(int32_t) * s0.nseg));
      }
      P->d_self.alloc(1); P->h_self.alloc(1);
    }
    P->fast = g.fast_ok && K <= kWave && message_mode == STEREO_TRWS_MESSAGES_EXACT;
    P->wide_allowed = g.fast_ok && (kernel == 1 || message_mode == STEREO_TRWS_MESSAGES_EXACT) && K > kWave && K <= 256;
    P->fast2 = g.fast_ok && K > kWave && K <= 2 * kWave && message_mode == STEREO_TRWS_MESSAGES_EXACT;   // (both smoothness kernels since round 5)
    if (const char *f = std::getenv("STEREO_HIP_TRWS_FAST")) {
      P->fast = P->fast && std::string(f) != "0";
      P->wide_allowed = P->wide_allowed && std::string(f) != "0";
      P->fast2 = P->fast2 && std::string(f) != "0";
    }
    if (nstrips > 1) {
      // a strip walks the chain schedule with one of the descriptor-driven kernels
      if (!(P->fast || P->wide_allowed || P->fast2))
        return fail("stereo_trws: row strips need a graph and label count the pipelined kernels take "
                    "(<= 8 edges per node; K <= 64, or K <= 128 with per-edge positions, or K <= 256 with shared ascending positions)", err, errcap);
    }
    if (strip_api) STEREO_HIP_CHECK(hipStreamCreateWithFlags(&P->own_stream, hipStreamNonBlocking));
    P->n_lb = nstrips > 1 ? g.strip_lb_terms[strip] : g.lb_terms;
    P->n_en = nstrips > 1 ? g.strip_nodes[strip] : N;
    // Strips that may have a neighbour on ANOTHER GPU keep the three arrays the neighbour writes into
    // (messages, flags, labels) in fine-grained memory (common.h); strips that share the only visible
    // device (logical strips, tests) stay in ordinary memory.  STEREO_HIP_STRIPS_FINEGRAINED=0/1 overrides.
    bool fine = nstrips > 1 && stereo_hip_device_count() > 1;
    if (const char *fg = std::getenv("STEREO_HIP_STRIPS_FINEGRAINED")) fine = nstrips > 1 && std::atoi(fg) != 0;
    // (behind the nodes' flags: the speculative schedule's, two per segment)
    const size_t n_flags = (size_t)P->Nl + (P->spec_allowed ? 2 * (size_t)g.sweep[0].spec.nseg + 2 : 0);
    if (fine) P->d_done.alloc_fine_grained(n_flags); else P->d_done.alloc(n_flags);
    P->d_ctl.alloc(kCtlWords);
    P->d_fallbacks.alloc(1);
    STEREO_HIP_CHECK(hipMemset(P->d_fallbacks.p, 0, sizeof(unsigned long long)));
    if (const char *c = std::getenv("STEREO_HIP_TRWS_CERTIFICATE")) P->certificate = std::string(c) != "0";
    {
      // how long a visit may wait for another workgroup before the launch gives up: inside one launch
      // a flag is late by microseconds; a neighbouring strip's launch belongs to another process and
      // may start seconds later (code-object load, a busy host)
      double secs = nstrips > 1 ? 120.0 : 20.0;
      if (const char *c = std::getenv("STEREO_HIP_TRWS_SPIN_SECONDS")) secs = std::max(0.001, std::atof(c));
      P->spin_ticks = (long long)(secs * 1e8);
    }
    if (std::getenv("STEREO_HIP_TRWS_PROF")) { P->d_prof.alloc(64); STEREO_HIP_CHECK(hipMemset(P->d_prof.p, 0, 512)); }
    if (std::getenv("STEREO_HIP_TRWS_TIMELINE"))
      P->d_timeline.alloc(4 * std::max({g.sweep[0].run_ptr.size(), g.sweep[0].chain_run_ptr.size(), g.sweep[1].chain_run_ptr.size(),
                                         g.sweep[0].spec.kind.size() + 1, g.sweep[1].spec.kind.size() + 1}) + 8);
    STEREO_HIP_CHECK(hipMemset(P->d_done.p, 0, sizeof(int32_t) * P->d_done.n));
    STEREO_HIP_CHECK(hipMemset(P->d_ctl.p, 0, sizeof(int32_t) * kCtlWords));
    {
      // one workgroup per concurrently active run, capped by what stays resident
      int64_t runs = std::max<int64_t>((int64_t)g.sweep[0].run_ptr.size() - 1, 1);
      if (g.fast_ok)
        runs = std::max<int64_t>({runs, (int64_t)g.sweep[0].chain_run_ptr.size() - 1, (int64_t)g.sweep[1].chain_run_ptr.size() - 1});
      if (nstrips > 1) runs = std::max<int64_t>({1, (int64_t)P->ntickets[0], (int64_t)P->ntickets[1]});
      if (P->spec_allowed) runs = std::max<int64_t>({runs, (int64_t)g.sweep[0].spec.run_order.size(), (int64_t)g.sweep[1].spec.run_order.size()});
      P->grid_blocks = (int)std::min<int64_t>(runs, P->cus * per_cu);
      if (max_blocks > 0) P->grid_blocks = std::min(P->grid_blocks, max_blocks);
    }
    if (fine) P->d_msg.alloc_fine_grained((size_t)P->El * K); else P->d_msg.alloc((size_t)P->El * K);
    P->d_lbterms.alloc(P->n_lb);
    P->d_eterms.alloc(P->n_en);
    if (fine) P->d_x.alloc_fine_grained(P->Nl); else P->d_x.alloc(P->Nl);
    P->h_lb.alloc(P->n_lb); P->h_en.alloc(P->n_en); P->h_x.alloc(P->Nl); P->h_ctl.alloc(kCtlWords);
    std::memset(P->h_ctl.p, 0, sizeof(int32_t) * kCtlWords);
    STEREO_HIP_CHECK(hipMemset(P->d_msg.p, 0, sizeof(double) * (size_t)P->El * K));
    STEREO_HIP_CHECK(hipMemset(P->d_x.p, 0, sizeof(int32_t) * P->Nl));
    STEREO_HIP_CHECK(hipEventCreate(&P->ev0));
    STEREO_HIP_CHECK(hipEventCreate(&P->ev1));
    STEREO_HIP_CHECK(hipEventCreateWithFlags(&P->ev_bwd, hipEventDisableTiming));
    STEREO_HIP_CHECK(hipEventCreateWithFlags(&P->ev_lb, hipEventDisableTiming));
    STEREO_HIP_CHECK(hipStreamCreateWithFlags(&P->copy_stream, hipStreamNonBlocking));
    STEREO_HIP_CHECK(hipDeviceSynchronize());
    // every sweep kernel may need more than the default 64 KiB of dynamic LDS
    const int plds = (int)persistent_lds_bytes(P->Kp);
    if (plds > 160 * 1024) return fail("stereo_trws: K too large for LDS", err, errcap);
    generic_set_attributes(plds);
    if (P->fast || strip_api) pipe_set_attributes();
    if (P->fast2) pipe2_set_attributes();
    if (P->wide_allowed || strip_api) wide_set_attributes();
    *plan = P.release();
    return 0;
  } catch (const HipError &e) {
    return fail(e.msg, err, errcap);
  } catch (const std::exception &e) {
    return fail(std::string("stereo_trws_plan_create: ") + e.what(), err, errcap);
  }
}

int stereo_trws_plan_create(int kernel, int K, int64_t N, int64_t E, const uint32_t *conn,
                            int message_mode, stereo_trws_plan **plan, char *err, size_t errcap) {
  return plan_create_impl(kernel, K, N, E, conn, message_mode, nullptr, 1, 0, 0, nullptr, false, plan, err, errcap);
}

int stereo_trws_plan_create_strip(int kernel, int K, int64_t N, int64_t E, const uint32_t *conn, int message_mode,
                                  const int32_t *owner, int nstrips, int strip, int max_workgroups,
                                  stereo_trws_plan *share_analysis_with, stereo_trws_plan **plan, char *err,
                                  size_t errcap) {
  return plan_create_impl(kernel, K, N, E, conn, message_mode, owner, nstrips, strip, max_workgroups,
                          share_analysis_with, true, plan, err, errcap);
}

void stereo_trws_plan_destroy(stereo_trws_plan *plan) {
  DeviceScope device_scope_(plan ? plan->device : -1);
  if (plan && plan->d_timeline.p) {
    const bool chain = plan->graph->fast_ok && (plan->wide || plan->fast2 || plan->fast);
    const bool spec = spec_active(plan);
    const size_t R = spec ? std::max(plan->graph->sweep[0].spec.kind.size(), plan->graph->sweep[1].spec.kind.size())
                          : (chain ? plan->graph->sweep[0].chain_run_ptr.size() : plan->graph->sweep[0].run_ptr.size()) - 1;
    std::vector<unsigned long long> t(4 * (R + 1) + 8);
    if (hipMemcpy(t.data(), plan->d_timeline.p, sizeof(unsigned long long) * (4 * R + 4), hipMemcpyDeviceToHost) == hipSuccess) {
      if (spec)
        for (int d = 0; d < 2; ++d) {
          const auto &sp = plan->graph->sweep[d].spec;
          const unsigned long long t0 = t[(2 * R + d) * 2];
          std::fprintf(stderr, "[stereo_hip timeline] dir %d speculative: runner %.0f us; segments (us since the runner started, start..commit): ", d,
                       (t[(2 * R + d) * 2 + 1] - t0) / 100.0);
          for (int q = 0; q < sp.nseg; q += std::max(1, sp.nseg / 8))
            std::fprintf(stderr, "seg%d[%.0f..%.0f] ", q, ((double)t[((size_t)d * R + sp.run + q) * 2] - (double)t0) / 100.0,
                         ((double)t[((size_t)d * R + sp.run + q) * 2 + 1] - (double)t0) / 100.0);
          std::fprintf(stderr, "last[..%.0f]\n", ((double)t[((size_t)d * R + sp.run + sp.nseg - 1) * 2 + 1] - (double)t0) / 100.0);
        }
      for (int d = 0; d < 2; ++d) {
        const unsigned long long t0 = t[(size_t)d * R * 2];
        std::fprintf(stderr, "[stereo_hip timeline] dir %d (us since run 0 start): ", d);
        for (size_t r = 0; r < R; r += (r < 8 ? 1 : R / 12 + 1))
          std::fprintf(stderr, "run%zu[%.0f..%.0f] ", r, (t[(d * R + r) * 2] - t0) / 100.0, (t[(d * R + r) * 2 + 1] - t0) / 100.0);
        std::fprintf(stderr, "last[%.0f..%.0f]\n", (t[(d * R + R - 1) * 2] - t0) / 100.0, (t[(d * R + R - 1) * 2 + 1] - t0) / 100.0);
      }
    }
  }
  if (plan && plan->d_prof.p) {
    unsigned long long v[64];
    if (hipMemcpy(v, plan->d_prof.p, sizeof(v), hipMemcpyDeviceToHost) == hipSuccess) {
      if (!plan->wide)
        std::fprintf(stderr, "[stereo_hip prof] cycles: p0 %llu p1 %llu p2 %llu p3 %llu p4 %llu | p5 %llu steps %llu\n",
                     v[0], v[1], v[2], v[3], v[4], v[5], v[6]);
      if (!plan->wide && v[6]) {
        std::fprintf(stderr, "[stereo_hip prof] cycles from barrier to barrier arrival per visit, per wave:");
        for (int i = 0; i < 12; ++i) std::fprintf(stderr, " %.0f", (double)v[32 + i] / v[6]);
        std::fprintf(stderr, "\n");
        if (v[48] | v[49] | v[50])  // -DSTEREO_HIP_VISIT_PROFILE
          std::fprintf(stderr, "[stereo_hip prof] wave 0 per visit: stage words %.0f | Di %.0f | H, positions %.0f | message %.0f | "
                               "hand-over %.0f | barrier %.0f\n", (double)v[48] / v[6], (double)v[49] / v[6], (double)v[50] / v[6],
                       (double)v[51] / v[6], (double)v[52] / v[6], (double)v[53] / v[6]);
        if ((v[48] | v[49] | v[50]) && v[19] && v[22])
          std::fprintf(stderr, "[stereo_hip prof] loader (steady state, per visit): until it polls %.0f | flags %.0f | fetch + stage %.0f; "
                               "storer: until the drain %.0f | drain %.0f\n", (double)v[16] / v[19], (double)v[17] / v[19],
                       (double)v[18] / v[19], (double)v[20] / v[22], (double)v[21] / v[22]);
        if (v[48] | v[49] | v[50])
          std::fprintf(stderr, "[stereo_hip prof] of the message: reduction + table %.0f | pair loop / flat path %.0f | margins + second look "
                               "%.0f | serial construction + walk %.0f | minimum %.0f\n", (double)v[56] / v[6], (double)v[57] / v[6],
                       (double)v[58] / v[6], (double)v[59] / v[6], (double)v[60] / v[6]);
      }
      if (!plan->wide && (v[56] | v[57] | v[58] | v[59]) && !(v[48] | v[49] | v[50]))
        std::fprintf(stderr, "[stereo_hip prof messages] useful sources per message: <= 8: %llu, <= 16: %llu, <= 32: %llu, more (flat path): %llu\n", v[56], v[57], v[58], v[59]);
      if (!plan->wide && v[9])
        std::fprintf(stderr, "[stereo_hip prof messages] certified attempt %.0f cycles x %llu | second look %.0f x %llu | "
                             "serial construction %.0f x %llu | walk %.0f x %llu\n",
                     (double)v[8] / v[9], v[9], v[11] ? (double)v[10] / v[11] : 0.0, v[11], v[13] ? (double)v[12] / v[13] : 0.0,
                     v[13], v[15] ? (double)v[14] / v[15] : 0.0, v[15]);
      if (!plan->wide && v[17])
        std::fprintf(stderr, "[stereo_hip prof closed form] thresholds %.0f cycles | rows %.0f | scan + fixed point %.0f | slots + fill %.0f | "
                             "x %llu, extra rounds %.2f (%.2f with late tests), pushed %.1f, rows computed %.1f, top-segment check failed %llu, "
                             "up-front tests %.0f cycles, rounds %.0f cycles\n",
                     (double)v[16] / v[17], v[19] ? (double)v[18] / v[19] : 0.0, v[21] ? (double)v[20] / v[21] : 0.0,
                     v[23] ? (double)v[22] / v[23] : 0.0, v[17], v[21] ? (double)v[24] / v[21] : 0.0, v[21] ? (double)v[28] / v[21] : 0.0,
                     v[21] ? (double)v[26] / v[21] : 0.0, v[21] ? (double)v[27] / v[21] : 0.0, v[25],
                     v[21] ? (double)v[29] / v[21] : 0.0, v[21] ? (double)v[30] / v[21] : 0.0);
      if (plan->wide && v[22]) {
        std::fprintf(stderr, "[stereo_hip prof wide] cycles per visit of wave 0:");
        for (int i = 0; i < 16; ++i) std::fprintf(stderr, " [%d] %.0f", i, (double)v[i] / v[22]);
        std::fprintf(stderr, " | loader A %.0f B %.0f storer %.0f primal %.0f | hw barrier wait %.0f | visits %llu\n",
                     (double)v[16] / v[22], (double)v[17] / v[22], (double)v[18] / v[22], (double)v[19] / v[22],
                     (double)v[21] / v[22], v[22]);
        std::fprintf(stderr, "[stereo_hip prof wide] cycles from barrier to barrier arrival, per wave:");
        for (int i = 0; i < 16; ++i) std::fprintf(stderr, " %.0f", (double)v[32 + i] / v[22]);
        if (v[28] | v[29])
          std::fprintf(stderr, "\n[stereo_hip prof wide] loader B: request inside the node's own visit %.0f cycles x %llu | staging (incl. wait for "
                               "parked loads) %.0f per visit | request two visits ahead %.0f x %llu",
                       v[28] ? (double)v[24] / v[28] : 0.0, v[28], (double)v[25] / v[22], v[29] ? (double)v[26] / v[29] : 0.0, v[29]);
        std::fprintf(stderr, "\n[stereo_hip prof wide] visits with more than 8000 cycles to the barrier, per wave:");
        for (int i = 0; i < 12; ++i) std::fprintf(stderr, " %llu", v[48 + i]);
        std::fprintf(stderr, "\n");
      }
    }
  }
  delete plan;
}

int stereo_trws_plan_upload(stereo_trws_plan *P, const double *unary, const double *q,
                            const double *qprim, const double *positions, const double *alphas,
                            double tol, char *err, size_t errcap) {
  DeviceScope device_scope_(P ? P->device : -1);
  if (!P || !unary || !alphas) return fail("stereo_trws_plan_upload: NULL argument", err, errcap);
  const bool shared = (q == nullptr && qprim == nullptr);
  if (shared && !positions) return fail("stereo_trws_plan_upload: need q/qprim or positions", err, errcap);
  if (!shared && (!q || !qprim)) return fail("stereo_trws_plan_upload: q and qprim must both be given", err, errcap);
  try {
    const size_t K = P->K;
    // a strip keeps the rows of its own nodes + halo and of the edges with an own endpoint
    std::vector<double> part;
    auto rows = [&](const double *full, const std::vector<int32_t> &ids, size_t width) {
      part.resize(ids.size() * width);
      for (size_t i = 0; i < ids.size(); ++i) std::memcpy(&part[i * width], full + (size_t)ids[i] * width, sizeof(double) * width);
      return part.data();
    };
    const bool local = P->nstrips > 1;
    P->o_unary.upload(local ? rows(unary, P->layout->nodes, K) : unary, (size_t)P->Nl * K);
    P->o_alpha.upload(local ? rows(alphas, P->layout->edges, 1) : alphas, (size_t)P->El);
    P->unary = P->o_unary.p; P->alpha = P->o_alpha.p;
    if (shared) {
      P->o_pos.upload(positions, K);
      P->pos = P->o_pos.p; P->q = P->qprim = nullptr;
      P->o_q.release(); P->o_qprim.release();
    } else {
      P->o_q.upload(local ? rows(q, P->layout->edges, K) : q, (size_t)P->El * K);
      P->o_qprim.upload(local ? rows(qprim, P->layout->edges, K) : qprim, (size_t)P->El * K);
      P->q = P->o_q.p; P->qprim = P->o_qprim.p; P->pos = nullptr;
    }
    P->lambda = tol;
    finish_inputs(P);
    return 0;
  } catch (const HipError &e) {
    return fail(e.msg, err, errcap);
  }
}

int stereo_trws_plan_bind_device(stereo_trws_plan *P, const double *d_unary, const double *d_q,
                                 const double *d_qprim, const double *d_positions,
                                 const double *d_alphas, double tol, char *err, size_t errcap) {
  DeviceScope device_scope_(P ? P->device : -1);
  if (!P || !d_unary || !d_alphas) return fail("stereo_trws_plan_bind_device: NULL argument", err, errcap);
  const bool shared = (d_q == nullptr && d_qprim == nullptr);
  if (shared && !d_positions) return fail("stereo_trws_plan_bind_device: need q/qprim or positions", err, errcap);
  if (!shared && (!d_q || !d_qprim)) return fail("stereo_trws_plan_bind_device: q and qprim must both be given", err, errcap);
  try {
    P->lambda = tol;
    if (P->nstrips > 1) {
      // the arrays cover the whole problem: the strip gathers its rows into arrays of its own
      // (the caller may free the full ones afterwards; stereo_trws_plan_bind_device_strip takes
      // arrays that are strip-local already)
      auto rows = [&](const double *full, DevBuf<double> &own, const DevBuf<int64_t> &ids, int64_t n, int width) {
        own.alloc((size_t)n * width);
        const unsigned gb = (unsigned)((n * width + kBlock - 1) / kBlock);
        hipLaunchKernelGGL(gather_rows_kernel, dim3(gb), dim3(kBlock), 0, 0, full, ids.p, n, width, own.p);
        STEREO_HIP_CHECK(hipGetLastError());
        return (const double *)own.p;
      };
      P->unary = rows(d_unary, P->o_unary, P->d_lnodes, P->Nl, P->K);
      P->alpha = rows(d_alphas, P->o_alpha, P->d_ledges, P->El, 1);
      if (shared) { P->pos = d_positions; P->q = P->qprim = nullptr; }
      else {
        P->q = rows(d_q, P->o_q, P->d_ledges, P->El, P->K);
        P->qprim = rows(d_qprim, P->o_qprim, P->d_ledges, P->El, P->K);
        P->pos = nullptr;
      }
      STEREO_HIP_CHECK(hipDeviceSynchronize());
    } else {
      P->unary = d_unary; P->alpha = d_alphas;
      if (shared) { P->pos = d_positions; P->q = P->qprim = nullptr; }
      else { P->q = d_q; P->qprim = d_qprim; P->pos = nullptr; }
    }
    finish_inputs(P);
    return 0;
  } catch (const HipError &e) {
    return fail(e.msg, err, errcap);
  }
}

int stereo_trws_plan_bind_device_strip(stereo_trws_plan *P, const double *d_unary, const double *d_q,
                                       const double *d_qprim, const double *d_positions,
                                       const double *d_alphas, double tol, char *err, size_t errcap) {
  DeviceScope device_scope_(P ? P->device : -1);
  if (!P || !d_unary || !d_alphas) return fail("stereo_trws_plan_bind_device_strip: NULL argument", err, errcap);
  const bool shared = (d_q == nullptr && d_qprim == nullptr);
  if (shared && !d_positions) return fail("stereo_trws_plan_bind_device_strip: need q/qprim or positions", err, errcap);
  if (!shared && (!d_q || !d_qprim)) return fail("stereo_trws_plan_bind_device_strip: q and qprim must both be given", err, errcap);
  try {
    P->unary = d_unary; P->alpha = d_alphas; P->lambda = tol;
    if (shared) { P->pos = d_positions; P->q = P->qprim = nullptr; }
    else { P->q = d_q; P->qprim = d_qprim; P->pos = nullptr; }
    finish_inputs(P);
    return 0;
  } catch (const HipError &e) {
    return fail(e.msg, err, errcap);
  }
}

int stereo_trws_plan_strip_layout(stereo_trws_plan *P, int64_t *n_nodes, int64_t *n_own, int64_t *n_edges,
                                  int32_t *nodes, int32_t *edges) {
  if (!P) return 1;
  if (n_nodes) *n_nodes = P->Nl;
  if (n_own) *n_own = P->n_en;
  if (n_edges) *n_edges = P->El;
  for (int64_t i = 0; nodes && i < P->Nl; ++i) nodes[i] = P->layout ? P->layout->nodes[i] : (int32_t)i;
  for (int64_t e = 0; edges && e < P->El; ++e) edges[e] = P->layout ? P->layout->edges[e] : (int32_t)e;
  return 0;
}

int stereo_trws_plan_reset(stereo_trws_plan *P, char *err, size_t errcap) {
  DeviceScope device_scope_(P ? P->device : -1);
  if (!P) return fail("stereo_trws_plan_reset: NULL plan", err, errcap);
  try {
    reset_state(P);
    return 0;
  } catch (const HipError &e) {
    return fail(e.msg, err, errcap);
  }
}

// One iteration's launches and device-to-host copies, without waiting for any of them.
static void issue_iteration(stereo_trws_plan *P, const DevParams &p, hipStream_t s) {
  persistent_iteration(P, p, s);
  if (!P->lb_in_flight)
    STEREO_HIP_CHECK(hipMemcpyAsync(P->h_lb.p, P->d_lbterms.p, sizeof(double) * P->n_lb, hipMemcpyDeviceToHost, s));
  STEREO_HIP_CHECK(hipMemcpyAsync(P->h_en.p, P->d_eterms.p, sizeof(double) * P->n_en, hipMemcpyDeviceToHost, s));
  STEREO_HIP_CHECK(hipMemcpyAsync(P->h_ctl.p, P->d_ctl.p, kCtlWords * sizeof(int32_t), hipMemcpyDeviceToHost, s));
  P->issued = true;
}

// Waits for the iteration issued last and sums its lower-bound and energy terms in the
// reference's order (minimize.cpp:82,92 and :260): sequential, bit exact.  Returns false if a
// sweep gave up waiting on a dependency flag.
static bool collect_iteration(stereo_trws_plan *P, hipStream_t s, double *lb_out, double *en_out) {
  double lb = 0, en = 0;
  if (P->lb_in_flight) {  // summed while the forward sweep + primal launch is still running
    STEREO_HIP_CHECK(hipEventSynchronize(P->ev_lb));
    for (int64_t i = 0; i < P->n_lb; ++i) lb += P->h_lb.p[i];
  }
  STEREO_HIP_CHECK(hipStreamSynchronize(s));
  P->issued = false;
  if (P->h_ctl.p[1]) return false;
  if (P->time_sweeps && (!P->timed_by || P->timed_by->time_sweeps)) {
    float ms = 0;
    stereo_trws_plan *T = P->timed_by ? P->timed_by : P;
    STEREO_HIP_CHECK(hipEventElapsedTime(&ms, T->ev0, T->ev1));
    P->sweep_ms += ms;
  }
  if (!P->lb_in_flight)
    for (int64_t i = 0; i < P->n_lb; ++i) lb += P->h_lb.p[i];
  P->lb_in_flight = false;
  for (int64_t i = 0; i < P->n_en; ++i) en += P->h_en.p[i];
  *lb_out = lb; *en_out = en;
  return true;
}

// What a sweep that gave up says (h_ctl[2..5] = report_give_up's words; all zero when the give-up came
// from a wait inside a workgroup, which has no report).
static std::string gave_up_text(const stereo_trws_plan *P) {
  const int32_t *c = P->h_ctl.p;
  char b[512];
  const double secs = (double)P->spin_ticks / 1e8;
  if (c[2] == 0 && c[3] == 0 && c[4] == 0 && c[5] == 0) {
    std::snprintf(b, sizeof(b), "stereo_trws: a persistent sweep gave up waiting on a dependency flag (strip %d of %d, device %d)",
                  P->strip, P->nstrips, P->device);
  } else {
    const bool halo = P->layout && (int64_t)c[3] >= P->layout->n_own;  // (strip-local ids: own nodes first, then the halo)
    std::snprintf(b, sizeof(b), "stereo_trws: a persistent sweep gave up waiting on a dependency flag: strip %d of %d (device %d), "
                  "the visit of rank %d waited %.0f s for the completion flag of rank %d (found %d, expected epoch %d)%s",
                  P->strip, P->nstrips, P->device, c[2], secs, c[3], c[4], c[5],
                  halo ? " -- a node of the NEIGHBOURING strip: is that strip's process / launch running?" : "");
  }
  return b;
}

static int strip_ready(stereo_trws_plan *P, const char *who, char *err, size_t errcap) {
  if (!P) return fail(std::string(who) + ": NULL plan", err, errcap);
  if (!P->have_inputs) return fail(std::string(who) + ": no inputs uploaded/bound", err, errcap);
  for (int w = 0; w < 2; ++w)
    if (P->need_peer[w] && !(P->peer_msg[w] && P->peer_done[w] && P->peer_x[w]))
      return fail(std::string(who) + ": strip is not connected to its " + (w ? "next" : "previous") + " neighbour", err, errcap);
  return 0;
}

int stereo_trws_plan_iterate(stereo_trws_plan *P, int iters, double max_relgap, void *stream,
                             int *done_iters, int *stopped, char *err, size_t errcap) {
  DeviceScope device_scope_(P ? P->device : -1);
  if (!P) return fail("stereo_trws_plan_iterate: NULL plan", err, errcap);
  if (!P->have_inputs) return fail("stereo_trws_plan_iterate: no inputs uploaded/bound", err, errcap);
  if (P->nstrips > 1)
    return fail("stereo_trws_plan_iterate: a strip iterates through stereo_trws_plan_issue / _collect / _commit "
                "(its energy and bound are partial sums)", err, errcap);
  if (done_iters) *done_iters = 0;
  if (stopped) *stopped = 0;
  hipStream_t s = (hipStream_t)stream;
  try {
    const DevParams p = make_params(P);
    // (the block once more in global memory: chain_runner / spec_commit read their parameters there; sent when it changes)
    if (!P->self_sent || std::memcmp(P->h_self.p, &p, sizeof(DevParams)) != 0) {
      std::memcpy(P->h_self.p, &p, sizeof(DevParams));
      STEREO_HIP_CHECK(hipMemcpy(P->d_self.p, P->h_self.p, sizeof(DevParams), hipMemcpyHostToDevice));
      P->self_sent = true;
    }
    for (int it = 0; it < iters; ++it) {
      issue_iteration(P, p, s);
      double lb = 0, en = 0;
      if (!collect_iteration(P, s, &lb, &en)) return fail(gave_up_text(P), err, errcap);
      P->lb = lb; P->energy = en; P->iterations += 1;
      if (done_iters) *done_iters += 1;
      const double rel_gap = (en - lb) / en;  // minimize.cpp:105
      if (rel_gap < max_relgap) {
        if (stopped) *stopped = 1;
        break;
      }
    }
    return 0;
  } catch (const HipError &e) {
    return fail(e.msg, err, errcap);
  }
}

// One fused launch for the strips of a group (what: as in launch_persistent).
static void launch_group(stereo_trws_plan *const *G, int n, int what, hipStream_t s) {
  stereo_trws_plan *P0 = G[0];
  GroupArgs ga{};
  ga.pp = P0->d_group.p; ga.n = n;
  int total = 0;
  const int epoch = P0->epoch + 1;
  for (int i = 0; i < n; ++i) {
    stereo_trws_plan *P = G[i];
    ++P->epoch;
    STEREO_HIP_CHECK(hipMemsetAsync(P->d_ctl.p, 0, sizeof(int32_t), s));  // ticket = 0
    ga.first[i] = total;
    total += (P0->wide || P0->fast2) ? std::min(P->grid_blocks, P->cus) : P->grid_blocks;
    if (what != 3) P->sweep_launches += 1;
  }
  ga.first[n] = total;
  if (P0->wide) launch_wide_group(P0->kernel, what, total, s, ga, epoch);
  else if (P0->fast2) launch_pipe2_group(P0->kernel, P0->pos != nullptr, what, total, s, ga, epoch);
  else launch_pipe_group(P0->kernel, P0->pos != nullptr, what, total, s, ga, epoch);
  STEREO_HIP_CHECK(hipGetLastError());
}

int stereo_trws_plans_issue(stereo_trws_plan *const *plans, int n, void *stream, char *err, size_t errcap) {
  DeviceScope device_scope_(plans && n > 0 && plans[0] ? plans[0]->device : -1);
  if (!plans || n < 1 || n > kMaxGroup) return fail("stereo_trws_plans_issue: need 1 .. 16 plans", err, errcap);
  for (int i = 0; i < n; ++i) {
    if (int rc = strip_ready(plans[i], "stereo_trws_plans_issue", err, errcap)) return rc;
    stereo_trws_plan *P = plans[i], *P0 = plans[0];
    if (P->issued) return fail("stereo_trws_plans_issue: the previous iteration has not been collected", err, errcap);
    if (P->device != P0->device || P->graph != P0->graph || P->K != P0->K || P->kernel != P0->kernel ||
        P->epoch != P0->epoch || P->fwd_pending != P0->fwd_pending || P->wide != P0->wide || P->fast != P0->fast || P->fast2 != P0->fast2 ||
        (P->pos == nullptr) != (P0->pos == nullptr) || P->mode != P0->mode)
      return fail("stereo_trws_plans_issue: the plans are not strips of one problem on one device in the same state", err, errcap);
    if (!(P->wide || P->fast || P->fast2))
      return fail("stereo_trws_plans_issue: strips run on the pipelined kernels only (K <= 64; K <= 128 with per-edge "
                  "positions; K <= 256 with shared ascending positions)", err, errcap);
  }
  try {
    stereo_trws_plan *P0 = plans[0];
    hipStream_t s = stream ? (hipStream_t)stream : P0->own_stream;
    if (!s) return fail("stereo_trws_plans_issue: a plain plan needs an explicit stream here", err, errcap);
    if (P0->d_group.n < (size_t)n) { P0->d_group.alloc(kMaxGroup); P0->h_group.alloc(kMaxGroup); }
    for (int i = 0; i < n; ++i) P0->h_group.p[i] = make_params(plans[i], false);   // (group launches keep the plain chain schedule)
    STEREO_HIP_CHECK(hipMemcpyAsync(P0->d_group.p, P0->h_group.p, sizeof(DevParams) * n, hipMemcpyHostToDevice, s));
    if (P0->time_sweeps) STEREO_HIP_CHECK(hipEventRecord(P0->ev0, s));
    if (!P0->fwd_pending) launch_group(plans, n, 0, s);
    launch_group(plans, n, 1, s);
    // the backward sweep's lower-bound terms travel while the next launch runs
    STEREO_HIP_CHECK(hipEventRecord(P0->ev_bwd, s));
    for (int i = 0; i < n; ++i) {
      stereo_trws_plan *P = plans[i];
      STEREO_HIP_CHECK(hipStreamWaitEvent(P->copy_stream, P0->ev_bwd, 0));
      STEREO_HIP_CHECK(hipMemcpyAsync(P->h_lb.p, P->d_lbterms.p, sizeof(double) * P->n_lb, hipMemcpyDeviceToHost, P->copy_stream));
      STEREO_HIP_CHECK(hipEventRecord(P->ev_lb, P->copy_stream));
      P->lb_in_flight = true;
    }
    launch_group(plans, n, 2, s);  // forward sweep of the NEXT iteration fused with this iteration's primal
    if (P0->time_sweeps) STEREO_HIP_CHECK(hipEventRecord(P0->ev1, s));
    for (int i = 0; i < n; ++i) {
      stereo_trws_plan *P = plans[i];
      P->fwd_pending = true;
      STEREO_HIP_CHECK(hipMemcpyAsync(P->h_en.p, P->d_eterms.p, sizeof(double) * P->n_en, hipMemcpyDeviceToHost, s));
      STEREO_HIP_CHECK(hipMemcpyAsync(P->h_ctl.p, P->d_ctl.p, kCtlWords * sizeof(int32_t), hipMemcpyDeviceToHost, s));
      P->issued = true; P->issue_stream = s; P->timed_by = P0;
    }
    return 0;
  } catch (const HipError &e) {
    return fail(e.msg, err, errcap);
  }
}

int stereo_trws_plan_issue(stereo_trws_plan *P, void *stream, char *err, size_t errcap) {
  return stereo_trws_plans_issue(&P, 1, stream, err, errcap);
}

int stereo_trws_plan_collect(stereo_trws_plan *P, double *lb_part, double *energy_part, char *err, size_t errcap) {
  DeviceScope device_scope_(P ? P->device : -1);
  if (!P) return fail("stereo_trws_plan_collect: NULL plan", err, errcap);
  if (!P->issued) return fail("stereo_trws_plan_collect: nothing was issued", err, errcap);
  try {
    double lb = 0, en = 0;
    if (!collect_iteration(P, P->issue_stream, &lb, &en)) return fail(gave_up_text(P), err, errcap);
    if (lb_part) *lb_part = lb;
    if (energy_part) *energy_part = en;
    return 0;
  } catch (const HipError &e) {
    return fail(e.msg, err, errcap);
  }
}

int stereo_trws_plan_commit(stereo_trws_plan *P, double lower_bound, double energy, char *err, size_t errcap) {
  DeviceScope device_scope_(P ? P->device : -1);
  if (!P) return fail("stereo_trws_plan_commit: NULL plan", err, errcap);
  P->lb = lower_bound; P->energy = energy; P->iterations += 1;
  return 0;
}

int stereo_trws_plan_connect(stereo_trws_plan *P, int which, stereo_trws_plan *peer, char *err, size_t errcap) {
  DeviceScope device_scope_(P ? P->device : -1);
  if (!P || !peer || (which != 0 && which != 1)) return fail("stereo_trws_plan_connect: bad argument", err, errcap);
  if (P->N != peer->N || P->E != peer->E || P->K != peer->K || P->nstrips != peer->nstrips ||
      peer->strip != P->strip + (which ? 1 : -1))
    return fail("stereo_trws_plan_connect: the peer is not the neighbouring strip of the same problem", err, errcap);
  try {
    if (peer->device != P->device) {  // one process driving several GPUs: map the neighbour's memory
      int can = 0;
      STEREO_HIP_CHECK(hipDeviceCanAccessPeer(&can, P->device, peer->device));
      if (!can) return fail("stereo_trws_plan_connect: no peer access between the two devices", err, errcap);
      int cur = 0;
      STEREO_HIP_CHECK(hipGetDevice(&cur));
      STEREO_HIP_CHECK(hipSetDevice(P->device));
      const hipError_t e = hipDeviceEnablePeerAccess(peer->device, 0);
      if (e != hipSuccess && e != hipErrorPeerAccessAlreadyEnabled) STEREO_HIP_CHECK(e);
      (void)hipGetLastError();
      STEREO_HIP_CHECK(hipSetDevice(cur));
    }
    P->peer_msg[which] = peer->d_msg.p; P->peer_done[which] = peer->d_done.p; P->peer_x[which] = peer->d_x.p;
    return 0;
  } catch (const HipError &e) {
    return fail(e.msg, err, errcap);
  }
}

int stereo_trws_plan_ipc_export(stereo_trws_plan *P, void *handles, size_t cap, char *err, size_t errcap) {
  DeviceScope device_scope_(P ? P->device : -1);
  if (!P || !handles) return fail("stereo_trws_plan_ipc_export: NULL argument", err, errcap);
  if (cap < STEREO_TRWS_IPC_BYTES) return fail("stereo_trws_plan_ipc_export: buffer smaller than STEREO_TRWS_IPC_BYTES", err, errcap);
  static_assert(3 * sizeof(hipIpcMemHandle_t) <= STEREO_TRWS_IPC_BYTES, "STEREO_TRWS_IPC_BYTES");
  try {
    hipIpcMemHandle_t h[3];
    STEREO_HIP_CHECK(hipIpcGetMemHandle(&h[0], P->d_msg.p));
    STEREO_HIP_CHECK(hipIpcGetMemHandle(&h[1], P->d_done.p));
    STEREO_HIP_CHECK(hipIpcGetMemHandle(&h[2], P->d_x.p));
    std::memset(handles, 0, STEREO_TRWS_IPC_BYTES);
    std::memcpy(handles, h, sizeof(h));
    return 0;
  } catch (const HipError &e) {
    return fail(e.msg, err, errcap);
  }
}

int stereo_trws_plan_ipc_connect(stereo_trws_plan *P, int which, const void *handles, char *err, size_t errcap) {
  DeviceScope device_scope_(P ? P->device : -1);
  if (!P || !handles || (which != 0 && which != 1)) return fail("stereo_trws_plan_ipc_connect: bad argument", err, errcap);
  try {
    hipIpcMemHandle_t h[3];
    std::memcpy(h, handles, sizeof(h));
    void *ptr[3] = {nullptr, nullptr, nullptr};
    for (int k = 0; k < 3; ++k) {
      if (P->ipc_mapped[which][k]) { (void)hipIpcCloseMemHandle(P->ipc_mapped[which][k]); P->ipc_mapped[which][k] = nullptr; }
      STEREO_HIP_CHECK(hipIpcOpenMemHandle(&ptr[k], h[k], hipIpcMemLazyEnablePeerAccess));
      P->ipc_mapped[which][k] = ptr[k];
    }
    P->peer_msg[which] = (double *)ptr[0]; P->peer_done[which] = (int32_t *)ptr[1]; P->peer_x[which] = (int32_t *)ptr[2];
    return 0;
  } catch (const HipError &e) {
    return fail(e.msg, err, errcap);
  }
}

int stereo_trws_plan_debug_flags(stereo_trws_plan *P, int32_t *done, int32_t *ctl) {
  DeviceScope device_scope_(P ? P->device : -1);
  if (!P) return 1;
  if (done && P->layout) {  // per global rank, like a plan of the whole problem (0 where the strip holds nothing)
    std::vector<int32_t> f(P->Nl);
    if (hipMemcpy(f.data(), P->d_done.p, sizeof(int32_t) * P->Nl, hipMemcpyDeviceToHost) != hipSuccess) return 1;
    std::fill(done, done + P->N, 0);
    for (int64_t i = 0; i < P->Nl; ++i) done[P->graph->rank[P->layout->nodes[i]]] = f[i];
  } else
  if (done && hipMemcpy(done, P->d_done.p, sizeof(int32_t) * P->N, hipMemcpyDeviceToHost) != hipSuccess) return 1;
  if (ctl && hipMemcpy(ctl, P->d_ctl.p, sizeof(int32_t) * 2, hipMemcpyDeviceToHost) != hipSuccess) return 1;
  return 0;
}

// Development aids: the lower-bound terms of the last backward sweep in the order the host sums them (rank N - 1 down to
// 0: the node's own term, then one per message it sent), and the message rows as they lie in HBM (E x K, edge-major).
int stereo_trws_plan_debug_terms(stereo_trws_plan *P, double *lb_terms, int64_t cap, int64_t *n_lb) {
  if (!P) return 1;
  if (n_lb) *n_lb = P->n_lb;
  if (lb_terms) std::memcpy(lb_terms, P->h_lb.p, sizeof(double) * (size_t)std::min<int64_t>(cap, P->n_lb));
  return 0;
}
int stereo_trws_plan_debug_messages(stereo_trws_plan *P, double *out, int64_t count) {
  DeviceScope device_scope_(P ? P->device : -1);
  if (!P || !out) return 1;
  if (hipDeviceSynchronize() != hipSuccess) return 1;
  const int64_t n = std::min<int64_t>(count, (int64_t)P->d_msg.n);
  return hipMemcpy(out, P->d_msg.p, sizeof(double) * (size_t)n, hipMemcpyDeviceToHost) == hipSuccess ? 0 : 1;
}

int stereo_trws_plan_strip_info(stereo_trws_plan *P, int *nstrips, int *strip, int64_t *own_nodes, int64_t *runs_forward,
                                int64_t *runs_backward, int *needs_previous, int *needs_next) {
  if (!P) return 1;
  if (nstrips) *nstrips = P->nstrips;
  if (strip) *strip = P->strip;
  if (own_nodes) *own_nodes = P->n_en;
  if (runs_forward) *runs_forward = P->nstrips > 1 ? P->ntickets[0] : (int64_t)P->graph->sweep[0].chain_run_ptr.size() - 1;
  if (runs_backward) *runs_backward = P->nstrips > 1 ? P->ntickets[1] : (int64_t)P->graph->sweep[1].chain_run_ptr.size() - 1;
  if (needs_previous) *needs_previous = P->need_peer[0] ? 1 : 0;
  if (needs_next) *needs_next = P->need_peer[1] ? 1 : 0;
  return 0;
}

int stereo_trws_plan_result(stereo_trws_plan *P, double *labelling, double *energy,
                            double *lower_bound, double *iterations, char *err, size_t errcap) {
  DeviceScope device_scope_(P ? P->device : -1);
  if (!P) return fail("stereo_trws_plan_result: NULL plan", err, errcap);
  try {
    if (labelling) {
      STEREO_HIP_CHECK(hipMemcpy(P->h_x.p, P->d_x.p, sizeof(int32_t) * P->Nl, hipMemcpyDeviceToHost));
      if (P->layout) {  // a strip: its own nodes and halo; label 1 elsewhere
        for (int64_t i = 0; i < P->N; ++i) labelling[i] = 1.0;
        for (int64_t i = 0; i < P->Nl; ++i) labelling[P->layout->nodes[i]] = (double)(P->h_x.p[i] + 1);
      } else
      for (int64_t i = 0; i < P->N; ++i) labelling[i] = (double)(P->h_x.p[i] + 1);  // trws_mex.cpp:137
    }
    if (energy) *energy = P->energy;
    if (lower_bound) *lower_bound = P->lb;
    if (iterations) *iterations = (double)P->iterations;
    return 0;
  } catch (const HipError &e) {
    return fail(e.msg, err, errcap);
  }
}

int stereo_trws_plan_info(stereo_trws_plan *P, int64_t *rank, int64_t *levels,
                          int64_t *max_level_nodes, char *err, size_t errcap) {
  if (!P) return fail("stereo_trws_plan_info: NULL plan", err, errcap);
  if (rank) for (int64_t i = 0; i < P->N; ++i) rank[i] = P->graph->rank[i];
  if (levels) *levels = (int64_t)P->graph->level_ptr.size() - 1;
  if (max_level_nodes) *max_level_nodes = P->graph->max_level_nodes;
  return 0;
}

int stereo_trws_plan_stats(stereo_trws_plan *P, double *sweep_ms, int64_t *sweep_launches, int reset) {
  DeviceScope device_scope_(P ? P->device : -1);
  if (!P) return 1;
  if (sweep_ms) *sweep_ms = P->sweep_ms;
  if (sweep_launches) *sweep_launches = P->sweep_launches;
  if (reset) { P->sweep_ms = 0; P->sweep_launches = 0; }
  P->time_sweeps = true;
  return 0;
}

int stereo_trws_plan_counters(stereo_trws_plan *P, int64_t *serial_messages, int reset) {
  DeviceScope device_scope_(P ? P->device : -1);
  if (!P) return 1;
  unsigned long long v = 0;
  if (hipMemcpy(&v, P->d_fallbacks.p, sizeof(v), hipMemcpyDeviceToHost) != hipSuccess) return 1;
  if (serial_messages) *serial_messages = (int64_t)v;
  if (reset && hipMemset(P->d_fallbacks.p, 0, sizeof(v)) != hipSuccess) return 1;
  return 0;
}

int stereo_trws_plan_spec_stats(stereo_trws_plan *P, int64_t out[4]) {
  DeviceScope device_scope_(P ? P->device : -1);
  if (!P || !out) return 1;
  out[0] = spec_active(P) ? 1 : 0; out[1] = out[2] = out[3] = 0;
  if (P->d_spec_stat.p) {
    unsigned long long v[32] = {0};
    if (hipMemcpy(v, P->d_spec_stat.p, sizeof(v), hipMemcpyDeviceToHost) != hipSuccess) return 1;
    if (std::getenv("STEREO_HIP_TRWS_TIMELINE"))
      std::fprintf(stderr, "[stereo_hip spec] last sweeps, roles done after (us): forward messages %.0f labels %.0f last loader %.0f publisher %.0f | backward messages %.0f "
                           "last loader %.0f publisher %.0f\n", v[8] / 100.0, v[9] / 100.0, v[10] / 100.0, v[11] / 100.0, v[12] / 100.0, v[14] / 100.0, v[15] / 100.0);
    if (false && v[8] && v[2])   // (-DSTEREO_HIP_RUNNER_PROFILE)
      std::fprintf(stderr, "[stereo_hip spec] message recurrence, cycles per visit: loop top %.0f | node in registers (incl. waits) %.0f | Di, next node asked for %.0f | "
                           "H, table, min H %.0f | window + row %.0f | publish, turn %.0f\n", (double)v[13] / v[2], (double)v[8] / v[2], (double)v[9] / v[2],
                   (double)v[10] / v[2], (double)v[11] / v[2], (double)v[12] / v[2]);
    out[1] = (int64_t)v[0]; out[2] = (int64_t)v[1]; out[3] = (int64_t)v[2];
    if (std::getenv("STEREO_HIP_TRWS_TIMELINE") && (v[16] || v[17]))   // (development, wide runner: where its roles wait, us in all)
      std::fprintf(stderr, "[stereo_hip spec] wide runner, us in all launches: meetings of the message waves forward %.0f backward %.0f | label wave waiting for its node %.0f | "
                           "loader 0: until the slot wait forward %.0f backward %.0f, slot wait %.0f / %.0f, staging %.0f / %.0f\n", v[16] / 100.0, v[17] / 100.0, v[18] / 100.0,
                   v[21] / 100.0, v[22] / 100.0, v[19] / 100.0, v[20] / 100.0, v[23] / 100.0, v[24] / 100.0);
    if (std::getenv("STEREO_HIP_TRWS_TIMELINE"))   // (development: how often, and for how long, the message recurrence found its next node not staged yet)
      std::fprintf(stderr, "[stereo_hip spec] runner visits %llu; the message recurrence found its node not staged yet: forward sweeps %llu times, %.1f us in all; "
                           "backward %llu times, %.1f us (incl. the wait for the rows in front of the chain)\n", v[2], v[3], (double)v[4] / 100.0, v[5], (double)v[6] / 100.0);
  }
  return 0;
}

int stereo_trws_messages(int kernel, int K, int64_t M, const double *Di, const double *gamma, const double *msg_in,
                         const double *q_source, const double *q_dest, const double *alpha, double lambda,
                         int certificate, int window, const double *shared_positions, double *msg_out,
                         double *vmin, int32_t *used_serial, char *err, size_t errcap) {
  if (kernel != 1 && kernel != 2) return fail("Unsupported kernel", err, errcap);
  if (K < 1 || K > kWave || M < 1) return fail("stereo_trws_messages: K must be in [1, 64], M >= 1", err, errcap);
  if (!Di || !gamma || !msg_in || !q_source || !q_dest || !alpha || !msg_out || !vmin)
    return fail("stereo_trws_messages: NULL argument", err, errcap);
  if (stereo_hip_device_count() < 1) return fail("stereo_trws_messages: no HIP device available", err, errcap);
  try {
    const size_t MK = (size_t)M * K;
    DevBuf<double> dD, dg, dm, dqs, dqd, da, dout, dv;
    DevBuf<uint16_t> dperm;
    DevBuf<int32_t> dser;
    DevBuf<unsigned long long> dfb;
    dD.upload(Di, MK); dg.upload(gamma, M); dm.upload(msg_in, MK); dqs.upload(q_source, MK); dqd.upload(q_dest, MK);
    da.upload(alpha, M); dout.alloc(MK); dv.alloc(M); dperm.alloc(MK); dser.alloc(M); dfb.alloc(4096);
    STEREO_HIP_CHECK(hipMemset(dfb.p, 0, sizeof(unsigned long long) * 4096));
    run_argsort(dqs.p, dperm.p, K, M, nullptr);
    fix_equal_positions(dqs.p, dperm.p, K, M);
    DevParams p{};
    p.K = K; p.Kp = (K + 1) & ~1; p.kernel = kernel; p.lambda = lambda; p.certificate = certificate ? 1 : 0;
    p.fallbacks = dfb.p;
    if (shared_positions) { p.pos_first = shared_positions[0]; p.pos_last = shared_positions[K - 1]; }
    // shared_positions: every message's q_source and q_dest ARE this vector (the caller's promise, as the
    // sweep kernels have it with fronto-parallel labels); strictly ascending ones take the compacted
    // certified loop of message_regs<.., true>
    bool shared_asc = false;
    if (shared_positions) {
      p.pos_gap = std::numeric_limits<double>::infinity();
      for (int k = 1; k < K; ++k) p.pos_gap = std::min(p.pos_gap, shared_positions[k] - shared_positions[k - 1]);
      shared_asc = K > 1 && p.pos_gap > 0 && std::isfinite(shared_positions[0]) && std::isfinite(shared_positions[K - 1]);
      if (!shared_asc && kernel == 1) p.pos_gap = 0;
    }
    if (const char *dbg = std::getenv("STEREO_HIP_TRWS_DEBUG")) p.debug = std::atoi(dbg);
    const unsigned grid = (unsigned)std::min<int64_t>(M, 4096);
#define STEREO_MSG_LAUNCH(KER, SH)                                                                                              \
    hipLaunchKernelGGL((trws_messages_kernel<KER, SH>), dim3(grid), dim3(kWave), 0, 0, p, K, M, dD.p, dg.p, dm.p, dqs.p, dqd.p, \
                       da.p, dperm.p, shared_positions ? window : -1, dout.p, dv.p, dser.p, dfb.p)
    if (kernel == 1) { if (shared_asc) STEREO_MSG_LAUNCH(1, true); else STEREO_MSG_LAUNCH(1, false); }
    else STEREO_MSG_LAUNCH(2, false);
#undef STEREO_MSG_LAUNCH
    STEREO_HIP_CHECK(hipGetLastError());
    STEREO_HIP_CHECK(hipDeviceSynchronize());
    STEREO_HIP_CHECK(hipMemcpy(msg_out, dout.p, sizeof(double) * MK, hipMemcpyDeviceToHost));
    STEREO_HIP_CHECK(hipMemcpy(vmin, dv.p, sizeof(double) * M, hipMemcpyDeviceToHost));
    if (used_serial) STEREO_HIP_CHECK(hipMemcpy(used_serial, dser.p, sizeof(int32_t) * M, hipMemcpyDeviceToHost));
    return 0;
  } catch (const HipError &e) {
    return fail(e.msg, err, errcap);
  }
}

int stereo_trws_plan_path(stereo_trws_plan *P) {
  if (!P) return -1;
  return P->wide ? 3 : P->fast2 ? 4 : P->fast ? 2 : 1;
}

}  // extern "C"

// ---- the gateway entry: what trws_mex reaches ------------------------------------------------------------
// A simultaneous fusion through trws.m:33 calls the gateway once per move with the SAME connectivity
// (dispmap_super.m:153-198: the neighbourhood of the object), and fronto-parallel proposals make every column
// of q and qprim one and the same vector (:177-183 evaluates each plane at every edge: [0 0 1 -d] gives d).
// So the gateway (i) keeps the plan of the last (kernel, K, N, E, connectivity, message mode, device) -- graph
// analysis, descriptors and device buffers survive the call -- and (ii) looks at q / qprim on the host before
// uploading anything: if all 2 E columns are bitwise one vector, that vector goes up as the plan's shared
// positions (8 K bytes instead of 16 K E) and the shared-position kernels run; results are the K x E form's bit
// for bit (tests/test_trws_gpu.py).  STEREO_HIP_TRWS_CACHE=0: a plan per call, K x E arrays always uploaded.
namespace {

// The environment switches a plan freezes at creation: part of the cache key (a cached plan must not outlive them).
std::string trws_env_key() {
  std::string k;
  for (const char *name : {"STEREO_HIP_GPUS", "STEREO_HIP_TRWS_CERTIFICATE", "STEREO_HIP_TRWS_FAST", "STEREO_HIP_TRWS_SPIN_SECONDS", "STEREO_HIP_TRWS_PROF",
                           "STEREO_HIP_TRWS_TIMELINE", "STEREO_HIP_TRWS_SPEC", "STEREO_HIP_TRWS_SPEC_SEG", "STEREO_HIP_STRIPS_FINEGRAINED"}) {
    const char *v = std::getenv(name);
    k += v ? v : "-";
    k += '|';
  }
  return k;
}

struct TrwsStripSet;
struct TrwsPlanCache {
  std::mutex mu;
  stereo_trws_plan *plan = nullptr;
  TrwsStripSet *strips = nullptr;   // ... or the row strips of the problem (STEREO_HIP_GPUS)
  std::string env;
  int kernel = 0, K = 0, mode = 0, device = -1;
  int64_t N = 0, E = 0;
  std::vector<uint32_t> conn;
};

TrwsPlanCache &trws_plan_cache() {
  static TrwsPlanCache *C = new TrwsPlanCache;   // (never destroyed: the HIP runtime may be gone before static destructors run)
  return *C;
}

// true iff every column of q and of qprim (K x E, column-major) equals q's first column bit for bit
bool columns_are_one_vector(const double *q, const double *qprim, int K, int64_t E) {
  if (E < 1) return false;
  const size_t row = sizeof(double) * (size_t)K;
  if (std::memcmp(q, qprim, row) != 0) return false;
  // a quick look at a few columns first: general planes differ on the first edge already
  for (int64_t e : {E / 2, E - 1})
    if (std::memcmp(q, q + (size_t)e * K, row) != 0 || std::memcmp(q, qprim + (size_t)e * K, row) != 0) return false;
  unsigned nt = std::thread::hardware_concurrency();
  nt = std::max(1u, std::min(nt ? nt : 1u, 16u));
  if ((size_t)E * K < (1u << 20)) nt = 1;
  std::vector<char> same(nt, 1);
  auto scan = [&](unsigned t) {
    const int64_t a = E * t / nt, b = E * (t + 1) / nt;
    for (int64_t e = a; e < b; ++e)
      if (std::memcmp(q, q + (size_t)e * K, row) != 0 || std::memcmp(q, qprim + (size_t)e * K, row) != 0) { same[t] = 0; return; }
  };
  std::vector<std::thread> th;
  for (unsigned t = 1; t < nt; ++t) th.emplace_back(scan, t);
  scan(0);
  for (auto &x : th) x.join();
  for (char c : same) if (!c) return false;
  return true;
}

int trws_solve_on(stereo_trws_plan *P, const double *unary, const double *q, const double *qprim, const double *alphas,
                  double tol, double maxiter, double max_relgap, bool look_for_shared, double *labelling, double *energy,
                  double *lower_bound, double *iterations, char *err, size_t errcap) {
  int rc;
  if (look_for_shared && columns_are_one_vector(q, qprim, P->K, P->E))
    rc = stereo_trws_plan_upload(P, unary, nullptr, nullptr, q, alphas, tol, err, errcap);
  else
    rc = stereo_trws_plan_upload(P, unary, q, qprim, nullptr, alphas, tol, err, errcap);
  if (rc) return rc;
  // Minimize_TRW_S always runs at least one iteration (minimize.cpp:31,100-101)
  int itmax = (int)maxiter;  // trws_mex.cpp:125
  if (itmax < 1) itmax = 1;
  rc = stereo_trws_plan_iterate(P, itmax, max_relgap, nullptr, nullptr, nullptr, err, errcap);
  if (rc) return rc;
  return stereo_trws_plan_result(P, labelling, energy, lower_bound, iterations, err, errcap);
}

// ---- the gateway on several devices (STEREO_HIP_GPUS = G): row strips of the image grid ---------------------------
// trws_mex hands over a graph, not an image; the image grid of dispmap_super.m:279-302 is recognised from it: nodes
// col * H + row (:281-282), every edge joins vertical (|a - b| == 1, same column) or horizontal (|a - b| == H)
// neighbours.  Band g of the rows goes to strip g; strip g runs on device g when the process sees at least G devices
// (peer access, stereo_trws_plan_connect), otherwise all strips share the current device as logical strips (one fused
// launch per sweep) -- the same kernels, the same hand-over protocol, the same bits.  Anything else (another graph, a
// label count the strip kernels do not take) stays on one device.
int64_t image_grid_height(int64_t N, int64_t E, const uint32_t *conn) {
  int64_t H = 0;
  for (int64_t e = 0; e < E; ++e) {
    const int64_t a = conn[2 * e], b = conn[2 * e + 1];
    const int64_t d = a > b ? a - b : b - a;
    if (d == 1) continue;
    if (H == 0) H = d;
    if (d != H) return 0;
  }
  if (H < 2 || N % H != 0) return 0;
  for (int64_t e = 0; e < E; ++e) {   // vertical edges stay inside a column
    const int64_t a = conn[2 * e], b = conn[2 * e + 1];
    if ((a > b ? a - b : b - a) == 1 && a / H != b / H) return 0;
  }
  return H;
}

// The terms of the bound and of the energy in the order ONE plan adds them (rank N - 1 down to 0: the node's own term, then
// one per message; rank 0 up: one per node), as runs of consecutive terms of one strip: a strip numbers its terms in that
// same order, so a run is as long as consecutive ranks stay with one owner (a band of rows: a few runs per image row at
// most).  Summed this way the gateway's two scalars are the single plan's to the last bit -- and with them the stop test
// and the iteration count (minimize.cpp:105).
struct TermRuns {
  std::vector<int32_t> strip;
  std::vector<int64_t> start, count;
  void add(int s, int64_t pos, int64_t n) {
    if (!strip.empty() && strip.back() == s && start.back() + count.back() == pos) { count.back() += n; return; }
    strip.push_back(s); start.push_back(pos); count.push_back(n);
  }
};

struct TrwsStripSet {
  std::vector<stereo_trws_plan *> plans;
  std::vector<int32_t> owner;
  TermRuns lb_runs, en_runs;
  bool one_device = true;
  ~TrwsStripSet() { for (stereo_trws_plan *P : plans) if (P) stereo_trws_plan_destroy(P); }
};

int strips_create(int kernel, int K, int64_t N, int64_t E, const uint32_t *conn, int mode, int G, int64_t H, TrwsStripSet &S, char *err,
                  size_t errcap) {
  S.owner.resize(N);
  for (int64_t i = 0; i < N; ++i) S.owner[i] = (int32_t)std::min<int64_t>((i % H) * G / H, G - 1);
  const int ndev = stereo_hip_device_count();
  S.one_device = ndev < G;
  int home = 0;
  if (hipGetDevice(&home) != hipSuccess) return fail("stereo_trws: no HIP device available (the HIP path has no CPU fallback)", err, errcap);
  int cus = 256;
  (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, home);
  const int per_strip = S.one_device ? std::max(2, cus / G) : 0;   // logical strips must all be resident together
  S.plans.assign(G, nullptr);
  int rc = 0;
  for (int g = 0; g < G && !rc; ++g) {
    if (!S.one_device && hipSetDevice(g) != hipSuccess) rc = fail("stereo_trws: hipSetDevice failed", err, errcap);
    if (!rc) rc = stereo_trws_plan_create_strip(kernel, K, N, E, conn, mode, g == 0 ? S.owner.data() : nullptr, G, g, per_strip,
                                                g ? S.plans[0] : nullptr, &S.plans[g], err, errcap);
  }
  (void)hipSetDevice(home);
  for (int g = 0; g < G && !rc; ++g) {
    if (g > 0) rc = stereo_trws_plan_connect(S.plans[g], 0, S.plans[g - 1], err, errcap);
    if (!rc && g + 1 < G) rc = stereo_trws_plan_connect(S.plans[g], 1, S.plans[g + 1], err, errcap);
  }
  if (!rc) {
    const TrwsGraph &g = *S.plans[0]->graph;
    for (int64_t r = N - 1; r >= 0; --r) {
      const int s = S.owner[g.order[r]];
      S.lb_runs.add(s, g.lb_pos_node[r], 1);
      for (int32_t k = g.bptr[r]; k < g.bptr[r + 1]; ++k) S.lb_runs.add(s, g.lb_pos_edge[g.bidx[k]], 1);
    }
    for (int64_t r = 0; r < N; ++r) S.en_runs.add(S.owner[g.order[r]], g.e_pos[r], 1);
  }
  return rc;
}

int strips_solve(TrwsStripSet &S, const double *unary, const double *q, const double *qprim, const double *alphas, double tol,
                 double maxiter, double max_relgap, double *labelling, double *energy, double *lower_bound, double *iterations, char *err,
                 size_t errcap) {
  const int G = (int)S.plans.size();
  stereo_trws_plan *P0 = S.plans[0];
  const bool shared = columns_are_one_vector(q, qprim, P0->K, P0->E);
  for (int g = 0; g < G; ++g) {
    const int rc = shared ? stereo_trws_plan_upload(S.plans[g], unary, nullptr, nullptr, q, alphas, tol, err, errcap)
                          : stereo_trws_plan_upload(S.plans[g], unary, q, qprim, nullptr, alphas, tol, err, errcap);
    if (rc) return rc;
  }
  int itmax = (int)maxiter;  // trws_mex.cpp:125; Minimize_TRW_S always runs at least one iteration (minimize.cpp:31,100-101)
  if (itmax < 1) itmax = 1;
  double lb = 0, en = 0;
  int done = 0;
  for (int it = 0; it < itmax; ++it) {
    int rc = 0;
    if (S.one_device) rc = stereo_trws_plans_issue(S.plans.data(), G, nullptr, err, errcap);
    else for (int g = 0; g < G && !rc; ++g) rc = stereo_trws_plan_issue(S.plans[g], nullptr, err, errcap);
    if (rc) return rc;
    // every strip's terms are on the host behind its collect; they are added in the order one plan adds them (TermRuns)
    for (int g = 0; g < G; ++g)
      if ((rc = stereo_trws_plan_collect(S.plans[g], nullptr, nullptr, err, errcap)) != 0) return rc;
    lb = 0; en = 0;
    for (size_t k = 0; k < S.lb_runs.strip.size(); ++k) {
      const double *t = S.plans[S.lb_runs.strip[k]]->h_lb.p + S.lb_runs.start[k];
      for (int64_t i = 0; i < S.lb_runs.count[k]; ++i) lb += t[i];
    }
    for (size_t k = 0; k < S.en_runs.strip.size(); ++k) {
      const double *t = S.plans[S.en_runs.strip[k]]->h_en.p + S.en_runs.start[k];
      for (int64_t i = 0; i < S.en_runs.count[k]; ++i) en += t[i];
    }
    for (int g = 0; g < G; ++g) (void)stereo_trws_plan_commit(S.plans[g], lb, en, err, errcap);
    ++done;
    if ((en - lb) / en < max_relgap) break;  // minimize.cpp:105
  }
  std::vector<double> part((size_t)P0->N);
  for (int g = 0; g < G; ++g) {
    const int rc = stereo_trws_plan_result(S.plans[g], part.data(), nullptr, nullptr, nullptr, err, errcap);
    if (rc) return rc;
    for (int64_t i = 0; i < P0->N; ++i)
      if (S.owner[i] == g) labelling[i] = part[i];
  }
  *energy = en; *lower_bound = lb; *iterations = (double)done;
  return 0;
}

thread_local int g_last_gateway_strips = 0;   // what the last stereo_trws call of this thread ran on (stereo_trws_gateway_strips)

// how many strips the gateway should cut the problem into (1: the plain single-device plan)
int gateway_strips(int K, int64_t N, int64_t E, const uint32_t *conn, const double *q, const double *qprim, int mode, int64_t *H_out) {
  const char *ge = std::getenv("STEREO_HIP_GPUS");
  const int G = ge ? std::atoi(ge) : 1;
  if (G < 2 || G > kMaxGroup || mode != STEREO_TRWS_MESSAGES_EXACT || K > 4 * kWave) return 1;
  // (the strip kernels take K <= 128 with any positions, up to 256 labels with one shared positions vector)
  if (K > 2 * kWave && !columns_are_one_vector(q, qprim, K, E)) return 1;
  const int64_t H = image_grid_height(N, E, conn);
  if (H < 2 * G) return 1;
  *H_out = H;
  return G;
}

}  // namespace

extern "C" {

int stereo_trws_gateway_strips(void) { return g_last_gateway_strips; }

void stereo_trws_cache_clear(void) {
  TrwsPlanCache &C = trws_plan_cache();
  std::lock_guard<std::mutex> lock(C.mu);
  if (C.plan) { stereo_trws_plan_destroy(C.plan); C.plan = nullptr; }
  if (C.strips) { delete C.strips; C.strips = nullptr; }
  C.conn.clear(); C.conn.shrink_to_fit();
}

int stereo_trws(int kernel, const double *unary, const uint32_t *conn, const double *q,
                const double *qprim, const double *alphas, double tol, double maxiter,
                double max_relgap, int K, int64_t N, int64_t E, double *labelling, double *energy,
                double *lower_bound, double *iterations, char *err, size_t errcap) {
  if (kernel != 1 && kernel != 2) return fail("Unsupported kernel", err, errcap);  // trws_mex.cpp:162
  if (!unary || !conn || !q || !qprim || !alphas || !labelling || !energy || !lower_bound || !iterations)
    return fail("stereo_trws: NULL argument", err, errcap);
  int mode = STEREO_TRWS_MESSAGES_EXACT;
  if (const char *m = std::getenv("STEREO_HIP_TRWS_MESSAGES"))
    if (std::string(m) == "minplus") mode = STEREO_TRWS_MESSAGES_MINPLUS;
  const char *ce = std::getenv("STEREO_HIP_TRWS_CACHE");
  const bool cached = (!ce || std::atoi(ce) != 0) && E > 0 && N > 0;
  int64_t gridH = 0;
  const int G = (E > 0 && N > 0) ? gateway_strips(K, N, E, conn, q, qprim, mode, &gridH) : 1;
  g_last_gateway_strips = G;
  if (G > 1 && !cached) {
    TrwsStripSet S;
    int rc = strips_create(kernel, K, N, E, conn, mode, G, gridH, S, err, errcap);
    if (!rc) rc = strips_solve(S, unary, q, qprim, alphas, tol, maxiter, max_relgap, labelling, energy, lower_bound, iterations, err, errcap);
    return rc;
  }
  if (!cached) {
    stereo_trws_plan *P = nullptr;
    int rc = stereo_trws_plan_create(kernel, K, N, E, conn, mode, &P, err, errcap);
    if (rc) return rc;
    rc = trws_solve_on(P, unary, q, qprim, alphas, tol, maxiter, max_relgap, false, labelling, energy, lower_bound, iterations,
                       err, errcap);
    stereo_trws_plan_destroy(P);
    return rc;
  }
  TrwsPlanCache &C = trws_plan_cache();
  std::lock_guard<std::mutex> lock(C.mu);
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) return fail("stereo_trws: no HIP device available (the HIP path has no CPU fallback)", err, errcap);
  const std::string env = trws_env_key();
  const bool hit = (C.plan || C.strips) && C.kernel == kernel && C.K == K && C.N == N && C.E == E && C.mode == mode && C.device == dev && C.env == env &&
                   std::memcmp(C.conn.data(), conn, sizeof(uint32_t) * 2 * (size_t)E) == 0;
  if (!hit) {
    if (C.plan) { stereo_trws_plan_destroy(C.plan); C.plan = nullptr; }
    if (C.strips) { delete C.strips; C.strips = nullptr; }
    if (G > 1) {
      TrwsStripSet *S = new TrwsStripSet;
      const int rc = strips_create(kernel, K, N, E, conn, mode, G, gridH, *S, err, errcap);
      if (rc) { delete S; return rc; }
      C.strips = S; C.kernel = kernel; C.K = K; C.N = N; C.E = E; C.mode = mode; C.device = dev; C.env = env;
      C.conn.assign(conn, conn + 2 * (size_t)E);
    }
  }
  if (C.strips) {
    const int rc = strips_solve(*C.strips, unary, q, qprim, alphas, tol, maxiter, max_relgap, labelling, energy, lower_bound, iterations, err, errcap);
    if (rc) { delete C.strips; C.strips = nullptr; }   // never keep plans an error went through
    return rc;
  }
  if (!hit) {
    stereo_trws_plan *P = nullptr;
    const int rc = stereo_trws_plan_create(kernel, K, N, E, conn, mode, &P, err, errcap);
    if (rc) return rc;
    C.plan = P; C.kernel = kernel; C.K = K; C.N = N; C.E = E; C.mode = mode; C.device = dev; C.env = env;
    C.conn.assign(conn, conn + 2 * (size_t)E);
  }
  const int rc = trws_solve_on(C.plan, unary, q, qprim, alphas, tol, maxiter, max_relgap, true, labelling, energy, lower_bound,
                               iterations, err, errcap);
  if (rc) { stereo_trws_plan_destroy(C.plan); C.plan = nullptr; }   // never keep a plan an error went through
  return rc;
}

}  // extern "C"

