// TRW-S pipelined sweep kernel for K <= 64 (both smoothness kernels): role-specialised waves, one
// barrier per visit.  Part of libstereo_hip.so; overview in trws_plan.hip.
#include <hip/hip_runtime.h>

#include <cstdint>

#include "../../include/stereo_hip.h"
#include "common.h"
#include "trws_dev.h"
#include "trws_launch.h"

namespace stereo {
namespace {

// ---- pipelined persistent sweep (K <= 64): role-specialised waves ---------------------
// Same dataflow schedule and arithmetic as trws_persistent_kernel, but the global-memory traffic
// of a visit is taken off the critical path by dedicated waves of the workgroup:
//   waves 0-7  compute: read the staged node from LDS, form Di, compute outgoing message
//              `wave` in registers, hand it over in LDS
//   wave 8     loader:  while node i is computed, decodes the descriptor of node i+1, waits
//              for its foreign completion flags, fetches unary / messages / weights /
//              positions / neighbour labels and stages them in LDS
//   wave 9     storer:  while node i is computed, writes node i-1's new messages and scalars
//              to HBM (write-through), drains, raises node i-1's completion flag
//   wave 11    primal:  labelling + energy term of node i (previous iteration's primal pass);
//              wave 10 idles so that the primal wave shares a SIMD with one compute wave only
// One s_barrier per visit.  The compute waves never touch global memory, so no load or
// store latency is ever exposed on the chain of dependent visits.
template <int KERNEL, bool BACKWARD, bool PRIMAL, bool UPDATE, bool SHARED>
__device__ __forceinline__ void pipe_body(DevParams p, int epoch) {
  extern __shared__ __attribute__((aligned(16))) double lds[];
  double *stage0 = lds;                                   // 2 stages
  double *hand = lds + 2 * kStageDoubles;                 // ring of 4 x 8 x 64: the last visits' new messages
  double *scal = hand + 4 * 8 * kWave;                    // 2 x kScalDoubles
  double *hqtab = scal + 2 * kScalDoubles;                // per compute wave: 64 x (h, q, u, v)
  int *ctl = (int *)(hqtab + kPipeCompute * kPipeTab);    // [0] run, [1] abort
  int *dring = ctl + 4;                                   // the last three descriptors (the storer's comes from here, not from HBM)
  double *zrow = (double *)(dring + 3 * kWave);           // 64 zeros: where the Di loop finds the message rows a node does not have
  const int K = p.K;
  const double inf = __builtin_huge_val();
  const int tid = threadIdx.x, lane = tid & (kWave - 1);
  const int wave = __builtin_amdgcn_readfirstlane(tid / kWave);
  constexpr int D = BACKWARD ? 1 : 0;
  constexpr int DW = TrwsGraph::kDescWords;
  const int32_t *desc = p.desc[D];
  const bool act = lane < K;
  const double posk = (SHARED && act) ? p.pos[lane] : 0.0;
  int look_streak = 0;  // failed second looks in a row (this compute wave): see message_regs
  const int perm_shared = (SHARED && wave < kPipeCompute) ? (act ? (int)p.perm_pos[lane] : lane) : -1;  // source order by position, once
  if (tid == 0) ctl[1] = 0;
  if (tid < kWave) zrow[tid] = 0.0;
  if (wave < kPipeCompute && lane < 2 * kPipePad) {
    // padding of the source tables (entries -16 .. -1 and 64 .. 79): never overwritten afterwards
    double *e = hqtab + wave * kPipeTab + 4 * (lane < kPipePad ? lane : kWave + lane);
    e[0] = inf; e[1] = 0; e[2] = 0; e[3] = 0;
  }
  if ((p.debug & 2) && !BACKWARD) p.prof = nullptr;  // profile backward sweeps only
  if ((p.debug & 4) && BACKWARD) p.prof = nullptr;   // profile forward sweeps only

  for (;;) {
    if (tid == 0) { const int t_ = atomicAdd(p.ticket, 1); ctl[0] = t_ < p.ntickets[D] ? (p.run_order[D] ? p.run_order[D][t_] : t_) : p.nruns[D]; }
    __syncthreads();
    const int run = __builtin_amdgcn_readfirstlane(ctl[0]);
    __syncthreads();
    if (run >= p.nruns[D]) break;
    const int p0 = p.run_ptr[D][run], p1 = p.run_ptr[D][run + 1];
    int xprev = 0, xprev2 = 0;  // primal wave: labels of the previous two nodes of the run
    int wnext = 0;              // loader: raw descriptor word of the node after next (prefetched)
    if (wave == kPipeCompute) wnext = desc[(size_t)p0 * DW + lane];
    // loader A (own data, two visits deep): descriptors of the next three nodes and the parked requests
    int wa1 = 0, wa2 = 0, wa3 = 0;
    double rdk = 0, rmv[8] = {0, 0, 0, 0, 0, 0, 0, 0}, rqv[8] = {0, 0, 0, 0, 0, 0, 0, 0}, rqpv[8] = {0, 0, 0, 0, 0, 0, 0, 0}, rav = 0;
#define PIPE_LOAD8(DST, PTR) DST = *(PTR)   /* plain loads: the compiler keeps them in flight across the barrier and waits at the first use */
#define PIPE_REQUEST_OWN(W)                                                                          \
    do {                                                                                             \
      const NodeDesc rq = decode_desc(W);                                                            \
      const int rtot = rq.nout + rq.nin;                                                             \
      if (act) { const double *a_ = p.unary + (size_t)rq.node * K + lane; PIPE_LOAD8(rdk, a_); }     \
      _Pragma("unroll") for (int j = 0; j < 8; ++j) {                                                \
        if (j < rtot && act) {                                                                       \
          const size_t off = (size_t)rq.e[j] * K + lane;                                             \
          if (j < rq.nout && (UPDATE || PRIMAL)) { const double *a_ = p.msg + off; PIPE_LOAD8(rmv[j], a_); } \
          if (!SHARED) {                                                                             \
            const double *a_ = p.q + off, *b_ = p.qprim + off;                                       \
            PIPE_LOAD8(rqv[j], a_); PIPE_LOAD8(rqpv[j], b_);                                         \
          }                                                                                          \
        }                                                                                            \
      }                                                                                              \
      rav = 0;                                                                                       \
      if (lane < rtot) {                                                                             \
        int ej = 0;                                                                                  \
        _Pragma("unroll") for (int j = 0; j < 8; ++j)                                                \
          if (lane == j) ej = rq.e[j];                                                               \
        rav = p.alpha[ej];                                                                           \
      }                                                                                              \
    } while (0)
    if (wave == kPipeCompute + 2) {
      wa1 = desc[(size_t)p0 * DW + lane];
      if (p0 + 1 < p1) wa2 = desc[(size_t)(p0 + 1) * DW + lane];
      if (p0 + 2 < p1) wa3 = desc[(size_t)(p0 + 2) * DW + lane];
      PIPE_REQUEST_OWN(wa1);
    }
    if (p.timeline && tid == 0) p.timeline[((size_t)D * p.nruns[0] + run) * 2] = wall_clock64();
    unsigned long long busy = 0;
#ifdef STEREO_HIP_VISIT_PROFILE
    unsigned long long vacc[6] = {0, 0, 0, 0, 0, 0}, macc[5] = {0, 0, 0, 0, 0}, lacc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#define VSTAMP(i) do { const long long n_ = (long long)__builtin_readcyclecounter(); vacc[i] += (unsigned long long)(n_ - vmark); vmark = n_; } while (0)
#else
#define VSTAMP(i) do { } while (0)
#endif

    for (int pos = p0 - 1; pos <= p1; ++pos) {
      const long long tstart = p.prof ? (long long)__builtin_readcyclecounter() : 0;
#ifdef STEREO_HIP_VISIT_PROFILE
      long long vmark = (long long)__builtin_readcyclecounter();
#endif
      double *st = stage0 + (pos & 1) * kStageDoubles;          // node `pos`
      double *stn = stage0 + ((pos + 1) & 1) * kStageDoubles;   // node `pos + 1`
      double *hcur = hand + (pos & 3) * 8 * kWave, *hprev = hand + ((pos - 1) & 3) * 8 * kWave;
      double *sc = scal + (pos & 1) * kScalDoubles;
      const bool have_node = pos >= p0 && pos < p1;

      if (wave < kPipeCompute) {
        // ------------------------------------------------------------ compute
        if (UPDATE && have_node) {
          const int *sti = (const int *)(st + kStI);
          const int f = __builtin_amdgcn_readfirstlane(sti[2]);
          const int nout = f & 15, md = (f >> 16) & 255;
          const int myrow = sti[72 + (lane & 7)];  // LDS offsets (doubles) of the node's message rows, from the loader
          VSTAMP(0);
          if (wave < nout || (BACKWARD && wave == 0)) {  // waves without a message stay out of the way
          double Di = act ? st[kStD + lane] : 0.0;
          // (this wave's own old message is one of the rows; it is read once more below rather than
          //  picked out of the loop with eight selects)
          const double mown = st[kStM + (wave < nout ? wave : 0) * kWave + lane];
          // (all eight rows are requested together and added in list order; a row the node does not have
          //  is the zero row -- x + 0.0 == x --, so nothing here branches on the node's degree and the
          //  reads share one LDS latency instead of paying one each)
          {
            double rowv[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) rowv[j] = lds[__builtin_amdgcn_readlane(myrow, j) + lane];
#pragma unroll
            for (int j = 0; j < 8; ++j) Di += rowv[j];
          }
          double node_vmin = 0;
          if (BACKWARD) {
            node_vmin = wave_min_dpp(act ? Di : inf);
            Di -= node_vmin;
            if (tid == 0) sc[8] = node_vmin;
          }
          VSTAMP(1);
          {
            const int j = wave;  // one compute wave per outgoing message
            if (j < nout) {
              const double gamma = st[kStG];  // (double)1 / (double)max(n_out, n_in), MRFEnergy.cpp:207-228
              const double h = act ? gamma * Di - mown : inf;
              const bool src_is_qprim = ((BACKWARD ? 1 : 0) == ((md >> j) & 1));
              double qsrc = posk, qdst = posk;
              const uint16_t *perm = p.perm_pos;
              if (!SHARED) {
                const double a_ = act ? st[kStQ + j * kWave + lane] : 0.0;
                const double b_ = act ? st[kStQP + j * kWave + lane] : 0.0;
                qsrc = src_is_qprim ? b_ : a_;
                qdst = src_is_qprim ? a_ : b_;
                const int e = __builtin_amdgcn_readfirstlane(sti[4 + j]);
                perm = (src_is_qprim ? p.perm_qp : p.perm_q) + (size_t)e * K;
              }
              const double alpha = st[kStA + j];
              VSTAMP(2);
              double newm = 0;
              const double v = message_regs<KERNEL, SHARED>(p, K, alpha, h, qsrc, qdst, perm, newm, lane,
                                                    hqtab + wave * kPipeTab + 4 * kPipePad, (SHARED && p.win_ok) ? p.window : -1, &look_streak
#ifdef STEREO_HIP_VISIT_PROFILE
                                                    , wave == 0 ? macc : nullptr
#else
                                                    , nullptr
#endif
                                                    , perm_shared);
              VSTAMP(3);
              if (act) hcur[j * kWave + lane] = newm;
              if (BACKWARD && lane == 0) sc[j] = v;
            }
          }
          }
        }
      } else if (wave == kPipeCompute) {
        // ------------------------------------------------------------ loader: stage node pos + 1
        if (pos + 1 >= p0 && pos + 1 < p1) {
          const int w = wnext;
          if (pos + 2 < p1) wnext = desc[(size_t)(pos + 2) * DW + lane];
          const NodeDesc nx = decode_desc(w);
          int *stni = (int *)(stn + kStI);
          stni[lane] = w;
          dring[((pos + 1) % 3) * kWave + lane] = w;
          {
            // where the compute waves find the node's message rows at visit pos + 1: a message handed
            // over inside the run sits in the ring of the last two visits, everything else in this
            // stage -- decided once here instead of by every compute wave in scalar code
            int slr = -1;
#pragma unroll
            for (int j = 0; j < 8; ++j)
              if (lane == j) slr = j >= nx.nout ? nx.slot[j] : -1;
            const int row = lane >= nx.nout + nx.nin ? (int)(zrow - lds)
                          : slr >= 8 ? (int)(hand - lds) + ((pos - 1) & 3) * 8 * kWave + (slr - 8) * kWave
                          : slr >= 0 ? (int)(hand - lds) + (pos & 3) * 8 * kWave + slr * kWave
                                     : (int)(stn - lds) + kStM + lane * kWave;
            if (lane < 8) stni[72 + lane] = row;
          }
          const int ntot = nx.nout + nx.nin;
          // (everything that does not depend on other workgroups -- unary, previous-sweep messages,
          //  positions, weights -- is loader A's, below, two visits ahead)
          double mv[8];
#pragma unroll
          for (int j = 0; j < 8; ++j) mv[j] = 0;
          int pxv = 0, xn = 0, sl = 0;
          if (lane < ntot) {
#pragma unroll
            for (int j = 0; j < 8; ++j)
              if (lane == j) { xn = nx.xn[j]; sl = nx.slot[j]; }
          }
          // the completion flags of the foreign neighbours, then their data
          // (all flags are polled together: lane j watches dependency j)
#ifdef STEREO_HIP_VISIT_PROFILE
          const long long lb0 = (long long)__builtin_readcyclecounter();
#endif
          wait_for_dependencies(p, nx.ndep, nx.dep[0], nx.dep[1], nx.dep[2], nx.dep[3], nx.rank, epoch, lane, ctl + 1);
#ifdef STEREO_HIP_VISIT_PROFILE
          const long long lb1 = (long long)__builtin_readcyclecounter();
#endif
#pragma unroll
          for (int j = 0; j < 8; ++j)
            if (UPDATE && j >= nx.nout && j < ntot && nx.slot[j] < 0 && act)
              mv[j] = ld_sc1(p.msg + (size_t)nx.e[j] * K + lane);
          if (PRIMAL && lane < ntot && lane >= nx.nout && sl < 0) pxv = ld_sc1(p.x + xn);
#pragma unroll
          for (int j = 0; j < 8; ++j)
            if (j >= nx.nout && j < ntot && act) stn[kStM + j * kWave + lane] = mv[j];
          if (lane < 8) stni[64 + lane] = pxv;
#ifdef STEREO_HIP_VISIT_PROFILE
          if (pos > p0 + 3) {  // steady state only: the first visits of a run wait for the wavefront to arrive
            asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
            const long long lb2 = (long long)__builtin_readcyclecounter();
            lacc[0] += (unsigned long long)(lb0 - vmark); lacc[1] += (unsigned long long)(lb1 - lb0);
            lacc[2] += (unsigned long long)(lb2 - lb1); lacc[3] += 1;
          }
#endif
        }
      } else if (wave == kPipeCompute + 2) {
        // ------------------------------------------------------------ loader A: own data of node pos + 1
        // The registers hold what was requested during visit pos - 1 (its HBM latency lies behind a
        // whole visit, not inside one -- on the serial chains of the reference's node order the visit
        // was as long as this loader's round trip); it goes to the stage, then node pos + 2's requests
        // go out and stay in flight across the barrier (plain loads: the compiler waits at their first use,
        // which is in the next visit).
        if (pos + 1 >= p0 && pos + 1 < p1) {
          const NodeDesc nx = decode_desc(wa1);
          const int ntot = nx.nout + nx.nin;
          if (act) stn[kStD + lane] = rdk;
#pragma unroll
          for (int j = 0; j < 8; ++j) {
            if (j < ntot && act) {
              if (j < nx.nout) stn[kStM + j * kWave + lane] = rmv[j];
              if (!SHARED) { stn[kStQ + j * kWave + lane] = rqv[j]; stn[kStQP + j * kWave + lane] = rqpv[j]; }
            }
          }
          if (lane < 8) stn[kStA + lane] = rav;
          if (lane == 0) stn[kStG] = (double)1 / (double)(nx.nout > nx.nin ? nx.nout : nx.nin);
          int wa4 = 0;
          if (pos + 4 < p1) wa4 = desc[(size_t)(pos + 4) * DW + lane];  // three nodes ahead: waited for at the top of the next visit
          wa1 = wa2; wa2 = wa3; wa3 = wa4;
          if (pos + 2 < p1) PIPE_REQUEST_OWN(wa1);
        }
      } else if (wave == kPipeCompute + 1) {
        // ------------------------------------------------------------ storer: node pos - 1
        if (pos - 1 >= p0) {
          const NodeDesc pd = decode_desc(dring[((pos - 1) % 3) * kWave + lane]);
          const double *scp = scal + ((pos + 1) & 1) * kScalDoubles;  // parity of pos - 1
          if (UPDATE) {
#pragma unroll
            for (int j = 0; j < 8; ++j) {
              if (j < pd.nout) {
                double *mb = ((pd.remote >> j) & 1) ? (((pd.remote >> (8 + j)) & 1) ? p.peer_msg1 : p.peer_msg0) : p.msg;
                const int ej = ((pd.remote >> j) & 1) ? pd.re[j] : pd.e[j];  // the neighbour numbers the edge itself
                if (act) st_sc1(mb + (size_t)ej * K + lane, hprev[j * kWave + lane]);
                if (BACKWARD && lane == 0) p.lbterms[pd.lbe[j]] = scp[j];
              }
            }
            if (BACKWARD && lane == 0) p.lbterms[pd.lbn] = scp[8];
          }
          if (PRIMAL && lane == 0) {
            const int xi = ((const int *)(scp + 10))[0];
            st_sc1(p.x + pd.node, xi);
            if (pd.remote & (1 << 16)) st_sc1(p.peer_x0 + pd.pn[0], xi);
            if (pd.remote & (1 << 17)) st_sc1(p.peer_x1 + pd.pn[1], xi);
            p.eterms[pd.epos] = scp[9];
          }
#ifdef STEREO_HIP_VISIT_PROFILE
          const long long sb0 = (long long)__builtin_readcyclecounter();
#endif
          asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#ifdef STEREO_HIP_VISIT_PROFILE
          if (pos > p0 + 3) {
            const long long sb1 = (long long)__builtin_readcyclecounter();
            lacc[4] += (unsigned long long)(sb0 - vmark); lacc[5] += (unsigned long long)(sb1 - sb0); lacc[6] += 1;
          }
#endif
          if (lane == 0) {
            st_sc1(p.done + pd.rank, epoch);
            if (pd.remote & (1 << 16)) st_sc1(p.peer_done0 + pd.pn[0], epoch);
            if (pd.remote & (1 << 17)) st_sc1(p.peer_done1 + pd.pn[1], epoch);
          }
        }
      } else if (wave == kPipeCompute + 3) {
        // ------------------------------------------------------------ primal of node pos
        if (PRIMAL && have_node) {
          const int *sti = (const int *)(st + kStI);
          const int f = __builtin_amdgcn_readfirstlane(sti[2]);
          const int nout = f & 15, nin = (f >> 4) & 15, md = (f >> 16) & 255, ntot = nout + nin;
          double db = act ? st[kStD + lane] : 0.0;
#pragma unroll
          for (int j = 0; j < 8; ++j) {
            if (j >= nout && j < ntot) {
              const int sl = __builtin_amdgcn_readfirstlane(sti[12 + j]);
              const int ks = sl >= 8 ? xprev2 : sl >= 0 ? xprev : __builtin_amdgcn_readfirstlane(sti[64 + j]);
              const int mdj = (md >> j) & 1;
              double d;
              if (SHARED) {
                const double pks = readlane_f64(posk, ks);
                d = mdj == 0 ? pks - posk : posk - pks;
              } else {
                const double qvj = act ? st[kStQ + j * kWave + lane] : 0.0;
                const double qpj = act ? st[kStQP + j * kWave + lane] : 0.0;
                d = mdj == 0 ? readlane_f64(qpj, ks) - qvj : qpj - readlane_f64(qvj, ks);
              }
              const double v = KERNEL == 1 ? fabs(d) : d * d;
              db += st[kStA + j] * min_raw(v, p.lambda);  // (v, lambda >= 0: the value of v < lambda ? v : lambda)
            }
          }
          double di = db;
#pragma unroll
          for (int j = 0; j < 8; ++j)
            if (j < nout) di += st[kStM + j * kWave + lane];
          // first minimum (AddColumn's vectorMin): the minimum by a plain reduction, then the lowest lane that has it
          const double dim = act ? di : inf;
          const double vbest = wave_min_dpp(dim);
          const int bi = __builtin_ctzll(__builtin_amdgcn_ballot_w64(dim == vbest));
          xprev2 = xprev; xprev = bi;
          const double eb = readlane_f64(db, bi);
          if (lane == 0) { sc[9] = eb; ((int *)(sc + 10))[0] = bi; }
        }
      }
      if (p.prof) busy += (unsigned long long)((long long)__builtin_readcyclecounter() - tstart);
      VSTAMP(4);
      __syncthreads();
      VSTAMP(5);
      if (ctl[1]) return;  // a dependency wait gave up (bounded spin); host reports it
    }
    if (p.timeline && tid == 0) p.timeline[((size_t)D * p.nruns[0] + run) * 2 + 1] = wall_clock64();
    if (p.prof && lane == 0 && (p.prof_run < 0 || run == p.prof_run)) {
      // busy cycles before the barrier per role: compute (wave 0), loader, storer, primal; steps
      const int slot = wave == 0 ? 0 : wave == kPipeCompute ? 1 : wave == kPipeCompute + 1 ? 2 : wave == kPipeCompute + 3 ? 3 : -1;
      if (slot >= 0) atomicAdd(p.prof + slot, busy);
      atomicAdd(p.prof + 32 + wave, busy);  // every wave: cycles from barrier to barrier arrival
#ifdef STEREO_HIP_VISIT_PROFILE
      if (wave == 0) for (int i = 0; i < 6; ++i) atomicAdd(p.prof + 48 + i, vacc[i]);
      if (wave == 0) for (int i = 0; i < 5; ++i) atomicAdd(p.prof + 56 + i, macc[i]);
      if (wave == kPipeCompute) for (int i = 0; i < 4; ++i) atomicAdd(p.prof + 16 + i, lacc[i]);
      if (wave == kPipeCompute + 1) for (int i = 4; i < 7; ++i) atomicAdd(p.prof + 16 + i, lacc[i]);
#endif
      if (wave == 0) atomicAdd(p.prof + 6, (unsigned long long)(p1 - p0));
    }
  }
}

template <int KERNEL, bool BACKWARD, bool PRIMAL, bool UPDATE, bool SHARED>
__global__ __launch_bounds__(kPipeThreads) void trws_pipe_kernel(DevParams p, int epoch) {
  pipe_body<KERNEL, BACKWARD, PRIMAL, UPDATE, SHARED>(p, epoch);
}

template <int KERNEL, bool BACKWARD, bool PRIMAL, bool UPDATE, bool SHARED>
__global__ __launch_bounds__(kPipeThreads) void trws_pipe_group_kernel(GroupArgs ga, int epoch) {
  pipe_body<KERNEL, BACKWARD, PRIMAL, UPDATE, SHARED>(ga.pp[group_strip(ga)], epoch);
}

#undef PIPE_REQUEST_OWN
#undef PIPE_LOAD8
#undef VSTAMP

}  // namespace

size_t pipe_lds_bytes() {
  return sizeof(double) * (2 * kStageDoubles + 4 * 8 * kWave + 2 * kScalDoubles + kPipeCompute * kPipeTab + 2 + 3 * kWave / 2 + kWave);
}
int pipe_threads() { return kPipeThreads; }

void pipe_set_attributes() {
  const int lds = (int)pipe_lds_bytes();
#define SET_P(KER, BW, PR, UP)                                                                                                          \
  STEREO_HIP_CHECK(hipFuncSetAttribute((const void *)trws_pipe_kernel<KER, BW, PR, UP, true>, hipFuncAttributeMaxDynamicSharedMemorySize, lds));        \
  STEREO_HIP_CHECK(hipFuncSetAttribute((const void *)trws_pipe_kernel<KER, BW, PR, UP, false>, hipFuncAttributeMaxDynamicSharedMemorySize, lds));       \
  STEREO_HIP_CHECK(hipFuncSetAttribute((const void *)trws_pipe_group_kernel<KER, BW, PR, UP, true>, hipFuncAttributeMaxDynamicSharedMemorySize, lds));  \
  STEREO_HIP_CHECK(hipFuncSetAttribute((const void *)trws_pipe_group_kernel<KER, BW, PR, UP, false>, hipFuncAttributeMaxDynamicSharedMemorySize, lds))
  SET_P(1, false, false, true); SET_P(1, true, false, true); SET_P(1, false, true, true); SET_P(1, false, true, false);
  SET_P(2, false, false, true); SET_P(2, true, false, true); SET_P(2, false, true, true); SET_P(2, false, true, false);
#undef SET_P
}

#define PIPE_SWITCH(NAME, ARG)                                                                                          \
  const size_t plds = pipe_lds_bytes();                                                                                 \
  const dim3 grid(blocks), block(kPipeThreads);                                                                         \
  _Pragma("clang diagnostic push")                                                                                      \
  if (kernel == 1) { PIPE4(NAME, 1, ARG) } else { PIPE4(NAME, 2, ARG) }                                                 \
  _Pragma("clang diagnostic pop")                                                                                       \
  STEREO_HIP_CHECK(hipGetLastError());
#define PIPE1(NAME, KER, BW, PR, UP, ARG)                                                                               \
  do {                                                                                                                  \
    if (shared) hipLaunchKernelGGL((NAME<KER, BW, PR, UP, true>), grid, block, plds, s, ARG, epoch);                     \
    else hipLaunchKernelGGL((NAME<KER, BW, PR, UP, false>), grid, block, plds, s, ARG, epoch);                           \
  } while (0)
#define PIPE4(NAME, KER, ARG)                                                                                           \
  switch (what) {                                                                                                       \
    case 0: PIPE1(NAME, KER, false, false, true, ARG); break;                                                           \
    case 1: PIPE1(NAME, KER, true, false, true, ARG); break;                                                            \
    case 2: PIPE1(NAME, KER, false, true, true, ARG); break;                                                            \
    default: PIPE1(NAME, KER, false, true, false, ARG); break;                                                          \
  }

void launch_pipe(int kernel, bool shared, int what, int blocks, hipStream_t s, const DevParams &p, int epoch) {
  PIPE_SWITCH(trws_pipe_kernel, p)
}
void launch_pipe_group(int kernel, bool shared, int what, int blocks, hipStream_t s, const GroupArgs &ga, int epoch) {
  PIPE_SWITCH(trws_pipe_group_kernel, ga)
}
#undef PIPE4
#undef PIPE1
#undef PIPE_SWITCH

}  // namespace stereo
