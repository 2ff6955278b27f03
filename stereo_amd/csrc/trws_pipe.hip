// TRW-S pipelined sweep kernel for K <= 64 (both smoothness kernels): role-specialised waves, one
// barrier per visit.  Part of libstereo_hip.so; overview in trws_plan.hip.
#include <hip/hip_runtime.h>

#include <cstdint>

#include "../../include/stereo_hip.h"
#include "common.h"
#include "trws_dev.h"
#include "trws_launch.h"
#include "trws_spec.h"

namespace stereo {
namespace {

#define RLI(v, i) __builtin_amdgcn_readlane((v), (i))

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
// SPEC: the instantiation that knows the speculative schedule (trws_graph.h: Sweep::Spec; trws_spec.h) -- a kernel of its
// own, launched only when a plan's sweeps use it: everything it adds (the runner's ticket, a segment's first visit,
// the held-back flags, the rows kept for a second walk, the commit) costs the visit loops registers, and the plain
// schedule's kernels stay what they were.
template <int KERNEL, bool BACKWARD, bool PRIMAL, bool UPDATE, bool SHARED, bool SPEC = false>
__device__ __forceinline__ void pipe_body(DevParams p, int epoch) {
  static_assert(!SPEC || (SHARED && KERNEL == 1), "the speculative schedule exists for shared positions and the linear kernel");
  extern __shared__ __attribute__((aligned(16))) double lds[];
  double *stage0 = lds;                                   // 2 stages
  double *hand = lds + 2 * kStageDoubles;                 // ring of 4 x 8 x 64: the last visits' new messages
  double *scal = hand + 4 * 8 * kWave;                    // 2 x kScalDoubles
  double *hqtab = scal + 2 * kScalDoubles;                // per compute wave: 64 x (h, q, u, v)
  int *ctl = (int *)(hqtab + kPipeCompute * kPipeTab);    // [0] run, [1] abort
  int *dring = ctl + 4;                                   // the last three descriptors (the storer's comes from here, not from HBM)
  double *zrow = (double *)(dring + 3 * kWave);           // 64 zeros: where the Di loop finds the message rows a node does not have
  double *gtab = zrow + kWave;                            // gtab[k] = (double)1 / (double)k, k = 1 .. 8 (MRFEnergy.cpp:207-228), divided once
  double *xchg = gtab + 16;                               // per compute wave (as helper): partial minima and match counts of a shared message
  int *xflag = (int *)(xchg + kPipeCompute * kPipeXchg);  // ... and the flag behind them (CoopPart, trws_dev.h)
  constexpr int D = BACKWARD ? 1 : 0;
  // The runner of the speculative schedule holds ticket 0 of either direction, and the workgroup that draws it calls
  // it HERE, before anything the visit loops keep in registers exists: a call in the ticket loop has all of that live
  // across it, the compiler spills what the callee touches over its whole live range -- reloads inside the visit
  // loops -- and which registers the callee touches changed with every edit of the runner (measured: a fifth of an
  // iteration either way).  At this point only the kernel arguments are live.
  bool have_ticket = false;
  if (SPEC) {
    if (threadIdx.x == 0) {
      const int t_ = atomicAdd(p.ticket, 1);
      ctl[0] = t_ < p.ntickets[D] ? (p.run_order[D] ? p.run_order[D][t_] : t_) : p.nruns[D];
      ctl[1] = 0; ctl[2] = 0; ctl[3] = 0;
    }
    __syncthreads();
    const int first = __builtin_amdgcn_readfirstlane(ctl[0]);
    __syncthreads();
    if (first == -1) chain_runner<BACKWARD, PRIMAL, UPDATE>(p.self, epoch, kPipeLdsDoubles, kPipeCtlOff);
    else have_ticket = true;
  }
  const int K = p.K;
  const double inf = __builtin_huge_val();
  const int tid = threadIdx.x, lane = tid & (kWave - 1);
  const int wave = __builtin_amdgcn_readfirstlane(tid / kWave);
  constexpr int DW = TrwsGraph::kDescWords;
  const int32_t *desc = p.desc[D];
  const bool act = lane < K;
  const double posk = (SHARED && act) ? p.pos[lane] : 0.0;
  int look_streak = 0;  // failed second looks in a row (this compute wave): see message_regs
  const int perm_shared = (SHARED && wave < kPipeCompute) ? (act ? (int)p.perm_pos[lane] : lane) : -1;  // source order by position, once
  if (!SPEC && tid == 0) { ctl[1] = 0; ctl[2] = 0; ctl[3] = 0; }
  if (tid < kWave) zrow[tid] = 0.0;
  if (tid < 16) gtab[tid] = (double)1 / (double)(tid > 0 ? tid : 1);
  if (tid < kPipeCompute) xflag[tid] = 0;
  // A visit is bound by the instructions the CU's four SIMDs issue for all twelve waves (~4400 per visit
  // before round 4, 45 % of them scalar), so the service waves are written branch-poor: no exec-mask region
  // per row -- a lane beyond the last label works on label K - 1 again (loads read an element that exists,
  // stores repeat lane K - 1's value at lane K - 1's address), a row the node does not have is skipped by
  // ONE scalar branch on a uniform count or on a mask the host put into the descriptor (word 55: which
  // incoming rows come from global memory), row addresses are one unsigned 32 x 32 -> 64 multiply and one
  // shift-add on a per-lane base, and values one lane needs from another lane's descriptor word come by
  // ds_bpermute instead of chains of eight compares and selects.
#define PIPE_ROW(BASE, E) ((BASE) + (size_t)((unsigned long long)(unsigned)(E) * (unsigned long long)(unsigned)Kv))
  if (wave < kPipeCompute && lane < 2 * kPipePad) {
    // padding of the source tables (entries -16 .. -1 and 64 .. 79): never overwritten afterwards
    double *e = hqtab + wave * kPipeTab + 4 * (lane < kPipePad ? lane : kWave + lane);
    e[0] = inf; e[1] = 0; e[2] = 0; e[3] = 0;
  }
  if ((p.debug & 2) && !BACKWARD) p.prof = nullptr;  // profile backward sweeps only
  if ((p.debug & 4) && BACKWARD) p.prof = nullptr;   // profile forward sweeps only

  for (;;) {
    // (ctl[3]: the workgroup walks the speculative segment it holds a second time -- no new ticket)
    if (tid == 0 && !(SPEC && (ctl[3] || have_ticket))) { const int t_ = atomicAdd(p.ticket, 1); ctl[0] = t_ < p.ntickets[D] ? (p.run_order[D] ? p.run_order[D][t_] : t_) : p.nruns[D]; }
    have_ticket = false;
    __syncthreads();
    const int run = __builtin_amdgcn_readfirstlane(ctl[0]);
    const int second_walk = SPEC ? __builtin_amdgcn_readfirstlane(ctl[3]) : 0;
    // (the helpers' exchange flags show schedule positions: a second walk of a speculative segment visits the SAME
    //  positions again, and a flag left by the first walk would pass for the helper's word of the second -- the
    //  finishing wave would merge the first walk's partial minima, which are those of almost the same message.
    //  Every wave is behind the last visit's barrier here: nothing is being published or collected.)
    if (tid < kPipeCompute && !(p.debug & 131072)) xflag[tid] = 0;   // (development switch 131072: flags left as they are)
    __syncthreads();
    if (SPEC && tid == 0) ctl[3] = 0;
    if (run >= p.nruns[D]) break;
    if (SPEC && run < 0) continue;   // (the runner's ticket is ticket 0: drawn and served above)
    const int p0 = p.run_ptr[D][run], p1 = p.run_ptr[D][run + 1];
    // a segment of the speculative schedule (trws_graph.h: Sweep::Spec): its first visit takes what the node in front
    // hands over from the runner's rows, its completion flags wait for the commit below the visit loops
    const int seg = (SPEC && p.spec_kind[D]) ? __builtin_amdgcn_readfirstlane(p.spec_kind[D][run]) - 1 : -1;
    unsigned long long busy = 0;
#ifdef STEREO_HIP_VISIT_PROFILE
    unsigned long long vacc[6] = {0, 0, 0, 0, 0, 0}, macc[5] = {0, 0, 0, 0, 0}, lacc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#define VSTAMP(i) do { const long long n_ = (long long)__builtin_readcyclecounter(); vacc[i] += (unsigned long long)(n_ - vmark); vmark = n_; } while (0)
#else
#define VSTAMP(i) do { } while (0)
#endif
    const bool spec_in = SPEC && seg > 0 && !second_walk;
    int xprev = 0, xprev2 = 0;  // primal wave: labels of the previous two nodes of the run
    int wnext = 0;              // loader: raw descriptor word of the node after next (prefetched)
    if (wave == kPipeCompute) wnext = desc[(size_t)p0 * DW + lane];
    // loader A (own data, two visits deep): descriptors of the next three nodes and the parked requests
    int wa1 = 0, wa2 = 0, wa3 = 0;
    double rdk = 0, rmv[8] = {0, 0, 0, 0, 0, 0, 0, 0}, rqv[8] = {0, 0, 0, 0, 0, 0, 0, 0}, rqpv[8] = {0, 0, 0, 0, 0, 0, 0, 0}, rav = 0;
#define PIPE_LOAD8(DST, PTR) DST = *(PTR)   /* plain loads: the compiler keeps them in flight across the barrier and waits at the first use */
#define PIPE_REQUEST_OWN(W)                                                                          \
    do {                                                                                             \
      const int rf_ = RLI((W), 2);                                                                   \
      const int rnout = rf_ & 15, rtot = rnout + ((rf_ >> 4) & 15);                                  \
      { const double *a_ = PIPE_ROW(p.unary + lkv, RLI((W), 0)); PIPE_LOAD8(rdk, a_); }                   \
      _Pragma("unroll") for (int j = 0; j < 8; ++j) {                                                \
        if (j < rnout && (UPDATE || PRIMAL)) { const double *a_ = PIPE_ROW((p.msg + lkv), RLI((W), 4 + j)); PIPE_LOAD8(rmv[j], a_); } \
        if (!SHARED && j < rtot) {                                                                   \
          const double *a_ = PIPE_ROW(p.q + lkv, RLI((W), 4 + j)), *b_ = PIPE_ROW(p.qprim + lkv, RLI((W), 4 + j)); \
          PIPE_LOAD8(rqv[j], a_); PIPE_LOAD8(rqpv[j], b_);                                           \
        }                                                                                            \
      }                                                                                              \
      rav = p.alpha[__shfl((W), 4 + (lane & 7), kWave)];  /* lane j: weight of the node's edge j (words of absent edges hold 0) */ \
    } while (0)
    // storer: node pos's descriptor word (per lane), fetched during visit pos and used right behind the barrier that ends it
    int sw = 0;
    if (wave == kPipeCompute + 2) {
      wa1 = desc[(size_t)p0 * DW + lane];
      if (p0 + 1 < p1) wa2 = desc[(size_t)(p0 + 1) * DW + lane];
      if (p0 + 2 < p1) wa3 = desc[(size_t)(p0 + 2) * DW + lane];
      { const int Kv = K, lkv = lane < K ? lane : K - 1; PIPE_REQUEST_OWN(wa1); }
    }
    if (p.timeline && tid == 0) p.timeline[((size_t)D * p.tl_stride + run) * 2] = wall_clock64();

    // Every role walks the run in its own loop -- the same visits, the same barrier at the end of each: the hardware
    // barrier counts arrivals, whichever s_barrier instruction a wave arrives at -- so that what one role keeps across
    // visits (the loaders' parked requests and descriptor words, the primal wave's labels) and the kernel parameters it
    // uses are live in ITS loop only: in one loop for all roles the function sat at the scalar-register limit, ~300
    // scalars spilled into VGPR lanes, and every edit anywhere moved spill code onto the compute waves' path.
    if (wave < kPipeCompute) {
    // ---- compute
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
      // (the workgroup's abort word -- a loader's wait gave up during the PREVIOUS visit -- is requested here and
      //  looked at in front of the barrier that ends this visit: read behind that barrier, as it used to be, its
      //  LDS round trip was the first thing on every wave's path into the next visit)
      const int aborted = __hip_atomic_load(ctl + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      // (opaque copies, renewed every visit: what is derived from them -- per-lane row bases, the label-count
      //  tests of the envelope code, ... -- is recomputed where it is used, one instruction each; left visible as
      //  loop invariants, the compiler hoists dozens of such values out of the visit loop and keeps them in
      //  spilled registers, scalar ones in VGPR lanes, vector ones in scratch memory)
      int Kv = K;
      asm volatile("" : "+s"(Kv));
      const int lkv = lane < Kv ? lane : Kv - 1;
      {
        // ------------------------------------------------------------ compute
        if (UPDATE && have_node) {
          const int *sti = (const int *)(st + kStI);
          const int f = __builtin_amdgcn_readfirstlane(sti[2]);
          const int nout = f & 15, md = (f >> 16) & 255;
          const int myrow = sti[72 + (lane & 7)];  // LDS offsets (doubles) of the node's message rows, from the loader
          const unsigned twins = SHARED ? (unsigned)__builtin_amdgcn_readfirstlane(sti[kDescTwin]) : 0x76543210u;
          VSTAMP(0);
          if (wave < nout || (BACKWARD && wave == 0)) {  // waves without a message stay out of the way
          double Di = act ? st[kStD + lane] : 0.0;
          // (this wave's own old message is one of the rows; it is read once more below rather than
          //  picked out of the loop with eight selects)
          const double mown = st[kStM + (wave < nout ? wave : 0) * kWave + lane];
          const int partner = wave < nout ? (int)((twins >> (4 * wave)) & 15u) : wave;
          const int twin_mask = (SHARED && KERNEL == 1) ? __builtin_amdgcn_readfirstlane((int)st[kStA + 9]) : 0;   // (loader A's verdict)
          const double alpha_me = st[kStA + (wave < nout ? wave : 0)];
          // (all eight rows are requested together and added in list order; a row the node does not have
          //  is the zero row -- x + 0.0 == x --, so nothing here branches on the node's degree and the
          //  reads share one LDS latency instead of paying one each)
          //  (per-edge positions: two batches of four -- that kernel has no registers to spare)
#pragma unroll
          for (int b = 0; b < 8; b += (SHARED ? 8 : 4)) {
            double rowv[8];
#pragma unroll
            for (int j = 0; j < (SHARED ? 8 : 4); ++j) rowv[j] = lds[__builtin_amdgcn_readlane(myrow, b + j) + lane];
#pragma unroll
            for (int j = 0; j < (SHARED ? 8 : 4); ++j) Di += rowv[j];
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
                perm = (src_is_qprim ? p.perm_qp : p.perm_q) + (size_t)e * Kv;
              }
              const double alpha = alpha_me;
              // a twin -- the node's other message to the same neighbour -- with the same weight and the same old
              // message is the same message: one wave finishes it, the other takes half of its loop (CoopPart)
              CoopPart cp;
              if (SHARED && KERNEL == 1 && ((twin_mask >> j) & 1)) {
                const int helper = j > partner ? j : partner;
                cp.word = 1 | (j > partner ? 2 : 0) | (helper << 4) | (((pos + 1) & 0x7fffff) << 8);
              }
              VSTAMP(2);
              double newm = 0;
              const double v = message_regs<KERNEL, SHARED>(p, Kv, alpha, h, qsrc, qdst, perm, newm, lane,
                                                    hqtab + wave * kPipeTab + 4 * kPipePad, (SHARED && p.win_ok) ? p.window : -1, &look_streak
#ifdef STEREO_HIP_VISIT_PROFILE
                                                    , wave == 0 ? macc : nullptr
#else
                                                    , nullptr
#endif
                                                    , perm_shared, cp);
              VSTAMP(3);
              if (!cp.part()) {
                hcur[j * kWave + lane] = newm;   // (lanes beyond K fill the row's padding)
                if (BACKWARD && lane == 0) sc[j] = v;
                if (cp.active()) {
                  hcur[partner * kWave + lane] = newm;
                  if (BACKWARD && lane == 0) sc[partner] = v;
                }
              }
            }
          }
          }
        }
      }
      if (p.prof) busy += (unsigned long long)((long long)__builtin_readcyclecounter() - tstart);
      VSTAMP(4);
      if (aborted) return;  // a dependency wait gave up (bounded spin); host reports it
      __syncthreads();
      VSTAMP(5);
    }
    } else if (wave == kPipeCompute) {
    // ---- loader: stage node pos + 1
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
      // (the workgroup's abort word -- a loader's wait gave up during the PREVIOUS visit -- is requested here and
      //  looked at in front of the barrier that ends this visit: read behind that barrier, as it used to be, its
      //  LDS round trip was the first thing on every wave's path into the next visit)
      const int aborted = __hip_atomic_load(ctl + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      // (opaque copies, renewed every visit: what is derived from them -- per-lane row bases, the label-count
      //  tests of the envelope code, ... -- is recomputed where it is used, one instruction each; left visible as
      //  loop invariants, the compiler hoists dozens of such values out of the visit loop and keeps them in
      //  spilled registers, scalar ones in VGPR lanes, vector ones in scratch memory)
      int Kv = K;
      asm volatile("" : "+s"(Kv));
      const int lkv = lane < Kv ? lane : Kv - 1;
      {
        // ------------------------------------------------------------ loader: stage node pos + 1
        if (pos + 1 >= p0 && pos + 1 < p1) {
          const int w = wnext;
          if (pos + 2 < p1) wnext = desc[(size_t)(pos + 2) * DW + lane];
          int *stni = (int *)(stn + kStI);
          // (first visit of a speculative segment: the rows of the node in front come from the runner, behind the
          //  segment's flag -- to everybody else in the workgroup they look like rows of another run)
          //  (a second walk takes the same rows from the messages themselves: the segment in front has committed)
          const bool sfirst = SPEC && seg > 0 && pos + 1 == p0;
          stni[lane] = (sfirst && lane >= 12 && lane < 20 && w >= 0) ? -1 : w;
          dring[((pos + 1) % 3) * kWave + lane] = w;
          const int f = RLI(w, 2);
          const int nout = f & 15, ntot = nout + ((f >> 4) & 15);
          int ndep = (f >> 8) & 15;
          int fm = RLI(w, kDescFetch) & 255;   // incoming rows that come from global memory (behind flags)
          const int j8 = lane & 7;
          int sl = __shfl(w, 12 + j8, kWave);  // lane j: hand-over slot of the node's edge j, label source
          const int xn = __shfl(w, 32 + j8, kWave);
          int smask = 0, deprank = __shfl(w, 20 + (lane & 3), kWave);
          if (sfirst) {
            smask = (int)(__builtin_amdgcn_ballot_w64(lane < 8 && lane >= nout && lane < ntot && sl >= 0) & 255ull);
            fm |= smask;
            sl = -1;
            deprank = lane == ndep ? p.N + seg : deprank;
            ++ndep;
          }
          {
            // where the compute waves find the node's message rows at visit pos + 1: a message handed
            // over inside the run sits in the ring of the last two visits, everything else in this
            // stage, a row the node does not have is the zero row -- decided once here instead of by
            // every compute wave in scalar code
            const int row = j8 >= ntot ? (int)(zrow - lds)
                          : (j8 < nout || sl < 0) ? (int)(stn - lds) + kStM + j8 * kWave
                          : sl >= 8 ? (int)(hand - lds) + ((pos - 1) & 3) * 8 * kWave + (sl - 8) * kWave
                                    : (int)(hand - lds) + (pos & 3) * 8 * kWave + sl * kWave;
            if (lane < 8) stni[72 + lane] = row;
          }
          // (everything that does not depend on other workgroups -- unary, previous-sweep messages,
          //  positions, weights -- is loader A's, below, two visits ahead)
          // the completion flags of the foreign neighbours, then their data
          // (all flags are polled together: lane j watches dependency j)
          // (the rows that come from global memory -- two per foreign dependency, four on an ordinary row or
          //  chain node, up to eight where a node hangs on four other runs -- are the set bits of the mask, walked
          //  four at a time: row number and edge id of the k-th one are scalars; the addresses of the first four are
          //  formed BEFORE the wait, so that behind the flags nothing but the loads themselves is left)
          double mv[4];
          const double *fa[4];
          int jr[4];
          int mrest = UPDATE ? fm : 0;
#pragma unroll
          for (int k = 0; k < 4; ++k) {
            jr[k] = mrest ? __builtin_ctz(mrest) : -1;
            mrest &= mrest - 1;
            const int jk = jr[k] >= 0 ? jr[k] : 0;
            fa[k] = (SPEC && spec_in && ((smask >> jk) & 1)) ? PIPE_ROW((p.spec_rows + lkv), seg * 8 + jk)
                                                 : PIPE_ROW((p.msg + lkv), __builtin_amdgcn_readlane(w, 4 + jk));
          }
          // (the label of the node in front of a speculative segment: the runner's, kept apart from p.x -- what the commit
          //  compares must be what was read here)
          const int32_t *xa = (SPEC && spec_in && ((smask >> j8) & 1)) ? p.spec_x + seg : p.x + xn;
#ifdef STEREO_HIP_VISIT_PROFILE
          const long long lb0 = (long long)__builtin_readcyclecounter();
#endif
          if (ndep > 0) wait_for_dependencies_w(p, ndep, deprank, RLI(w, 1), epoch, lane, ctl + 1);
#ifdef STEREO_HIP_VISIT_PROFILE
          const long long lb1 = (long long)__builtin_readcyclecounter();
#endif
#pragma unroll
          for (int k = 0; k < 4; ++k)
            if (jr[k] >= 0) mv[k] = ld_sc1(fa[k]);
          int pxv = 0;
          if (PRIMAL && ((fm >> j8) & 1)) pxv = ld_sc1(xa);
#pragma unroll
          for (int k = 0; k < 4; ++k)
            if (jr[k] >= 0) stn[kStM + jr[k] * kWave + lane] = mv[k];
          while (mrest) {   // rows five to eight (rare)
            const int jx = __builtin_ctz(mrest);
            mrest &= mrest - 1;
            stn[kStM + jx * kWave + lane] = ld_sc1((SPEC && spec_in && ((smask >> jx) & 1)) ? PIPE_ROW((p.spec_rows + lkv), seg * 8 + jx)
                                                                                : PIPE_ROW((p.msg + lkv), __builtin_amdgcn_readlane(w, 4 + jx)));
          }
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
      }
      if (p.prof) busy += (unsigned long long)((long long)__builtin_readcyclecounter() - tstart);
      VSTAMP(4);
      if (aborted) return;  // a dependency wait gave up (bounded spin); host reports it
      __syncthreads();
      VSTAMP(5);
    }
    } else if (wave == kPipeCompute + 2) {
    // ---- loader A: own data of node pos + 1
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
      // (the workgroup's abort word -- a loader's wait gave up during the PREVIOUS visit -- is requested here and
      //  looked at in front of the barrier that ends this visit: read behind that barrier, as it used to be, its
      //  LDS round trip was the first thing on every wave's path into the next visit)
      const int aborted = __hip_atomic_load(ctl + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      // (opaque copies, renewed every visit: what is derived from them -- per-lane row bases, the label-count
      //  tests of the envelope code, ... -- is recomputed where it is used, one instruction each; left visible as
      //  loop invariants, the compiler hoists dozens of such values out of the visit loop and keeps them in
      //  spilled registers, scalar ones in VGPR lanes, vector ones in scratch memory)
      int Kv = K;
      asm volatile("" : "+s"(Kv));
      const int lkv = lane < Kv ? lane : Kv - 1;
      {
        // ------------------------------------------------------------ loader A: own data of node pos + 1
        // The registers hold what was requested during visit pos - 1 (its HBM latency lies behind a
        // whole visit, not inside one -- on the serial chains of the reference's node order the visit
        // was as long as this loader's round trip); it goes to the stage, then node pos + 2's requests
        // go out and stay in flight across the barrier (plain loads: the compiler waits at their first use,
        // which is in the next visit).
        if (pos + 1 >= p0 && pos + 1 < p1) {
          const int f = RLI(wa1, 2);
          const int nout = f & 15, nin = (f >> 4) & 15;
          stn[kStD + lane] = rdk;
#pragma unroll
          for (int j = 0; j < 8; ++j) {
            if (j < nout) stn[kStM + j * kWave + lane] = rmv[j];
            if (!SHARED && j < nout + nin) { stn[kStQ + j * kWave + lane] = rqv[j]; stn[kStQP + j * kWave + lane] = rqpv[j]; }
          }
          if (SPEC && UPDATE && seg >= 0 && !second_walk) {
            // a speculative segment keeps the rows its visits overwrite: a second walk starts from them (below)
#pragma unroll
            for (int j = 0; j < 4; ++j)
              if (j < nout) st_sc1(PIPE_ROW(p.spec_undo + lkv, (seg * p.spec_max_len + (pos + 1 - p0)) * 4 + j), rmv[j]);
          }
          if (lane < 8) stn[kStA + lane] = rav;
          if (lane == 0) stn[kStG] = gtab[nout > nin ? nout : nin];
          if (SHARED && KERNEL == 1 && UPDATE) {
            // Which outgoing messages have a twin that IS the same message (word 56 names the candidates; equal weights,
            // bitwise equal old rows -- this wave holds them in registers -- and shared positions make it so): decided
            // here, off the compute waves' path, for ordinary nodes (up to four outgoing messages); bit j of the mask.
            const unsigned tw = (unsigned)RLI(wa1, kDescTwin);
            int mask = 0;
            if (nout <= 4 && !(p.debug & 8192)) {   // (development switch 8192: no twins)
#pragma unroll
              for (int j = 0; j < 4; ++j) {
#pragma unroll
                for (int k = j + 1; k < 4; ++k) {
                  if (k < nout && (int)((tw >> (4 * j)) & 15u) == k) {
                    const bool same = !UNI(act && rmv[j] != rmv[k]) && RLI(__double2hiint(rav), j) == RLI(__double2hiint(rav), k) &&
                                      RLI(__double2loint(rav), j) == RLI(__double2loint(rav), k);
                    if (same) mask |= (1 << j) | (1 << k);
                  }
                }
              }
            }
            if (lane == 0) stn[kStA + 9] = (double)mask;
          }
          int wa4 = 0;
          if (pos + 4 < p1) wa4 = desc[(size_t)(pos + 4) * DW + lane];  // three nodes ahead: waited for at the top of the next visit
          wa1 = wa2; wa2 = wa3; wa3 = wa4;
          if (pos + 2 < p1) PIPE_REQUEST_OWN(wa1);
        }
      }
      if (p.prof) busy += (unsigned long long)((long long)__builtin_readcyclecounter() - tstart);
      VSTAMP(4);
      if (aborted) return;  // a dependency wait gave up (bounded spin); host reports it
      __syncthreads();
      VSTAMP(5);
    }
    } else if (wave == kPipeCompute + 1) {
    // ---- storer: node pos - 1
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
      // (the workgroup's abort word -- a loader's wait gave up during the PREVIOUS visit -- is requested here and
      //  looked at in front of the barrier that ends this visit: read behind that barrier, as it used to be, its
      //  LDS round trip was the first thing on every wave's path into the next visit)
      const int aborted = __hip_atomic_load(ctl + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      // (opaque copies, renewed every visit: what is derived from them -- per-lane row bases, the label-count
      //  tests of the envelope code, ... -- is recomputed where it is used, one instruction each; left visible as
      //  loop invariants, the compiler hoists dozens of such values out of the visit loop and keeps them in
      //  spilled registers, scalar ones in VGPR lanes, vector ones in scratch memory)
      int Kv = K;
      asm volatile("" : "+s"(Kv));
      const int lkv = lane < Kv ? lane : Kv - 1;
      {
        // ------------------------------------------------------------ storer: node pos - 1
        // The node's descriptor word is in a register since the previous visit (below) and nothing but the
        // fields the stores need is read from it, each row address being a scalar multiply and a shift-add:
        // the flag a row below waits for used to rise ~3 k cycles into this visit, 2.2 k of them the
        // decoding of all 55 descriptor fields and an exec-mask region per row.
        if (pos - 1 >= p0) {
          const double *scp = scal + ((pos + 1) & 1) * kScalDoubles;  // parity of pos - 1
          const int s_nout = RLI(sw, 2) & 15;
          const int s_remote = RLI(sw, kDescRemote), s_pn0 = RLI(sw, kDescPeerNode), s_pn1 = RLI(sw, kDescPeerNode + 1);
          if (UPDATE) {
            if (s_remote & 255) {   // strips: some rows go to a neighbour's array, under the neighbour's edge numbers
#pragma unroll
              for (int j = 0; j < 8; ++j) {
                if (j < s_nout) {
                  double *mb = ((s_remote >> j) & 1) ? (((s_remote >> (8 + j)) & 1) ? p.peer_msg1 : p.peer_msg0) : p.msg;
                  const int ej = ((s_remote >> j) & 1) ? RLI(sw, kDescPeerEdge + j) : RLI(sw, 4 + j);
                  st_sc1(PIPE_ROW(mb + lkv, ej), hprev[j * kWave + lkv]);
                }
              }
            } else {
#pragma unroll
              for (int j = 0; j < 8; ++j)
                if (j < s_nout) st_sc1(PIPE_ROW((p.msg + lkv), RLI(sw, 4 + j)), hprev[j * kWave + lkv]);
            }
            if (BACKWARD) {
              const int s_lbe = __shfl(sw, 24 + (lane & 7), kWave);   // lane j: position of message j's lower-bound term
              if (lane < s_nout) p.lbterms[s_lbe] = scp[lane];
              if (lane == 0) p.lbterms[RLI(sw, 3)] = scp[8];
            }
          }
          if (PRIMAL && lane == 0) {
            const int xi = ((const int *)(scp + 10))[0];
            st_sc1(p.x + RLI(sw, 0), xi);
            if (s_remote & (1 << 16)) st_sc1(p.peer_x0 + s_pn0, xi);
            if (s_remote & (1 << 17)) st_sc1(p.peer_x1 + s_pn1, xi);
            p.eterms[RLI(sw, kDescEpos)] = scp[9];
          }
          const int s_rank = RLI(sw, 1);
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
            if (!(SPEC && seg >= 0)) st_sc1(p.done + s_rank, epoch);   // (a speculative segment raises its flags when it commits)
            if (s_remote & (1 << 16)) st_sc1(p.peer_done0 + s_pn0, epoch);
            if (s_remote & (1 << 17)) st_sc1(p.peer_done1 + s_pn1, epoch);
          }
        }
        if (have_node) {
          // node pos's descriptor (it reached dring during visit pos - 1)
          sw = dring[(pos % 3) * kWave + lane];
        }
      }
      if (p.prof) busy += (unsigned long long)((long long)__builtin_readcyclecounter() - tstart);
      VSTAMP(4);
      if (aborted) return;  // a dependency wait gave up (bounded spin); host reports it
      __syncthreads();
      VSTAMP(5);
    }
    } else {
    // ---- primal of node pos
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
      // (the workgroup's abort word -- a loader's wait gave up during the PREVIOUS visit -- is requested here and
      //  looked at in front of the barrier that ends this visit: read behind that barrier, as it used to be, its
      //  LDS round trip was the first thing on every wave's path into the next visit)
      const int aborted = __hip_atomic_load(ctl + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      // (opaque copies, renewed every visit: what is derived from them -- per-lane row bases, the label-count
      //  tests of the envelope code, ... -- is recomputed where it is used, one instruction each; left visible as
      //  loop invariants, the compiler hoists dozens of such values out of the visit loop and keeps them in
      //  spilled registers, scalar ones in VGPR lanes, vector ones in scratch memory)
      int Kv = K;
      asm volatile("" : "+s"(Kv));
      const int lkv = lane < Kv ? lane : Kv - 1;
      {
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
      if (aborted) return;  // a dependency wait gave up (bounded spin); host reports it
      __syncthreads();
      VSTAMP(5);
    }
    }
    if (SPEC && seg >= 0) {
      const int verdict = spec_commit<BACKWARD, PRIMAL, UPDATE>(p.self, epoch, p0, p1, seg, spec_in ? 1 : 0);
      if (verdict == 2) return;
      if (verdict == 1) continue;   // (ctl[3] is set: the same run once more)
    }
    if (p.timeline && tid == 0) p.timeline[((size_t)D * p.tl_stride + run) * 2 + 1] = wall_clock64();
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

template <bool BACKWARD, bool PRIMAL, bool UPDATE>
__global__ __launch_bounds__(kPipeThreads) void trws_pipe_spec_kernel(DevParams p, int epoch) {
  pipe_body<1, BACKWARD, PRIMAL, UPDATE, true, true>(p, epoch);
}

template <int KERNEL, bool BACKWARD, bool PRIMAL, bool UPDATE, bool SHARED>
__global__ __launch_bounds__(kPipeThreads) void trws_pipe_group_kernel(GroupArgs ga, int epoch) {
  pipe_body<KERNEL, BACKWARD, PRIMAL, UPDATE, SHARED>(ga.pp[group_strip(ga)], epoch);
}

#undef PIPE_REQUEST_OWN
#undef PIPE_LOAD8
#undef VSTAMP
#undef PIPE_ROW
#undef RLI

}  // namespace

size_t pipe_lds_bytes() {
  static_assert(kPipeLdsDoubles == 2 * kStageDoubles + 4 * 8 * kWave + 2 * kScalDoubles + kPipeCompute * kPipeTab + 2 + 3 * kWave / 2 + kWave + 16 +
                                   kPipeCompute * kPipeXchg + kPipeCompute / 2, "LDS layout of pipe_body");
  return sizeof(double) * kPipeLdsDoubles;
}
size_t pipe_spec_lds_bytes() { return sizeof(double) * (kPipeLdsDoubles + kRunDoubles + kRunDummy); }   // ... with the runner's ring behind
int pipe_threads() { return kPipeThreads; }

void pipe_set_attributes() {
  const int lds = (int)pipe_lds_bytes(), slds = (int)pipe_spec_lds_bytes();
#define SET_S(BW, PR, UP) STEREO_HIP_CHECK(hipFuncSetAttribute((const void *)trws_pipe_spec_kernel<BW, PR, UP>, hipFuncAttributeMaxDynamicSharedMemorySize, slds))
  SET_S(false, false, true); SET_S(true, false, true); SET_S(false, true, true); SET_S(false, true, false);
#undef SET_S
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
  if (p.spec_kind[0] != nullptr && p.spec_kind[1] != nullptr && kernel == 1 && shared) {
    // the speculative schedule's kernel (the runner's LDS lies behind the visits')
    const size_t slds = pipe_spec_lds_bytes();
    const dim3 grid(blocks), block(kPipeThreads);
    switch (what) {
      case 0: hipLaunchKernelGGL((trws_pipe_spec_kernel<false, false, true>), grid, block, slds, s, p, epoch); break;
      case 1: hipLaunchKernelGGL((trws_pipe_spec_kernel<true, false, true>), grid, block, slds, s, p, epoch); break;
      case 2: hipLaunchKernelGGL((trws_pipe_spec_kernel<false, true, true>), grid, block, slds, s, p, epoch); break;
      default: hipLaunchKernelGGL((trws_pipe_spec_kernel<false, true, false>), grid, block, slds, s, p, epoch); break;
    }
    STEREO_HIP_CHECK(hipGetLastError());
    return;
  }
  PIPE_SWITCH(trws_pipe_kernel, p)
}
void launch_pipe_group(int kernel, bool shared, int what, int blocks, hipStream_t s, const GroupArgs &ga, int epoch) {
  PIPE_SWITCH(trws_pipe_group_kernel, ga)   // (strips keep the plain chain schedule)
}
#undef PIPE4
#undef PIPE1
#undef PIPE_SWITCH

}  // namespace stereo
