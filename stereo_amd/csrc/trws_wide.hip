// TRW-S pipelined sweep kernel for 64 < K <= 256 on shared strictly ascending positions (the large
// grids), linear kernel.  Part of libstereo_hip.so; overview in trws_plan.hip.
#include <hip/hip_runtime.h>

#include <cstdint>

#include "../../include/stereo_hip.h"
#include "common.h"
#include "trws_dev.h"
#include "trws_launch.h"
#include "trws_wspec.h"

namespace stereo {
namespace {

// per-phase cycle counters (STEREO_HIP_TRWS_PROF) are compiled in only with -DSTEREO_HIP_MESSAGE_PROFILE:
// sixteen 64-bit accumulators live across the visit loop cost registers the compute role needs
#ifdef STEREO_HIP_MESSAGE_PROFILE
#define WIDE_PROF(p) ((p).prof != nullptr)
#else
#define WIDE_PROF(p) false
#endif

// ---- wide-label pipelined sweep: 64 < K <= 256, shared strictly ascending positions -------
// The regime of the large grids (3000x2000x256).  A lane holds four labels (k = c * 64 + lane).
// One compute wave per outgoing message, one per SIMD for the four messages of an ordinary node
// (waves 4 .. 7 serve the six to eight messages of the interleaved last rows): each forms
// Di = S + incoming rows (S = D + the node's own previous-sweep rows, pre-summed by loader A in list
// order) and H = gamma Di - m, then
//  * with up to kWideSparse useful cones (h < vTrunc: five to eight of 256 on real volumes) looks at
//    nothing else: per useful cone the cost every destination gets from it, the smallest and second
//    smallest cost per destination, and the number of destinations whose own h lies on an arm of
//    that cone (the certificate's tangency test for the pairs that matter);
//  * with more (flat H) runs the dense windowed min-plus from a padded source table (a source farther
//    than lambda costs >= vTrunc exactly, by monotone rounding) and a closest-pair test on all u and
//    all v by bucketing -- conservative, O(K) -- refined on the useful cones if it finds a pair.
// If the certificate fails the wave runs the reference's serial envelope construction in LDS (one at
// a time per workgroup: shared scratch, rare).  Loader / storer / primal waves as in
// trws_pipe_kernel; the loader is split in two -- data nobody else writes, and data behind
// completion flags -- and both run two visits ahead with their requests parked in registers, so
// that no HBM round trip lies inside a visit.  One hardware barrier per visit.
// Kernel 1 (truncated linear) and, since round 4, kernel 2 (truncated quadratic: the hull-slope certificate of
// message_quad_fast on four labels per lane; exact messages only -- the MINPLUS option stays kernel 1).
// (Round 4: a "reach-aware" useful-cone loop -- the ~10-instruction update only for the label chunks within the
// truncation window of a cone, a 4-instruction key test u_t - u_i / v_t - v_i for the chunks beyond, one uniform
// branch per cone choosing a body specialised per source chunk -- executes 30 % fewer instructions per cone and was
// measured 8-11 % SLOWER in two variants (per-lane add-with-carry counts; scalar popcounts): 77 -> 83 / 86 ms per
// iteration at 1500x1000x256.  The short dependent compare -> mask -> count chains of the far test stall where the
// uniform body's four interleaved chunks cover each other's latency, and every cone pays a taken branch.)
// (Round-3 history, DESIGN.md 4.4: three waves per message with separate closest-pair waves, then
// three / two waves sharing the destinations of a message, were all slower than one wave per message:
// a visit is bound by the instructions its SIMDs issue, and every split repeats Di, H and the
// reductions.)
constexpr int kWideCompute = 8;
constexpr int kWideWaves = kWideCompute + 4;  // + loader (own data), loader (foreign data), storer, primal
constexpr int kWideThreads = kWideWaves * kWave;
constexpr int kWS = 260;    // LDS row stride in doubles (>= 256 + 1 breakpoints, multiple of 4)
constexpr int kWPad = 16;   // min-plus source table is padded by this many (+inf, 0) entries on both sides
constexpr int kWScr = 2 * (256 + 2 * kWPad);  // per compute wave scratch: (h, q) source table | 256 keys + 516 ints
constexpr int kWBuckets = 512;
constexpr int kWideSparse = 32;  // up to this many useful cones are evaluated one by one, more by the dense window loop
constexpr int kWStG = kWS + 8 * kWS + 8;            // gamma = 1 / max(n_out, n_in) of the node (the loader's division)
constexpr int kWStI = kWS + 8 * kWS + 10;           // int area of a stage (in doubles)
constexpr int kWStS = kWStI + 44;                   // S = D + the node's own (previous-sweep) message rows, added in list order by loader A
constexpr int kWStage = kWStS + kWS;  // ints: desc[64] px[8] row[8] inrow[8] (inrow[k]: where the k-th INCOMING row lives, the zero row beyond the last)
// stage: D[kWS] m[8][kWS] a[8] | ints desc[64] px[8] row[8] (row: where Di's k-th message row lives in LDS, in doubles) | S[kWS]

struct WidePtrs {
  double *stage0, *hand, *scr, *fb, *pos, *scal, *zrow;
  int *dring, *ctl, *xflag;
};
__device__ __forceinline__ WidePtrs wide_carve(double *lds) {
  WidePtrs w;
  w.stage0 = lds;                            // 2 * kWStage
  w.hand = w.stage0 + 2 * kWStage;           // 3 * 8 * kWS : new messages of the last three visits
  w.scr = w.hand + 3 * 8 * kWS;              // kWideCompute * kWScr
  w.fb = w.scr + kWideCompute * kWScr;       // 4 * kWS : sources, stack, breakpoints of the serial construction
  w.pos = w.fb + 4 * kWS;                    // kWS
  w.scal = w.pos + kWS;                      // 2 * kScalDoubles
  w.dring = (int *)(w.scal + 2 * kScalDoubles);  // 3 * 64 descriptor words (for the storer)
  w.ctl = w.dring + 3 * 64;                  // [0] run, [1] abort, [2] lock of the serial scratch
  w.xflag = w.ctl + 4;                       // kWideCompute flag words of the twin exchange (CoopPart, trws_dev.h)
  w.zrow = (double *)(w.ctl + 4 + kWideCompute);   // kWS zeros: the incoming rows a node does not have
  return w;
}
constexpr int kWideLdsDoubles = 2 * kWStage + 3 * 8 * kWS + kWideCompute * kWScr + 4 * kWS + kWS + 2 * kScalDoubles + 96 + 2 + kWideCompute / 2 + kWS;
static_assert(kWideLdsDoubles * 8 <= 160 * 1024, "wide kernel LDS");

#define WSYNC() do { __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront"); } while (0)

// inclusive prefix sum over the wave (Hillis-Steele inside rows of 16, then row broadcasts)
__device__ __forceinline__ int wave_incl_scan_i32(int v) {
  v += __builtin_amdgcn_update_dpp(0, v, 0x111, 0xF, 0xF, false);  // row_shr:1
  v += __builtin_amdgcn_update_dpp(0, v, 0x112, 0xF, 0xF, false);  // row_shr:2
  v += __builtin_amdgcn_update_dpp(0, v, 0x114, 0xF, 0xF, false);  // row_shr:4
  v += __builtin_amdgcn_update_dpp(0, v, 0x118, 0xF, 0xF, false);  // row_shr:8
  v += __builtin_amdgcn_update_dpp(0, v, 0x142, 0xA, 0xF, false);  // row_bcast:15 -> rows 1, 3
  v += __builtin_amdgcn_update_dpp(0, v, 0x143, 0xC, 0xF, false);  // row_bcast:31 -> rows 2, 3
  return v;
}

// True if two of the first K keys (key i = c * 64 + lane lives in r[c] of `lane`) are within
// delta of each other.  mn <= every key <= mn + span.  One wave, no sort: counting sort into
// 512 equal-width buckets (LDS counters), then every key is compared with the rest of its
// bucket and the whole next bucket.  Buckets are wider than 2 delta, so keys two or more
// buckets apart cannot be near.  Conservative `true` on degenerate key distributions.
// scr: 256 doubles + 516 ints.
__device__ __forceinline__ bool keys_within(const double (&r)[4], int K, int C, double delta, double mn,
                                            double span, double *scr, int lane) {
  const double inf = __builtin_huge_val();
  double *sorted = scr;
  int *cnt = (int *)(scr + 256);  // counters, afterwards start[0 .. kWBuckets + 1]
  if (!(span > (2.0 * kWBuckets) * delta) || !(span < inf)) return true;
  const double scale = (double)kWBuckets / span;
  ((int4 *)cnt)[2 * lane] = make_int4(0, 0, 0, 0);
  ((int4 *)cnt)[2 * lane + 1] = make_int4(0, 0, 0, 0);
  WSYNC();
  int b[4], rank[4];
#pragma unroll
  for (int c = 0; c < 4; ++c) {
    b[c] = 0; rank[c] = 0;
    if (c < C && c * kWave + lane < K) {
      const int bb = (int)((r[c] - mn) * scale);
      b[c] = bb > kWBuckets - 1 ? kWBuckets - 1 : bb < 0 ? 0 : bb;
      rank[c] = __hip_atomic_fetch_add(cnt + b[c], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
    }
  }
  WSYNC();
  const int4 c0 = ((int4 *)cnt)[2 * lane], c1 = ((int4 *)cnt)[2 * lane + 1];  // buckets 8 lane .. 8 lane + 7
  const int tot = c0.x + c0.y + c0.z + c0.w + c1.x + c1.y + c1.z + c1.w;
  const int incl = wave_incl_scan_i32(tot);
  int e = incl - tot;
  int4 s0, s1;
  s0.x = e; e += c0.x; s0.y = e; e += c0.y; s0.z = e; e += c0.z; s0.w = e; e += c0.w;
  s1.x = e; e += c1.x; s1.y = e; e += c1.y; s1.z = e; e += c1.z; s1.w = e;
  ((int4 *)cnt)[2 * lane] = s0;
  ((int4 *)cnt)[2 * lane + 1] = s1;
  if (lane == kWave - 1) { cnt[kWBuckets] = incl; cnt[kWBuckets + 1] = incl; }
  WSYNC();
  int st4[4];
#pragma unroll
  for (int c = 0; c < 4; ++c) st4[c] = cnt[b[c]];  // unconditional: one LDS round trip for all four
#pragma unroll
  for (int c = 0; c < 4; ++c)
    if (c < C && c * kWave + lane < K) sorted[st4[c] + rank[c]] = r[c];
  WSYNC();
  double u[4];
  int len[4], bq[4];
#pragma unroll
  for (int c = 0; c < 4; ++c) u[c] = sorted[c * kWave + lane];
#pragma unroll
  for (int c = 0; c < 4; ++c) {
    const int bb = (c < C && c * kWave + lane < K) ? (int)((u[c] - mn) * scale) : 0;
    bq[c] = bb > kWBuckets - 1 ? kWBuckets - 1 : bb < 0 ? 0 : bb;
  }
#pragma unroll
  for (int c = 0; c < 4; ++c) len[c] = cnt[bq[c] + 2];
#pragma unroll
  for (int c = 0; c < 4; ++c) {
    const int q = c * kWave + lane;
    len[c] = (c < C && q < K) ? len[c] - q : 0;  // keys q+1 .. q+len-1 share the bucket or the next one
  }
  bool bad = false;
  for (int i = 1;; ++i) {
    bool any = false;
    double o[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) {  // unconditional reads first: one LDS round trip per step
      const int q = c * kWave + lane + i;
      o[c] = sorted[q < 256 ? q : 255];
    }
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      const bool on = i < len[c];
      any = any || on;
      bad = bad || (on && !(fabs(o[c] - u[c]) > delta));
    }
    if (!UNI(any)) break;
    if (i >= 24) { bad = true; break; }  // crowded buckets: give up, serial path decides
  }
  return UNI(bad);
}

// Tangency test of the certified path: does any cone have its u or v key within delta of the key of
// one of the USEFUL cones (bit masks um, one per 64-label chunk)?  Cone number n of the useful ones
// (in label order) is looked at when n % mod == r, so that several waves can share the work.  A
// useful cone's keys are read from its own lane, so it matches itself -- and,
// positions being strictly ascending, nobody else unless there is a near tangency: the number of
// matches is counted with scalar instructions and compared with what the cones alone give.
__device__ __forceinline__ bool useful_cone_ties(const unsigned long long (&um)[4], int mod, int r,
                                                 const double (&uu)[4], const double (&vv)[4], double delta) {
  int turn = 0, matches = 0, mine = 0;
#pragma unroll
  for (int c = 0; c < 4; ++c) {
    unsigned long long mk = um[c];
    while (mk) {
      const int l = __builtin_ctzll(mk);
      mk &= mk - 1;
      const bool take = turn == r;
      turn = turn + 1 == mod ? 0 : turn + 1;
      if (take) {
        const double ui = readlane_f64(uu[c], l), vi = readlane_f64(vv[c], l);
        ++mine;
#pragma unroll
        for (int cc = 0; cc < 4; ++cc) {
          matches += __builtin_popcountll(__builtin_amdgcn_ballot_w64(fabs(uu[cc] - ui) <= delta));
          matches += __builtin_popcountll(__builtin_amdgcn_ballot_w64(fabs(vv[cc] - vi) <= delta));
        }
      }
    }
  }
  return matches != 2 * mine;
}

// Exchange of a twin pair (CoopPart in trws_dev.h, four labels per lane here): the helper leaves its partial minima,
// second minima (equal costs of two cones count twice in this kernel) and match count in ITS OWN scratch behind a flag.
__device__ __forceinline__ void wide_publish(double *xd, int *flag, int seq, const double (&m1)[4], const double (&m2)[4],
                                             int matches, int lane) {
#pragma unroll
  for (int c = 0; c < 4; ++c) { xd[c * kWave + lane] = m1[c]; xd[(4 + c) * kWave + lane] = m2[c]; }
  if (lane == 0) ((int *)(xd + 8 * kWave))[0] = matches;
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
  if (lane == 0) __hip_atomic_store(flag, seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}
__device__ __forceinline__ void wide_collect(const double *xd, int *flag, int seq, double (&m1)[4], double (&m2)[4], int &matches,
                                             int lane, int *abort_word) {
  int spins = 0;
  while (__hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) != seq && ++spins < kCoopSpinLimit) { }
  // (a partner that never publishes: the launch gives up through the abort word instead of merging whatever is there)
  if (spins >= kCoopSpinLimit && lane == 0) __hip_atomic_store(abort_word, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
  double b1[4], b2[4];
#pragma unroll
  for (int c = 0; c < 4; ++c) { b1[c] = xd[c * kWave + lane]; b2[c] = xd[(4 + c) * kWave + lane]; }
  matches += __builtin_amdgcn_readfirstlane(((const int *)(xd + 8 * kWave))[0]);
#pragma unroll
  for (int c = 0; c < 4; ++c) {
    const double lo = min_raw(m1[c], b1[c]), hi = max_raw(m1[c], b1[c]);
    m2[c] = min_raw(hi, min_raw(m2[c], b2[c]));
    m1[c] = lo;
  }
}

// The commit of a speculative segment (SPEC instantiation below; the protocol is trws_spec.h's spec_commit): called by all
// waves behind the barrier that ended the segment's last visit.  The segment in front commits first (its flag); then what
// this segment started from is compared, bit for bit, with what that segment's last node really handed over: equal ->
// the nodes' flags go up; different -> the overwritten rows are put back and the caller walks the visits again from the
// real rows.  Returns 0 committed, 1 walk again (ctl[3] set), 2 gave up.
template <bool BACKWARD, bool PRIMAL, bool UPDATE>
__device__ __attribute__((noinline)) int wide_spec_commit(const DevParams *pp_, int epoch_, int p0_, int p1_, int seg_, int compare_, int ctl_off_) {
  extern __shared__ __attribute__((aligned(16))) double wc_lds[];
  const DevParams &p = *pp_;
  const int epoch = epoch_, p0 = p0_, p1 = p1_, seg = seg_, compare = compare_;
  constexpr int D = BACKWARD ? 1 : 0;
  constexpr int DW = TrwsGraph::kDescWords;
  const int32_t *desc = p.desc[D];
  int *ctl = (int *)wc_lds + ctl_off_;   // [1] abort, [3] walk again, [4] (the first exchange flag, idle here) the verdict
  const int tid = threadIdx.x, lane = tid & (kWave - 1);
  const int wave = __builtin_amdgcn_readfirstlane(tid / kWave);
  const int K = p.K;
  if (compare) {
    if (wave == 0) {
      int differ = 0;
      if (!wait_flag(p, p.N + p.spec_nseg + seg, epoch, -2)) { if (lane == 0) ctl[1] = 1; }
      else {
        const int w = desc[(size_t)p0 * DW + lane];
        const int f = WRLI(w, 2);
        const int nout = f & 15, ntot = nout + ((f >> 4) & 15);
        for (int j = nout; j < ntot; ++j) {
          if (__builtin_amdgcn_readlane(w, 12 + j) < 0) continue;
          if (UPDATE) {
            const size_t ea = ((size_t)seg * 8 + j) * K, eb = (size_t)__builtin_amdgcn_readlane(w, 4 + j) * K;
            for (int c = 0; c < 4; ++c) {
              const int kk = c * kWave + lane, kc = kk < K ? kk : K - 1;
              const double a = ld_sc1(p.spec_rows + ea + kc), b = ld_sc1(p.msg + eb + kc);
              differ |= UNI(__double_as_longlong(a) != __double_as_longlong(b)) ? 1 : 0;
            }
          }
          if (PRIMAL) differ |= ld_sc1(p.spec_x + seg) != ld_sc1(p.x + __builtin_amdgcn_readlane(w, 32 + j)) ? 1 : 0;
        }
      }
      if (lane == 0) ctl[4] = differ;
    }
    __syncthreads();
    const int differ = __builtin_amdgcn_readfirstlane(ctl[4]);
    const int gave_up = __builtin_amdgcn_readfirstlane(ctl[1]);
    __syncthreads();
    if (tid == 0) ctl[4] = 0;
    if (gave_up) return 2;
    if (differ) {
      if (UPDATE) {
        for (int pos = p0 + wave; pos < p1; pos += kWideWaves) {
          const int w = desc[(size_t)pos * DW + lane];
          const int nout = WRLI(w, 2) & 15;
          for (int j = 0; j < nout && j < 4; ++j) {
            const size_t em = (size_t)__builtin_amdgcn_readlane(w, 4 + j) * K, eu = ((size_t)(seg * p.spec_max_len + (pos - p0)) * 4 + j) * K;
            for (int c = 0; c < 4; ++c) {
              const int kk = c * kWave + lane;
              if (kk < K) st_sc1(p.msg + em + kk, ld_sc1(p.spec_undo + eu + kk));
            }
          }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "agent");   // (the second walk's plain loads must not find this CU's L1 holding anything of the first)
      }
      if (tid == 0) { ctl[3] = 1; if (p.spec_stat) atomicAdd(p.spec_stat, 1ull); }
      __syncthreads();
      return 1;
    }
  }
  if (wave == 0) {
    // the segment behind first (the commits are a serial chain), then the nodes' own flags
    if (seg + 1 < p.spec_nseg && lane == 0) st_sc1(p.done + p.N + p.spec_nseg + seg + 1, epoch);
    for (int pos = p0 + lane; pos < p1; pos += kWave) st_sc1(p.done + desc[(size_t)pos * DW + 1], epoch);
    if (lane == 0 && p.spec_stat) atomicAdd(p.spec_stat + 1, 1ull);
  }
  return 0;
}

// SPEC: the instantiation that knows the speculative schedule of the long serial run (trws_wspec.h; even K, linear kernel,
// uniformly spaced positions): a kernel of its own, launched only when a plan's sweeps use it.
template <int KERNEL, bool BACKWARD, bool PRIMAL, bool UPDATE, bool SPEC = false>
__device__ __forceinline__ void wide_body(DevParams p, int epoch) {
  static_assert(!SPEC || KERNEL == 1, "the speculative schedule exists for the linear kernel");
  extern __shared__ __attribute__((aligned(16))) double lds[];
  const WidePtrs L = wide_carve(lds);
  const int tid = threadIdx.x, lane = tid & (kWave - 1);
  const int wave = __builtin_amdgcn_readfirstlane(tid / kWave);
  constexpr int D = BACKWARD ? 1 : 0;
  // The runner of the speculative schedule holds ticket 0 of either direction; the workgroup that draws it serves it HERE,
  // before anything of the visit loops exists: its LDS overlays theirs, and nothing is live across the call.
  bool have_ticket = false;
  int first_ticket = 0;
  if (SPEC) {
    int *ent = (int *)lds;
    if (tid == 0) {
      const int t_ = atomicAdd(p.ticket, 1);
      ent[0] = t_ < p.ntickets[D] ? (p.run_order[D] ? p.run_order[D][t_] : t_) : p.nruns[D];
    }
    __syncthreads();
    first_ticket = __builtin_amdgcn_readfirstlane(ent[0]);
    __syncthreads();
    if (first_ticket == -1) wide_chain_runner<BACKWARD, PRIMAL, UPDATE>(p.self, epoch);
    else have_ticket = true;
  }
  const int K = p.K;
  const int C = (K + kWave - 1) / kWave;  // 64-label chunks
  const double inf = __builtin_huge_val();
  constexpr int DW = TrwsGraph::kDescWords;
  const int32_t *desc = p.desc[D];
  for (int k = tid; k < kWS; k += kWideThreads) L.pos[k] = k < K ? p.pos[k] : inf;
  // positions on an exact arithmetic progression (checked on the host: pos[k+d] - pos[k] == d * step
  // bit for bit): the min-plus source table then holds h only and alpha |d step| is formed once per d
  const double ustep = p.uniform_step;
  const bool uniform = ustep != 0;
  if (tid == 0) { L.ctl[0] = first_ticket; L.ctl[1] = 0; L.ctl[2] = 0; L.ctl[3] = 0; }
  if (tid < kWideCompute) L.xflag[tid] = 0;
  for (int k = tid; k < kWS; k += kWideThreads) L.zrow[k] = 0.0;
#define WPOS(c) (L.pos[(c) * kWave + lane])        // this lane's four label positions (+inf beyond K)
#define WVALID(c) ((c) * kWave + lane < K)
  const double pos_first = p.pos[0], pos_last = p.pos[K - 1];
  // development profile (STEREO_HIP_TRWS_PROF): cycles of wave 0 per phase [0..15], busy cycles of
  // loader / storer / primal [16..18], hardware-barrier wait of wave 0 [19], visits [20]
  unsigned long long pacc[16] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, pbusy = 0, pwait = 0, pvis = 0, plate = 0;
#define WSTAMP(i) do { if (wprof) { const long long now_ = (long long)__builtin_readcyclecounter(); pacc[i] += (unsigned long long)(now_ - tmark); tmark = now_; } } while (0)
  __syncthreads();

  for (;;) {
    // (ctl[3]: the workgroup walks the speculative segment it holds a second time -- no new ticket)
    if (tid == 0 && !(SPEC && (L.ctl[3] || have_ticket))) { const int t_ = atomicAdd(p.ticket, 1); L.ctl[0] = t_ < p.ntickets[D] ? (p.run_order[D] ? p.run_order[D][t_] : t_) : p.nruns[D]; }
    have_ticket = false;
    __syncthreads();
    const int run = __builtin_amdgcn_readfirstlane(L.ctl[0]);
    const int second_walk = SPEC ? __builtin_amdgcn_readfirstlane(L.ctl[3]) : 0;
    // (the twins' exchange flags show schedule positions: a second walk visits the same positions, see trws_pipe.hip)
    if (SPEC && tid < kWideCompute) L.xflag[tid] = 0;
    __syncthreads();
    if (SPEC && tid == 0) L.ctl[3] = 0;
    if (run >= p.nruns[D]) break;
    if (SPEC && run < 0) continue;   // (the runner's ticket is ticket 0: drawn and served above)
    const int p0 = p.run_ptr[D][run], p1 = p.run_ptr[D][run + 1];
    // a segment of the speculative schedule: its first visit takes what the node in front hands over from the runner's
    // rows, its completion flags wait for the commit below the visit loops
    const int seg = (SPEC && p.spec_kind[D]) ? __builtin_amdgcn_readfirstlane(p.spec_kind[D][run]) - 1 : -1;
    const bool spec_in = SPEC && seg > 0 && !second_walk;
    (void)spec_in;
    const bool wprof = WIDE_PROF(p) && (p.prof_run < 0 || run == p.prof_run);  // STEREO_HIP_TRWS_PROF_RUN: one run only
    (void)wprof;
    if (p.timeline && tid == 0) p.timeline[((size_t)D * p.tl_stride + run) * 2] = wall_clock64();

    // One visit loop per role (not one loop with a role switch inside): state carried from visit
    // to visit -- the loader's parked registers -- then occupies registers in that role only.
#define WIDE_VISITS_BEGIN     for (int pos = p0 - 1; pos <= p1; ++pos) { \
      double *st = L.stage0 + (pos & 1) * kWStage; \
      double *stn = L.stage0 + ((pos + 1) & 1) * kWStage; \
      const int hb = ((pos % 3) + 3) % 3, hb1 = (((pos - 1) % 3) + 3) % 3, hb2 = (((pos - 2) % 3) + 3) % 3; \
      double *hcur = L.hand + hb * 8 * kWS, *hprev = L.hand + hb1 * 8 * kWS, *hprev2 = L.hand + hb2 * 8 * kWS; \
      double *sc = L.scal + (pos & 1) * kScalDoubles; \
      const bool have_node = pos >= p0 && pos < p1; \
      const int aborted_ = __hip_atomic_load(L.ctl + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); /* a loader's wait gave up during the previous visit */ \
      long long tmark = wprof ? (long long)__builtin_readcyclecounter() : 0; \
      const long long tvisit = tmark; \
      (void)st; (void)stn; (void)hcur; (void)hprev; (void)hprev2; (void)sc; (void)have_node; (void)tvisit;
#define WIDE_VISITS_END_(BARRIER)       if (wprof) { \
        const long long now_ = (long long)__builtin_readcyclecounter(); \
        if (wave == 0) pvis += have_node ? 1 : 0; \
        pbusy += (unsigned long long)(now_ - tvisit); \
        if (now_ - tvisit > 8000) plate += 1; \
        tmark = now_; \
      } \
      if (aborted_) { /* (looked at in front of the barrier: behind it the LDS round trip was every wave's first step into the next visit) */ \
        if (tid == 0) st_sc1(p.abort_flag, 1); \
        return; \
      } \
      BARRIER; \
      if (wprof && wave == 0) pwait += (unsigned long long)((long long)__builtin_readcyclecounter() - tmark); \
    }
#define WIDE_VISITS_END WIDE_VISITS_END_(__syncthreads())
    if (wave < kWideCompute) {
      WIDE_VISITS_BEGIN
        // ======================================================== compute waves
        // One wave per outgoing message (message j0 = wave): the four messages of an ordinary node run
        // on four SIMDs and do not compete for an instruction stream, nothing is handed over between
        // waves inside the visit.  Waves 4 .. 7 only work on the nodes with more than four outgoing
        // messages (the interleaved last rows, six to eight each), whose visits would otherwise take
        // two rounds on the critical path of the sweep's last serial stretch.
        const int j0 = wave;
        if (UPDATE && have_node) {
          const int *sti = (const int *)(st + kWStI);
          const int f = __builtin_amdgcn_readfirstlane(sti[2]);
          const int nout = f & 15, nin = (f >> 4) & 15;
          const bool fast_msg = p.certificate != 0;
          const bool working = j0 < nout;
          const unsigned twins = (unsigned)__builtin_amdgcn_readfirstlane(sti[kDescTwin]);
          if (working || (BACKWARD && wave == 0)) {
            double di[4] = {inf, inf, inf, inf};
            {
            const int inrow = sti[80 + (lane & 7)];  // LDS offsets of the INCOMING message rows (written by loader A)
            // Di = D + messages in list order (from the ring where the neighbour was one of
            // the last two visits of this run); the prefix D + rows 0 .. nout - 1 comes from loader A.
            // The first four incoming rows -- all an ordinary node has -- are requested together, without a
            // branch on the node's degree: a row the node does not have is the zero row (x + 0.0 == x), so
            // the sixteen reads share one LDS latency instead of one per row; rows five to eight as before.
#pragma unroll
            for (int c = 0; c < 4; ++c) di[c] = st[kWStS + c * kWave + lane];
            {
              double rin[4][4];
#pragma unroll
              for (int k = 0; k < 4; ++k) {
                const double *src = lds + __builtin_amdgcn_readlane(inrow, k);
#pragma unroll
                for (int c = 0; c < 4; ++c) rin[k][c] = src[c * kWave + lane];
              }
#pragma unroll
              for (int k = 0; k < 4; ++k)
#pragma unroll
                for (int c = 0; c < 4; ++c) di[c] += rin[k][c];
            }
            if (nin > 4) {
#pragma unroll
              for (int k = 4; k < 8; ++k) {
                if (k < nin) {
                  const double *src = lds + __builtin_amdgcn_readlane(inrow, k);
#pragma unroll
                  for (int c = 0; c < 4; ++c) di[c] += src[c * kWave + lane];
                }
              }
            }
#pragma unroll
            for (int c = 0; c < 4; ++c) di[c] = WVALID(c) ? di[c] : inf;
            if (BACKWARD) {
              const double dm = min_raw(min_raw(di[0], di[1]), min_raw(di[2], di[3]));
              const double node_vmin = wave_min_dpp(dm);
              if (wave == 0 && lane == 0) sc[8] = node_vmin;
#pragma unroll
              for (int c = 0; c < 4; ++c) di[c] -= node_vmin;
            }
            }
            WSTAMP(0);
            for (int j = j0; working && j < nout; j += kWideCompute) {
              const double gamma = st[kWStG];  // (double)1 / (double)max(n_out, n_in)
              const double alpha = st[kWS + 8 * kWS + j];
              const bool constant = UNI(alpha == 0);
              // Twin messages (trws_dev.h CoopPart): the node's other message to the same neighbour, with the same weight
              // and the same old message, IS this message -- the lower-numbered wave finishes it and writes both rows,
              // the other one takes every second useful cone (or the window loop while the first tests the keys).
              // (one scalar word: bit 0 sharing, bit 1 this wave is the helper, bits 4-7 the twin's number -- everything else
              //  is derived where it is used: more live scalars across this routine end up spilled into VGPR lanes)
              int coopw = (int)((twins >> (4 * j)) & 15u) << 4;
              // (this message's old row and its twin's are requested together, compared without branches, and the old row
              //  is used for H below: one LDS latency, a dozen instructions)
              double mold[4];
              {
                const int tw = (coopw >> 4) < kWideCompute ? (coopw >> 4) : j;
                double mtw[4];
#pragma unroll
                for (int c = 0; c < 4; ++c) { mold[c] = st[kWS + j * kWS + c * kWave + lane]; mtw[c] = st[kWS + tw * kWS + c * kWave + lane]; }
                const double alpha_tw = st[kWS + 8 * kWS + tw];
                bool differ = alpha_tw != alpha;
#pragma unroll
                for (int c = 0; c < 4; ++c) differ = differ | (WVALID(c) & (mtw[c] != mold[c]));
                const bool may = KERNEL == 1 && fast_msg && !p.lean && !constant && tw != j && nout <= kWideCompute && !(p.debug & 8192);   // (development switch 8192: no twins)
                if (may && !UNI(differ)) coopw |= 1 | (j > tw ? 2 : 0);
              }
#define coop_n ((coopw & 1) + 1)
#define coop_part ((coopw >> 1) & 1)
#define partner (coopw >> 4)
#define xd (L.scr + (j > partner ? j : partner) * kWScr)      /* the helper's own scratch: m1[4][64], m2[4][64], one int */
#define xfl (L.xflag + (j > partner ? j : partner))
              double h[4], hmin, hmax;
              {
                double hlo = inf, hhi = -inf;
#pragma unroll
                for (int c = 0; c < 4; ++c) h[c] = gamma * di[c] - mold[c];
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                  hlo = min_raw(hlo, WVALID(c) ? h[c] : inf); hhi = max_raw(hhi, WVALID(c) ? h[c] : -inf);
                  h[c] = WVALID(c) ? h[c] : inf;
                }
                wave_min_max_dpp(hlo, hhi);  // both reductions in one interleaved pass (two independent chains)
                hmin = hlo; hmax = hhi;
              }
              const double vtrunc = hmin + alpha * p.lambda;
              const double ap0 = alpha * pos_first, ap1 = alpha * pos_last;
              const double aplo = min_raw(ap0, ap1), aphi = max_raw(ap0, ap1);
              const double mag = max_raw(fabs(hmin), fabs(hmax)) + 2 * max_raw(fabs(ap0), fabs(ap1));
              const double delta = 1e-9 * (mag + fabs(alpha * p.lambda));
              double *scr = L.scr + wave * kWScr;
              WSTAMP(1);
              double out[4] = {0, 0, 0, 0}, vmin = 0;
              bool serial = !fast_msg;
              if (constant) {
                // typeStereoLinear.h:390-396: message = min H everywhere, normalised to zero
                serial = false;
#pragma unroll
                for (int c = 0; c < 4; ++c) out[c] = hmin;
                vmin = hmin;
              } else if (p.lean && KERNEL == 1) {
                // ---- STEREO_TRWS_MESSAGES_MINPLUS: the message is the plain min-plus, nothing else -------
                // (min over ALL sources = min over the useful ones, truncated: every other source costs
                //  >= vTrunc; no margins, no tangency test, no serial construction)
                serial = false;
                const int w = p.window;
                unsigned long long um[4];
                int nuse = 0;
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                  um[c] = __builtin_amdgcn_ballot_w64(WVALID(c) && h[c] < vtrunc);
                  nuse += __builtin_popcountll(um[c]);
                }
                double pq[4];
#pragma unroll
                for (int c = 0; c < 4; ++c) pq[c] = WPOS(c);
                double m1[4] = {inf, inf, inf, inf};
                if (nuse <= kWideSparse) {
#pragma unroll
                  for (int c = 0; c < 4; ++c) {
                    unsigned long long mk = um[c];
                    while (mk) {
                      const int l = __builtin_ctzll(mk);
                      mk &= mk - 1;
                      const double hi = readlane_f64(h[c], l), qi = readlane_f64(pq[c], l);
#pragma unroll
                      for (int cc = 0; cc < 4; ++cc) m1[cc] = min_raw(m1[cc], pair_cost<1>(alpha, pq[cc] - qi, hi));
                    }
                  }
                } else {
                  double2 *mtab = (double2 *)scr + kWPad;
                  double *htab = scr + kWPad;
                  if (lane < 2 * kWPad) {
                    if (uniform) scr[lane < kWPad ? lane : K + lane] = inf;
                    else ((double2 *)scr)[lane < kWPad ? lane : K + lane] = make_double2(inf, 0.0);
                  }
#pragma unroll
                  for (int c = 0; c < 4; ++c) {
                    if (WVALID(c)) {
                      if (uniform) htab[c * kWave + lane] = h[c];
                      else mtab[c * kWave + lane] = make_double2(h[c], pq[c]);
                    }
                  }
                  WSYNC();
                  if (uniform && w <= kWPad) {
                    for (int d = -w; d <= w; ++d) {
                      const double ad = alpha * fabs((double)d * ustep);  // == alpha |t - q| exactly
#pragma unroll
                      for (int c = 0; c < 4; ++c) m1[c] = min_raw(m1[c], ad + htab[c * kWave + lane + d]);
                    }
                  } else {
                    for (int d = -w; d <= w; ++d) {
#pragma unroll
                      for (int c = 0; c < 4; ++c) {
                        const int i = c * kWave + lane + d;
                        const double2 sv = mtab[w <= kWPad ? i : i < 0 ? 0 : i > K - 1 ? K - 1 : i];
                        double cst = pair_cost<1>(alpha, pq[c] - sv.y, sv.x);
                        if (w > kWPad) cst = (i >= 0 && i < K) ? cst : inf;
                        m1[c] = min_raw(m1[c], cst);
                      }
                    }
                  }
                  WSYNC();
                }
#pragma unroll
                for (int c = 0; c < 4; ++c) out[c] = m1[c] < vtrunc ? m1[c] : vtrunc;
                vmin = hmin;  // (destination t sees source t at distance 0, vTrunc >= min H)
                WSTAMP(3);
              } else if (fast_msg && KERNEL == 2) {
                // ---- truncated QUADRATIC kernel (typeStereoQuadratic.h:329-501), round 4 ------------------
                // The certificate of message_quad_fast (trws_dev.h) on four labels per lane: plain min-plus
                // over the useful parabolas (h < vTrunc), smallest and second smallest cost per destination;
                // if the smallest is delta-separated from every other cost and from vTrunc, and the rounding
                // of a breakpoint that involves it (<= ~7 eps G / (alpha gap)) stays below the slope margin
                // delta / (2 alpha Q), the reference's hull construction returns exactly that parabola's value.
                // No tangency test.  Shared strictly ascending positions: gap = the smallest distance of two
                // neighbouring positions, Q = their span, G bounded by max |h| + 2 alpha max pos^2 (a larger
                // scale only makes delta, and with it the certificate, more conservative).
                const int w = p.window;   // sources farther than sqrt(lambda (1 + 1e-9)) cost >= vTrunc bit for bit
                unsigned long long um[4];
                int nuse = 0;
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                  um[c] = __builtin_amdgcn_ballot_w64(WVALID(c) && h[c] < vtrunc);
                  nuse += __builtin_popcountll(um[c]);
                }
                double pq[4];
#pragma unroll
                for (int c = 0; c < 4; ++c) pq[c] = WPOS(c);
                const double pmax = max_raw(fabs(pos_first), fabs(pos_last));
                const double scale = max_raw(fabs(hmin), fabs(hmax)) + 2 * (alpha * pmax * pmax);
                const double qdelta = 1e-9 * (scale + fabs(alpha * p.lambda) + fabs(vtrunc));
                double m1[4] = {inf, inf, inf, inf}, m2[4] = {inf, inf, inf, inf};
                if (nuse <= kWideSparse) {
#pragma unroll
                  for (int c = 0; c < 4; ++c) {
                    unsigned long long mk = um[c];
                    while (mk) {
                      const int l = __builtin_ctzll(mk);
                      mk &= mk - 1;
                      const double hi = readlane_f64(h[c], l), qi = readlane_f64(pq[c], l);
#pragma unroll
                      for (int cc = 0; cc < 4; ++cc) {
                        const double cst = pair_cost<2>(alpha, pq[cc] - qi, hi);
                        const double lo_ = min_raw(m1[cc], cst), hi_ = max_raw(m1[cc], cst);
                        m2[cc] = min_raw(m2[cc], hi_);   // second smallest; equal costs of two sources count
                        m1[cc] = lo_;
                      }
                    }
                  }
                } else {
                  double2 *mtab = (double2 *)scr + kWPad;
                  if (lane < 2 * kWPad) ((double2 *)scr)[lane < kWPad ? lane : K + lane] = make_double2(inf, 0.0);
#pragma unroll
                  for (int c = 0; c < 4; ++c)
                    if (WVALID(c)) mtab[c * kWave + lane] = make_double2(h[c], pq[c]);
                  WSYNC();
                  for (int d = -w; d <= w; ++d) {
                    double2 sv[4];
#pragma unroll
                    for (int c = 0; c < 4; ++c) {
                      const int i = c * kWave + lane + d;
                      sv[c] = mtab[w <= kWPad ? i : i < 0 ? 0 : i > K - 1 ? K - 1 : i];
                    }
#pragma unroll
                    for (int c = 0; c < 4; ++c) {
                      const int i = c * kWave + lane + d;
                      double cst = pair_cost<2>(alpha, pq[c] - sv[c].y, sv[c].x);
                      if (w > kWPad) cst = (i >= 0 && i < K) ? cst : inf;
                      const double lo_ = min_raw(m1[c], cst), hi_ = max_raw(m1[c], cst);
                      m2[c] = min_raw(m2[c], hi_);
                      m1[c] = lo_;
                    }
                  }
                  WSYNC();
                }
                bool bad = !(qdelta < inf) || !(alpha > 0) || !(p.pos_gap > 4e-8);
                bad = bad || !(1e-13 * scale * (pos_last - pos_first) < qdelta * p.pos_gap);
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                  if (WVALID(c)) {
                    bad = bad || (m1[c] < vtrunc && !(m2[c] - m1[c] > qdelta && vtrunc - m1[c] > qdelta));
                    out[c] = m1[c] < vtrunc ? m1[c] : vtrunc;
                  }
                }
                vmin = hmin;  // (destination t sees source t at distance 0; every other cost is some h plus a square)
                serial = UNI(bad);
                WSTAMP(3);
              } else if (fast_msg) {
                // ---- the certified path ---------------------------------------------------------------
                // Only USEFUL cones (h < vTrunc; bit masks um) can give a destination a cost below
                // vTrunc, and only pairs with a useful cone matter to the certificate (the test of
                // trws_pipe_kernel's path).  With up to kWideSparse of them -- five to eight of 256 on
                // real volumes -- nothing else is looked at: for useful cone i every lane forms the cost
                // its four destinations get from i (the reference's own expression), keeps the smallest
                // and second smallest cost per destination, and counts the destinations t where that
                // cost is within delta of h_t, i.e. where cone t lies on an arm of cone i (u_t = u_i
                // to the right of i, v_t = v_i to its left -- a tangency; t = i itself matches, once).
                // With more useful cones (flat H): the dense windowed min-plus from a padded source
                // table, and the conservative all-pairs closest-pair test on u and on v by bucketing.
                const int w = p.window;
                unsigned long long um[4];
                int nuse = 0;
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                  um[c] = __builtin_amdgcn_ballot_w64(WVALID(c) && h[c] < vtrunc);
                  nuse += __builtin_popcountll(um[c]);
                }
                double pq[4];
#pragma unroll
                for (int c = 0; c < 4; ++c) pq[c] = WPOS(c);
                double m1[4] = {inf, inf, inf, inf}, m2[4] = {inf, inf, inf, inf};
                bool bad = false;
                if (!(delta < inf)) bad = true;  // no finite scale: the serial construction decides
                else if (nuse <= kWideSparse) {
                  int matches = 0;
#ifdef STEREO_WIDE_FP32_CONES
                  // MEASUREMENT FLAVOUR ONLY (tools/gpu_fp32_cones.sh): the useful-cone loop in fp32 --
                  // costs, both minima and the match test on single-precision copies of H and the positions
                  // (SURVEY 8(d) configs 4 / 5 allow fp32 messages).  Not the reference's bits and not a product mode:
                  // built into its own library to put a number on "would fp32 messages be faster here".
                  float hf[4], pf[4], m1f[4], m2f[4];
                  const float af = (float)alpha, df = (float)delta + 1e-6f * ((float)mag + 1.0f);
#pragma unroll
                  for (int c = 0; c < 4; ++c) { hf[c] = (float)h[c]; pf[c] = (float)pq[c]; m1f[c] = __builtin_huge_valf(); m2f[c] = __builtin_huge_valf(); }
#pragma unroll
                  for (int c = 0; c < 4; ++c) {
                    unsigned long long mk = um[c];
                    while (mk) {
                      const int l = __builtin_ctzll(mk);
                      mk &= mk - 1;
                      const float hi = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, hf[c]), l));
                      const float qi = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, pf[c]), l));
#pragma unroll
                      for (int cc = 0; cc < 4; ++cc) {
                        const float cst = af * __builtin_fabsf(pf[cc] - qi) + hi;
                        const float lo_ = __builtin_fminf(m1f[cc], cst), hi_ = __builtin_fmaxf(m1f[cc], cst);
                        m2f[cc] = __builtin_fminf(m2f[cc], hi_);
                        m1f[cc] = lo_;
                        matches += __builtin_popcountll(__builtin_amdgcn_ballot_w64(__builtin_fabsf(cst - hf[cc]) <= df));
                      }
                    }
                  }
#pragma unroll
                  for (int c = 0; c < 4; ++c) { m1[c] = (double)m1f[c]; m2[c] = (double)m2f[c] + (double)df * 4; }
                  bad = false; (void)matches;   // (no serial fallbacks in the measurement: fp32 cannot decide the certificate)
#else
                  // (two waves: every second cone each, from three cones on -- an exchange costs about one cone)
                  const bool split = coop_n > 1 && nuse > ((p.debug & 4096) ? 1000 : 2);   // (development switch 4096: twins found, nothing shared)
                  if (wprof) { pacc[6] += split ? 1000 : 0; pacc[7] += coop_n > 1 ? 1000 : 0; pacc[8] += 1000ull * nuse; }   // (development profile: per mille of wave 0's messages)
                  if (coop_part && !split) continue;   // (nothing to share: the first wave does it alone)
                  if (split) {
                    // (its own copy of the loop: the one below keeps the shape the compiler schedules best)
                    int turn = 0;
#pragma unroll
                    for (int c = 0; c < 4; ++c) {
                      unsigned long long mk = um[c];
                      while (mk) {
                        const int l = __builtin_ctzll(mk);
                        mk &= mk - 1;
                        const bool skip = turn != coop_part;
                        turn ^= 1;
                        if (skip) continue;
                        const double hi = readlane_f64(h[c], l), qi = readlane_f64(pq[c], l);
#pragma unroll
                        for (int cc = 0; cc < 4; ++cc) {
                          const double cst = pair_cost<1>(alpha, pq[cc] - qi, hi);
                          const double lo_ = min_raw(m1[cc], cst), hi_ = max_raw(m1[cc], cst);
                          m2[cc] = min_raw(m2[cc], hi_);
                          m1[cc] = lo_;
                          matches += __builtin_popcountll(__builtin_amdgcn_ballot_w64(fabs(cst - h[cc]) <= delta));
                        }
                      }
                    }
                    if (coop_part) { wide_publish(xd, xfl, pos + 1, m1, m2, matches, lane); continue; }
                    wide_collect(xd, xfl, pos + 1, m1, m2, matches, lane, L.ctl + 1);
                  } else {
#pragma unroll
                  for (int c = 0; c < 4; ++c) {
                    unsigned long long mk = um[c];
                    while (mk) {
                      const int l = __builtin_ctzll(mk);
                      mk &= mk - 1;
                      const double hi = readlane_f64(h[c], l), qi = readlane_f64(pq[c], l);
#pragma unroll
                      for (int cc = 0; cc < 4; ++cc) {
                        const double cst = pair_cost<1>(alpha, pq[cc] - qi, hi);
                        const double lo_ = min_raw(m1[cc], cst), hi_ = max_raw(m1[cc], cst);
                        m2[cc] = min_raw(m2[cc], hi_);
                        m1[cc] = lo_;
                        matches += __builtin_popcountll(__builtin_amdgcn_ballot_w64(fabs(cst - h[cc]) <= delta));
                      }
                    }
                  }
                  }
                  bad = matches != nuse;
#endif
                } else {
                  bool crowded = false;
                  if (!coop_part) {
                    double r[4];
#pragma unroll
                    for (int c = 0; c < 4; ++c) r[c] = h[c] - alpha * pq[c];
                    crowded = keys_within(r, K, C, delta, hmin - aphi, (hmax - aplo) - (hmin - aphi), scr, lane);
                    WSYNC();
                    if (!crowded) {
#pragma unroll
                      for (int c = 0; c < 4; ++c) r[c] = h[c] + alpha * pq[c];
                      crowded = keys_within(r, K, C, delta, hmin + aplo, (hmax + aphi) - (hmin + aplo), scr, lane);
                      WSYNC();
                    }
                  }
                  if (crowded) {
                    // the all-pairs test found two cones close together: only pairs with a useful
                    // cone matter (flat H with more than 64 of them: the serial construction decides)
                    double uu[4], vv[4];
#pragma unroll
                    for (int c = 0; c < 4; ++c) {
                      const double aq = alpha * pq[c];
                      uu[c] = h[c] - aq; vv[c] = h[c] + aq;
                    }
                    bad = nuse > 64 || useful_cone_ties(um, 1, 0, uu, vv, delta);
                  }
                  // (two waves: the first tests the keys, above; the helper walks the window, below)
                  if (coop_n > 1 && !coop_part) {
                    int none = 0;
                    wide_collect(xd, xfl, pos + 1, m1, m2, none, lane, L.ctl + 1);
                  } else {
                  // source table (the scratch again): (h, q) pairs at index kWPad + k -- h only on uniform
                  // positions --, (+inf, 0) padding on both sides
                  double2 *mtab = (double2 *)scr + kWPad;
                  double *htab = scr + kWPad;
                  if (lane < 2 * kWPad) {
                    if (uniform) scr[lane < kWPad ? lane : K + lane] = inf;
                    else ((double2 *)scr)[lane < kWPad ? lane : K + lane] = make_double2(inf, 0.0);
                  }
#pragma unroll
                  for (int c = 0; c < 4; ++c) {
                    if (WVALID(c)) {
                      if (uniform) htab[c * kWave + lane] = h[c];
                      else mtab[c * kWave + lane] = make_double2(h[c], pq[c]);
                    }
                  }
                  WSYNC();
                  if (uniform && w <= kWPad) {
                    for (int d = -w; d <= w; ++d) {
                      const double ad = alpha * fabs((double)d * ustep);  // == alpha |t - q| exactly
                      double hs[4];
#pragma unroll
                      for (int c = 0; c < 4; ++c) hs[c] = htab[c * kWave + lane + d];
#pragma unroll
                      for (int c = 0; c < 4; ++c) {
                        const double cst = ad + hs[c];
                        const double lo_ = min_raw(m1[c], cst), hi_ = max_raw(m1[c], cst);
                        m2[c] = min_raw(m2[c], hi_);
                        m1[c] = lo_;
                      }
                    }
                  } else {
                    for (int d = -w; d <= w; ++d) {
                      double2 sv[4];
#pragma unroll
                      for (int c = 0; c < 4; ++c) {
                        const int i = c * kWave + lane + d;
                        sv[c] = mtab[w <= kWPad ? i : i < 0 ? 0 : i > K - 1 ? K - 1 : i];
                      }
#pragma unroll
                      for (int c = 0; c < 4; ++c) {
                        const int i = c * kWave + lane + d;
                        double cst = pair_cost<1>(alpha, pq[c] - sv[c].y, sv[c].x);
                        if (w > kWPad) cst = (i >= 0 && i < K) ? cst : inf;
                        const double lo_ = min_raw(m1[c], cst), hi_ = max_raw(m1[c], cst);
                        m2[c] = min_raw(m2[c], hi_);
                        m1[c] = lo_;
                      }
                    }
                  }
                  WSYNC();
                  if (coop_part) { wide_publish(xd, xfl, pos + 1, m1, m2, 0, lane); continue; }
                  }
                }
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                  if (WVALID(c)) {
                    bad = bad || (m1[c] < vtrunc && !(m2[c] - m1[c] > delta && vtrunc - m1[c] > delta));
                    out[c] = m1[c] < vtrunc ? m1[c] : vtrunc;
                  }
                }
#ifdef STEREO_WIDE_FP32_CONES
                if (nuse <= kWideSparse) bad = false;   // (measurement flavour: margins of 1e-9 mean nothing in fp32)
#endif
                // the smallest entry of a min-plus message on shared positions is min H itself
                // (destination t sees source t at distance 0; vTrunc >= min H): no reduction
                vmin = hmin;
                serial = UNI(bad);
                WSTAMP(3);
              }
              if (serial) {
                // the reference's serial construction in LDS; the stack lives in a scratch
                // shared by the workgroup (rare path): take its lock
                if (lane == 0 && p.fallbacks) atomicAdd(p.fallbacks, 1);
                if (lane == 0) {
                  int spins = 0;
                  while (__hip_atomic_exchange(L.ctl + 2, 1, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) != 0) {
                    __builtin_amdgcn_s_sleep(2);
                    if (++spins > kSpinLimit) { L.ctl[1] = 1; break; }
                  }
                }
                WSYNC();
                double *sh = L.fb, *sq = L.fb + kWS, *z = L.fb + 2 * kWS, *Hs = L.fb + 3 * kWS;
#pragma unroll
                for (int c = 0; c < 4; ++c)
                  if (WVALID(c)) Hs[c * kWave + lane] = h[c];
                WSYNC();
                if (lane == 0) build_envelope<KERNEL>(K, alpha, Hs, L.pos, sh, sq, z);
                WSYNC();
                double vloc = inf;
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                  const int k = c * kWave + lane;
                  if (c < C && k < K) {
                    int jj = 0;
                    while (z[jj + 1] < WPOS(c)) ++jj;
                    const double cst = pair_cost<KERNEL>(alpha, WPOS(c) - sq[jj], sh[jj]);
                    out[c] = cst < vtrunc ? cst : vtrunc;
                    vloc = min_raw(vloc, out[c]);
                  }
                }
                WSYNC();
                if (lane == 0) __hip_atomic_store(L.ctl + 2, 0, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
                vmin = wave_min_dpp(vloc);
                WSTAMP(4);
              }
#pragma unroll
              for (int c = 0; c < 4; ++c) {
                const int k = c * kWave + lane;
                if (c < C && k < K) {
                  hcur[j * kWS + k] = out[c] - vmin;
                  if (coop_n > 1) hcur[partner * kWS + k] = out[c] - vmin;
                }
              }
              if (BACKWARD && lane == 0) { sc[j] = vmin; if (coop_n > 1) sc[partner] = vmin; }
#undef coop_n
#undef coop_part
#undef partner
#undef xd
#undef xfl
              WSTAMP(5);
            }
          }
        }
      WIDE_VISITS_END
    } else if (wave == kWideCompute && (K & 1) == 0) {
      // ======================================================== loader A: own data, two visits deep
      // During visit pos the registers hold node pos + 1's unary, previous-sweep messages and weight,
      // requested during visit pos - 1: the HBM latency of one node lies behind the whole visit of
      // the node before it instead of inside its own.  They go to the stage, then node pos + 2's
      // requests go out and stay in flight across the barrier (16 bytes per lane: labels 2 lane, 2 lane + 1
      // of each half row; K is even here).
      typedef int wide_v4i __attribute__((ext_vector_type(4)));
      const wide_v4i zero4 = {0, 0, 0, 0};
      wide_v4i rd0 = zero4, rd1 = zero4, rm[8][2];
#pragma unroll
      for (int j = 0; j < 8; ++j) { rm[j][0] = zero4; rm[j][1] = zero4; }
      double av = 0;
      const bool ok0 = 2 * lane < K, ok1 = 2 * kWave + 2 * lane < K;
      // descriptors are fetched three nodes ahead: every load of a visit is waited for at the top of the
      // next one, so a descriptor requested at the end of a visit would put its latency there
      int w1 = desc[(size_t)p0 * DW + lane];                                  // the node in the registers
      int w2 = p0 + 1 < p1 ? desc[(size_t)(p0 + 1) * DW + lane] : 0;          // the one after it
      int w3 = p0 + 2 < p1 ? desc[(size_t)(p0 + 2) * DW + lane] : 0;
#define WIDE_LOAD16(DST, PTR, OFF) DST = *(const wide_v4i *)((const char *)(PTR) + (OFF))  /* plain loads: in flight across the barrier, waited for at their first use */
#define WIDE_REQUEST_OWN(W)                                                                               \
      do {                                                                                                \
        const NodeDesc rq = decode_desc(W);                                                               \
        const double *ub = p.unary + (size_t)rq.node * K + 2 * lane;                                      \
        if (ok0) WIDE_LOAD16(rd0, ub, 0);                                                                 \
        if (ok1) WIDE_LOAD16(rd1, ub, 1024);                                                              \
        _Pragma("unroll") for (int j = 0; j < 8; ++j) {                                                   \
          if (j < rq.nout && (UPDATE || PRIMAL)) {                                                        \
            const double *mb = p.msg + (size_t)rq.e[j] * K + 2 * lane;                                    \
            if (ok0) WIDE_LOAD16(rm[j][0], mb, 0);                                                        \
            if (ok1) WIDE_LOAD16(rm[j][1], mb, 1024);                                                     \
          }                                                                                               \
        }                                                                                                 \
        av = 0;                                                                                           \
        if (lane < rq.nout + rq.nin) {                                                                    \
          int ej = 0;                                                                                     \
          _Pragma("unroll") for (int j = 0; j < 8; ++j)                                                   \
            if (lane == j) ej = rq.e[j];                                                                  \
          av = p.alpha[ej];                                                                               \
        }                                                                                                 \
      } while (0)
      WIDE_REQUEST_OWN(w1);
      WIDE_VISITS_BEGIN
        if (pos + 1 >= p0 && pos + 1 < p1) {
          const int w = w1;
          const NodeDesc nx = decode_desc(w);
          int *stni = (int *)(stn + kWStI);
          // (first visit of a speculative segment: the rows of the node in front come from the runner, behind the
          //  segment's flag -- to everybody else in the workgroup they look like rows of another run)
          const bool sfirst = SPEC && seg > 0 && pos + 1 == p0;
          stni[lane] = (sfirst && lane >= 12 && lane < 20 && w >= 0) ? -1 : w;
          if (SPEC && UPDATE && seg >= 0 && !second_walk) {
            // a speculative segment keeps the rows its visits overwrite: a second walk starts from them
#pragma unroll
            for (int j = 0; j < 4; ++j) {
              if (j < nx.nout) {
                typedef double wide_v2d_ __attribute__((ext_vector_type(2)));
                double *ub = p.spec_undo + ((size_t)(seg * p.spec_max_len + (pos + 1 - p0)) * 4 + j) * K + 2 * lane;
                const wide_v2d_ lo_ = __builtin_bit_cast(wide_v2d_, rm[j][0]), hi_ = __builtin_bit_cast(wide_v2d_, rm[j][1]);
                if (ok0) { st_sc1(ub, lo_.x); st_sc1(ub + 1, lo_.y); }
                if (ok1) { st_sc1(ub + 2 * kWave, hi_.x); st_sc1(ub + 2 * kWave + 1, hi_.y); }
              }
            }
          }
          {
            // where the compute waves find the node's message rows at visit pos + 1 (offsets into the
            // workgroup's LDS, in doubles): a message handed over inside the run sits in the ring of
            // the last two visits, everything else in this stage -- decided here, once, instead of by
            // every compute wave in scalar code on its critical path
            int sl = -1;
#pragma unroll
            for (int j = 0; j < 8; ++j)
              if (lane == j) sl = (j >= nx.nout && !sfirst) ? nx.slot[j] : -1;
            const int hb1n = ((pos % 3) + 3) % 3, hb2n = (((pos - 1) % 3) + 3) % 3;  // hprev / hprev2 of visit pos + 1
            const int row = sl >= 8 ? (int)(L.hand - lds) + hb2n * 8 * kWS + (sl - 8) * kWS
                          : sl >= 0 ? (int)(L.hand - lds) + hb1n * 8 * kWS + sl * kWS
                                    : (int)(stn - lds) + kWS + lane * kWS;
            if (lane < 8) stni[72 + lane] = row;
            {
              const int kin = nx.nout + (lane & 7);
              const int rk = __shfl(row, kin & 7, kWave);
              if (lane < 8) stni[80 + lane] = (kin < nx.nout + nx.nin && kin < 8) ? rk : (int)(L.zrow - lds);
            }
          }
          L.dring[((pos + 1) % 3) * 64 + lane] = w;
          if (ok0) *(wide_v4i *)(stn + 2 * lane) = rd0;
          if (ok1) *(wide_v4i *)(stn + 2 * kWave + 2 * lane) = rd1;
          // S = D + rows 0 .. nout - 1, added in list order: the prefix of Di that is known a visit ahead
          typedef double wide_v2d __attribute__((ext_vector_type(2)));
          wide_v2d s0 = __builtin_bit_cast(wide_v2d, rd0), s1 = __builtin_bit_cast(wide_v2d, rd1);
#pragma unroll
          for (int j = 0; j < 8; ++j) {
            if (j < nx.nout) {
              if (ok0) *(wide_v4i *)(stn + kWS + j * kWS + 2 * lane) = rm[j][0];
              if (ok1) *(wide_v4i *)(stn + kWS + j * kWS + 2 * kWave + 2 * lane) = rm[j][1];
              s0 += __builtin_bit_cast(wide_v2d, rm[j][0]);
              s1 += __builtin_bit_cast(wide_v2d, rm[j][1]);
            }
          }
          if (ok0) *(wide_v2d *)(stn + kWStS + 2 * lane) = s0;
          if (ok1) *(wide_v2d *)(stn + kWStS + 2 * kWave + 2 * lane) = s1;
          if (lane < 8) stn[kWS + 8 * kWS + lane] = av;
          if (lane == 0) stn[kWStG] = (double)1 / (double)(nx.nout > nx.nin ? nx.nout : nx.nin);
          int w4 = 0;
          if (pos + 4 < p1) w4 = desc[(size_t)(pos + 4) * DW + lane];
          w1 = w2; w2 = w3; w3 = w4;
          if (pos + 2 < p1) WIDE_REQUEST_OWN(w1);
        }
      WIDE_VISITS_END
#undef WIDE_REQUEST_OWN
#undef WIDE_LOAD16
    } else if (wave == kWideCompute) {
      // loader A for odd K (no 16-byte alignment of the rows): node pos + 1 during visit pos
      int wnext = desc[(size_t)p0 * DW + lane];
      WIDE_VISITS_BEGIN
        // ======================================================== loader A: node pos + 1, own data
        if (pos + 1 >= p0 && pos + 1 < p1) {
          const int w = wnext;
          if (pos + 2 < p1) wnext = desc[(size_t)(pos + 2) * DW + lane];
          const NodeDesc nx = decode_desc(w);
          int *stni = (int *)(stn + kWStI);
          stni[lane] = w;
          {
            // where the compute waves find the node's message rows at visit pos + 1 (offsets into the
            // workgroup's LDS, in doubles): a message handed over inside the run sits in the ring of
            // the last two visits, everything else in this stage -- decided here, once, instead of by
            // every compute wave in scalar code on its critical path
            int sl = -1;
#pragma unroll
            for (int j = 0; j < 8; ++j)
              if (lane == j) sl = j >= nx.nout ? nx.slot[j] : -1;
            const int hb1n = ((pos % 3) + 3) % 3, hb2n = (((pos - 1) % 3) + 3) % 3;  // hprev / hprev2 of visit pos + 1
            const int row = sl >= 8 ? (int)(L.hand - lds) + hb2n * 8 * kWS + (sl - 8) * kWS
                          : sl >= 0 ? (int)(L.hand - lds) + hb1n * 8 * kWS + sl * kWS
                                    : (int)(stn - lds) + kWS + lane * kWS;
            if (lane < 8) stni[72 + lane] = row;
            {
              const int kin = nx.nout + (lane & 7);
              const int rk = __shfl(row, kin & 7, kWave);
              if (lane < 8) stni[80 + lane] = (kin < nx.nout + nx.nin && kin < 8) ? rk : (int)(L.zrow - lds);
            }
          }
          L.dring[((pos + 1) % 3) * 64 + lane] = w;
          const int ntot = nx.nout + nx.nin;
          // all requests go out before anything is consumed (registers first, LDS at the end)
          double dk[4], mv[8][4];
          bool okc[4];
#pragma unroll
          for (int c = 0; c < 4; ++c) {
            okc[c] = c < C && c * kWave + lane < K;
            dk[c] = okc[c] ? p.unary[(size_t)nx.node * K + c * kWave + lane] : 0.0;
          }
#pragma unroll
          for (int j = 0; j < 8; ++j) {
#pragma unroll
            for (int c = 0; c < 4; ++c) {
              mv[j][c] = 0;
              if (j < nx.nout && (UPDATE || PRIMAL) && okc[c]) mv[j][c] = p.msg[(size_t)nx.e[j] * K + c * kWave + lane];
            }
          }
          double av = 0;
          if (lane < ntot) {
            int ej = 0;
#pragma unroll
            for (int j = 0; j < 8; ++j)
              if (lane == j) ej = nx.e[j];
            av = p.alpha[ej];
          }
#pragma unroll
          for (int c = 0; c < 4; ++c) {
            if (okc[c]) {
              stn[c * kWave + lane] = dk[c];
              double sum = dk[c];  // S = D + rows 0 .. nout - 1 in list order
#pragma unroll
              for (int j = 0; j < 8; ++j)
                if (j < nx.nout) { stn[kWS + j * kWS + c * kWave + lane] = mv[j][c]; sum += mv[j][c]; }
              stn[kWStS + c * kWave + lane] = sum;
            }
          }
          if (lane < 8) stn[kWS + 8 * kWS + lane] = av;
          if (lane == 0) stn[kWStG] = (double)1 / (double)(nx.nout > nx.nin ? nx.nout : nx.nin);
        }
      WIDE_VISITS_END
    } else if (wave == kWideCompute + 1 && (K & 1) == 0) {
      // ======================================================== loader B: data behind flags, two visits deep
      // During visit pos the registers hold node pos + 1's foreign messages and labels; they go to the
      // stage, then this wave waits for node pos + 2's flags and sends its requests, which stay in
      // flight across the barrier: flag round trip and fetch round trip of a node overlap with the
      // visit before its own instead of adding up inside it.  Only where the descriptor allows it
      // (bit 12 of word 2, trws_graph.cpp): what is waited for must not depend on the nodes this
      // workgroup has not yet made visible; elsewhere (the interleaved last rows) the node is fetched
      // during the visit before its own, as loader B always did.
      typedef int wide_v4i __attribute__((ext_vector_type(4)));
      typedef double wide_v2d __attribute__((ext_vector_type(2)));
      const wide_v4i zero4 = {0, 0, 0, 0};
      wide_v4i rb[8][2];
#pragma unroll
      for (int j = 0; j < 8; ++j) { rb[j][0] = zero4; rb[j][1] = zero4; }
      int pxv = 0;
      bool parked = false;  // the registers hold node pos + 1's data
      const bool ok0 = 2 * lane < K, ok1 = 2 * kWave + 2 * lane < K;
      int w1 = desc[(size_t)p0 * DW + lane];
      int w2 = p0 + 1 < p1 ? desc[(size_t)(p0 + 1) * DW + lane] : 0;
      int w3 = p0 + 2 < p1 ? desc[(size_t)(p0 + 2) * DW + lane] : 0;  // three nodes ahead, as in loader A
#define WIDE_LOAD16_SC1(DST, PTR, OFF)                                                                     \
      do {                                                                                                \
        const double lo_ = ld_sc1((const double *)((const char *)(PTR) + (OFF)));                         \
        const double hi_ = ld_sc1((const double *)((const char *)(PTR) + (OFF)) + 1);                     \
        wide_v2d pr_; pr_.x = lo_; pr_.y = hi_;                                                           \
        DST = __builtin_bit_cast(wide_v4i, pr_);                                                          \
      } while (0)
// (SF: the first node of a speculative segment -- the rows and the label of the node in front (slots >= 0) are the
//  runner's, behind the segment's flag; a second walk takes them from the messages themselves: the segment in front has
//  committed)
#define WIDE_REQUEST_FOREIGN(W, SF)                                                                       \
      do {                                                                                                \
        const NodeDesc rq = decode_desc(W);                                                               \
        const int rtot = rq.nout + rq.nin;                                                                \
        int xn = 0, sl = 0;                                                                               \
        if (lane < rtot) {                                                                                \
          _Pragma("unroll") for (int j = 0; j < 8; ++j)                                                   \
            if (lane == j) { xn = rq.xn[j]; sl = rq.slot[j]; }                                            \
        }                                                                                                 \
        if (SPEC && (SF) && !wait_flag(p, p.N + seg, epoch, -2)) { if (lane == 0) L.ctl[1] = 1; }         \
        wait_for_dependencies(p, rq.ndep, rq.dep[0], rq.dep[1], rq.dep[2], rq.dep[3], rq.rank, epoch, lane, L.ctl + 1);                                             \
        _Pragma("unroll") for (int j = 0; j < 8; ++j) {                                                   \
          if (UPDATE && j >= rq.nout && j < rtot && (rq.slot[j] < 0 || (SPEC && (SF)))) {                 \
            const double *mb = (SPEC && (SF) && spec_in && rq.slot[j] >= 0) ? p.spec_rows + ((size_t)seg * 8 + j) * K + 2 * lane \
                                                                            : p.msg + (size_t)rq.e[j] * K + 2 * lane;            \
            if (ok0) WIDE_LOAD16_SC1(rb[j][0], mb, 0);                                                    \
            if (ok1) WIDE_LOAD16_SC1(rb[j][1], mb, 1024);                                                 \
          }                                                                                               \
        }                                                                                                 \
        pxv = 0;                                                                                          \
        if (PRIMAL && lane < rtot && lane >= rq.nout && (sl < 0 || (SPEC && (SF))))                       \
          pxv = ld_sc1((SPEC && (SF) && spec_in && sl >= 0) ? p.spec_x + seg : p.x + xn);                 \
      } while (0)
#ifdef STEREO_HIP_MESSAGE_PROFILE
      unsigned long long lbacc[6] = {0, 0, 0, 0, 0, 0};
#define LBSTAMP(i) do { if (wprof) { const long long n_ = (long long)__builtin_readcyclecounter(); lbacc[i] += (unsigned long long)(n_ - lbm); lbm = n_; } } while (0)
#define LBCOUNT(i) do { if (wprof) lbacc[i] += 1; } while (0)
#else
#define LBSTAMP(i) do { } while (0)
#define LBCOUNT(i) do { } while (0)
#endif
      WIDE_VISITS_BEGIN
#ifdef STEREO_HIP_MESSAGE_PROFILE
        long long lbm = (long long)__builtin_readcyclecounter();
#endif
        if (pos + 1 >= p0 && pos + 1 < p1) {
          const bool sfirst = SPEC && seg > 0 && pos + 1 == p0;
          if (!parked) { WIDE_REQUEST_FOREIGN(w1, sfirst); LBSTAMP(0); LBCOUNT(4); }
          const NodeDesc nx = decode_desc(w1);
          int *stni = (int *)(stn + kWStI);
          const int ntot = nx.nout + nx.nin;
#pragma unroll
          for (int j = 0; j < 8; ++j) {
            if (UPDATE && j >= nx.nout && j < ntot && (nx.slot[j] < 0 || sfirst)) {
              if (ok0) *(wide_v4i *)(stn + kWS + j * kWS + 2 * lane) = rb[j][0];
              if (ok1) *(wide_v4i *)(stn + kWS + j * kWS + 2 * kWave + 2 * lane) = rb[j][1];
            }
          }
          if (lane < 8) stni[64 + lane] = pxv;
#ifdef STEREO_HIP_MESSAGE_PROFILE
          asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
#endif
          LBSTAMP(1);
          int w4 = 0;
          if (pos + 4 < p1) w4 = desc[(size_t)(pos + 4) * DW + lane];
          w1 = w2; w2 = w3; w3 = w4;
          parked = pos + 2 < p1 && ((__builtin_amdgcn_readlane(w1, 2) >> 12) & 1) != 0;
          if (parked) { WIDE_REQUEST_FOREIGN(w1, false); LBSTAMP(2); LBCOUNT(5); }
        }
      WIDE_VISITS_END
#ifdef STEREO_HIP_MESSAGE_PROFILE
      if (wprof && lane == 0) for (int i = 0; i < 6; ++i) atomicAdd(p.prof + 24 + i, lbacc[i]);
#endif
#undef LBSTAMP
#undef LBCOUNT
#undef WIDE_REQUEST_FOREIGN
#undef WIDE_LOAD16_SC1
    } else if (wave == kWideCompute + 1) {
      // loader B for odd K (no 16-byte alignment of the rows): node pos + 1 during visit pos
      int wnext = desc[(size_t)p0 * DW + lane];
      WIDE_VISITS_BEGIN
        // ======================================================== loader B: node pos + 1, data behind flags
        if (pos + 1 >= p0 && pos + 1 < p1) {
          const int w = wnext;
          if (pos + 2 < p1) wnext = desc[(size_t)(pos + 2) * DW + lane];
          const NodeDesc nx = decode_desc(w);
          int *stni = (int *)(stn + kWStI);
          const int ntot = nx.nout + nx.nin;
          int pxv = 0, xn = 0, sl = 0;
          if (lane < ntot) {
#pragma unroll
            for (int j = 0; j < 8; ++j)
              if (lane == j) { xn = nx.xn[j]; sl = nx.slot[j]; }
          }
          wait_for_dependencies(p, nx.ndep, nx.dep[0], nx.dep[1], nx.dep[2], nx.dep[3], nx.rank, epoch, lane, L.ctl + 1);
          double mv[8][4];
#pragma unroll
          for (int j = 0; j < 8; ++j) {
#pragma unroll
            for (int c = 0; c < 4; ++c) {
              mv[j][c] = 0;
              if (UPDATE && j >= nx.nout && j < ntot && nx.slot[j] < 0 && c < C && c * kWave + lane < K)
                mv[j][c] = ld_sc1(p.msg + (size_t)nx.e[j] * K + c * kWave + lane);
            }
          }
          if (PRIMAL && lane < ntot && lane >= nx.nout && sl < 0) pxv = ld_sc1(p.x + xn);
#pragma unroll
          for (int j = 0; j < 8; ++j) {
            if (UPDATE && j >= nx.nout && j < ntot && nx.slot[j] < 0) {
#pragma unroll
              for (int c = 0; c < 4; ++c)
                if (c < C && c * kWave + lane < K) stn[kWS + j * kWS + c * kWave + lane] = mv[j][c];
            }
          }
          if (lane < 8) stni[64 + lane] = pxv;
        }
      WIDE_VISITS_END
    } else if (wave == kWideCompute + 2) {
      WIDE_VISITS_BEGIN
        // ======================================================== storer: node pos - 1
        if (pos - 1 >= p0) {
          const NodeDesc pd = decode_desc(L.dring[((pos - 1) % 3) * 64 + lane]);
          const double *scp = L.scal + ((pos + 1) & 1) * kScalDoubles;
          if (UPDATE) {
#pragma unroll
            for (int j = 0; j < 8; ++j) {
              if (j < pd.nout) {
                double *mb = ((pd.remote >> j) & 1) ? (((pd.remote >> (8 + j)) & 1) ? p.peer_msg1 : p.peer_msg0) : p.msg;
                const int ej = ((pd.remote >> j) & 1) ? pd.re[j] : pd.e[j];  // the neighbour numbers the edge itself
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                  const int k = c * kWave + lane;
                  if (c < C && k < K) st_sc1(mb + (size_t)ej * K + k, hprev[j * kWS + k]);
                }
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
          asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
          if (lane == 0) {
            if (!(SPEC && seg >= 0)) st_sc1(p.done + pd.rank, epoch);   // (a speculative segment raises its flags when it commits)
            if (pd.remote & (1 << 16)) st_sc1(p.peer_done0 + pd.pn[0], epoch);
            if (pd.remote & (1 << 17)) st_sc1(p.peer_done1 + pd.pn[1], epoch);
          }
        }
      WIDE_VISITS_END
    } else {
      int xprev = 0, xprev2 = 0;
      WIDE_VISITS_BEGIN
        // ======================================================== primal of node pos
        if (PRIMAL && have_node) {
          const int *sti = (const int *)(st + kWStI);
          const int f = __builtin_amdgcn_readfirstlane(sti[2]);
          const int nout = f & 15, nin = (f >> 4) & 15, md = (f >> 16) & 255, ntot = nout + nin;
          double db[4], di[4], pq[4];
#pragma unroll
          for (int c = 0; c < 4; ++c) {
            const int k = c * kWave + lane;
            db[c] = (c < C && k < K) ? st[k] : inf;
            pq[c] = WPOS(c);
          }
#pragma unroll
          for (int jj = 0; jj < 8; ++jj) {
            if (jj >= nout && jj < ntot) {
              const int sl = __builtin_amdgcn_readfirstlane(sti[12 + jj]);
              const int ks = sl >= 8 ? xprev2 : sl >= 0 ? xprev : __builtin_amdgcn_readfirstlane(sti[64 + jj]);
              const double pks = L.pos[ks], aj = st[kWS + 8 * kWS + jj];
              const bool fwd = ((md >> jj) & 1) == 0;
#pragma unroll
              for (int c = 0; c < 4; ++c) {
                if (c < C) {
                  const double d = fwd ? pks - pq[c] : pq[c] - pks;
                  const double v = KERNEL == 1 ? fabs(d) : d * d;
                  db[c] += aj * min_raw(v, p.lambda);  // (v, lambda >= 0: the same value as v < lambda ? v : lambda)
                }
              }
            }
          }
#pragma unroll
          for (int c = 0; c < 4; ++c) di[c] = db[c];
#pragma unroll
          for (int jj = 0; jj < 8; ++jj) {
            if (jj < nout) {
#pragma unroll
              for (int c = 0; c < 4; ++c) {
                const int k = c * kWave + lane;
                if (c < C && k < K) di[c] += st[kWS + jj * kWS + k];
              }
            }
          }
          // first minimum over the labels (AddColumn's vectorMin): the minimum itself by a plain reduction,
          // then the lowest label k = 64 c + lane that attains it -- lowest chunk with a hit, lowest lane in it
#pragma unroll
          for (int c = 0; c < 4; ++c) di[c] = (c < C && c * kWave + lane < K) ? di[c] : inf;
          const double vbest = wave_min_dpp(min_raw(min_raw(di[0], di[1]), min_raw(di[2], di[3])));
          int bi = 0;
          double eb = 0;
          bool found = false;
#pragma unroll
          for (int c = 0; c < 4; ++c) {
            const unsigned long long hit = __builtin_amdgcn_ballot_w64(di[c] == vbest);
            if (!found && hit) {
              const int l = __builtin_ctzll(hit);
              bi = c * kWave + l;
              eb = readlane_f64(db[c], l);
              found = true;
            }
          }
          xprev2 = xprev; xprev = bi;
          if (lane == 0) { sc[9] = eb; ((int *)(sc + 10))[0] = bi; }
        }
            WIDE_VISITS_END
    }
#undef WIDE_VISITS_BEGIN
#undef WIDE_VISITS_END
#undef WIDE_VISITS_END_
    if (SPEC && seg >= 0) {
      const int verdict = wide_spec_commit<BACKWARD, PRIMAL, UPDATE>(p.self, epoch, p0, p1, seg, spec_in ? 1 : 0, (int)(L.ctl - (int *)lds));
      if (verdict == 2) { if (tid == 0) st_sc1(p.abort_flag, 1); return; }
      if (verdict == 1) continue;   // (ctl[3] is set: the same run once more)
    }
    if (p.timeline && tid == 0) p.timeline[((size_t)D * p.tl_stride + run) * 2 + 1] = wall_clock64();
  }
#undef WSTAMP
#undef WPOS
#undef WVALID
  if (WIDE_PROF(p) && lane == 0) {
    if (wave == 0) {
      for (int i = 0; i < 16; ++i) atomicAdd(p.prof + i, pacc[i]);
      atomicAdd(p.prof + 21, pwait);
      atomicAdd(p.prof + 22, pvis);
    }
    if (wave >= kWideCompute) atomicAdd(p.prof + 16 + (wave - kWideCompute), pbusy);
    atomicAdd(p.prof + 32 + wave, pbusy);
    atomicAdd(p.prof + 48 + wave, plate);   // visits in which this wave needed more than 8000 cycles to reach the barrier
  }
}
template <int KERNEL, bool BACKWARD, bool PRIMAL, bool UPDATE>
__global__ __launch_bounds__(kWideThreads) void trws_wide_kernel(DevParams p, int epoch) {
  wide_body<KERNEL, BACKWARD, PRIMAL, UPDATE>(p, epoch);
}
template <bool BACKWARD, bool PRIMAL, bool UPDATE>
__global__ __launch_bounds__(kWideThreads) void trws_wide_spec_kernel(DevParams p, int epoch) {
  wide_body<1, BACKWARD, PRIMAL, UPDATE, true>(p, epoch);
}
template <int KERNEL, bool BACKWARD, bool PRIMAL, bool UPDATE>
__global__ __launch_bounds__(kWideThreads) void trws_wide_group_kernel(GroupArgs ga, int epoch) {
  wide_body<KERNEL, BACKWARD, PRIMAL, UPDATE>(ga.pp[group_strip(ga)], epoch);
}
#undef WSYNC

}  // namespace

size_t wide_lds_bytes() { return sizeof(double) * kWideLdsDoubles; }
// (the runner of the speculative schedule overlays the visit loops' LDS and needs a little more)
static_assert(kWrDoubles * 8 <= 160 * 1024, "wide runner LDS");
size_t wide_spec_lds_bytes() { return sizeof(double) * (size_t)(kWideLdsDoubles > kWrDoubles ? kWideLdsDoubles : kWrDoubles); }

void wide_set_attributes() {
  const int wlds = (int)wide_lds_bytes();
  {
    const int slds = (int)wide_spec_lds_bytes();
#define SET_S(BW, PR, UP) STEREO_HIP_CHECK(hipFuncSetAttribute((const void *)trws_wide_spec_kernel<BW, PR, UP>, hipFuncAttributeMaxDynamicSharedMemorySize, slds))
    SET_S(false, false, true); SET_S(true, false, true); SET_S(false, true, true); SET_S(false, true, false);
#undef SET_S
  }
#define SET_W(NAME)                                                                                                             \
  STEREO_HIP_CHECK(hipFuncSetAttribute((const void *)NAME<1, false, false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, wlds)); \
  STEREO_HIP_CHECK(hipFuncSetAttribute((const void *)NAME<1, true, false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, wlds));  \
  STEREO_HIP_CHECK(hipFuncSetAttribute((const void *)NAME<1, false, true, true>, hipFuncAttributeMaxDynamicSharedMemorySize, wlds));  \
  STEREO_HIP_CHECK(hipFuncSetAttribute((const void *)NAME<1, false, true, false>, hipFuncAttributeMaxDynamicSharedMemorySize, wlds))
  SET_W(trws_wide_kernel); SET_W(trws_wide_group_kernel);
#undef SET_W
#define SET_W(NAME)                                                                                                             \
  STEREO_HIP_CHECK(hipFuncSetAttribute((const void *)NAME<2, false, false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, wlds)); \
  STEREO_HIP_CHECK(hipFuncSetAttribute((const void *)NAME<2, true, false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, wlds));  \
  STEREO_HIP_CHECK(hipFuncSetAttribute((const void *)NAME<2, false, true, true>, hipFuncAttributeMaxDynamicSharedMemorySize, wlds));  \
  STEREO_HIP_CHECK(hipFuncSetAttribute((const void *)NAME<2, false, true, false>, hipFuncAttributeMaxDynamicSharedMemorySize, wlds))
  SET_W(trws_wide_kernel); SET_W(trws_wide_group_kernel);
#undef SET_W
}

#define WIDE_SWITCH(NAME, ARG)                                                                                     \
  const size_t wlds = wide_lds_bytes();                                                                            \
  const dim3 wgrid(blocks), wblock(kWideThreads);                                                                  \
  if (kernel == 2) {                                                                                               \
    switch (what) {                                                                                                \
      case 0: hipLaunchKernelGGL((NAME<2, false, false, true>), wgrid, wblock, wlds, s, ARG, epoch); break;        \
      case 1: hipLaunchKernelGGL((NAME<2, true, false, true>), wgrid, wblock, wlds, s, ARG, epoch); break;         \
      case 2: hipLaunchKernelGGL((NAME<2, false, true, true>), wgrid, wblock, wlds, s, ARG, epoch); break;         \
      default: hipLaunchKernelGGL((NAME<2, false, true, false>), wgrid, wblock, wlds, s, ARG, epoch); break;       \
    }                                                                                                              \
  } else {                                                                                                         \
    switch (what) {                                                                                                \
      case 0: hipLaunchKernelGGL((NAME<1, false, false, true>), wgrid, wblock, wlds, s, ARG, epoch); break;        \
      case 1: hipLaunchKernelGGL((NAME<1, true, false, true>), wgrid, wblock, wlds, s, ARG, epoch); break;         \
      case 2: hipLaunchKernelGGL((NAME<1, false, true, true>), wgrid, wblock, wlds, s, ARG, epoch); break;         \
      default: hipLaunchKernelGGL((NAME<1, false, true, false>), wgrid, wblock, wlds, s, ARG, epoch); break;       \
    }                                                                                                              \
  }                                                                                                                \
  STEREO_HIP_CHECK(hipGetLastError());

void launch_wide(int kernel, int what, int blocks, hipStream_t s, const DevParams &p, int epoch) {
  if (p.spec_kind[0] != nullptr && p.spec_kind[1] != nullptr && kernel == 1) {
    // the speculative schedule's kernel
    const size_t slds = wide_spec_lds_bytes();
    const dim3 grid(blocks), block(kWideThreads);
    switch (what) {
      case 0: hipLaunchKernelGGL((trws_wide_spec_kernel<false, false, true>), grid, block, slds, s, p, epoch); break;
      case 1: hipLaunchKernelGGL((trws_wide_spec_kernel<true, false, true>), grid, block, slds, s, p, epoch); break;
      case 2: hipLaunchKernelGGL((trws_wide_spec_kernel<false, true, true>), grid, block, slds, s, p, epoch); break;
      default: hipLaunchKernelGGL((trws_wide_spec_kernel<false, true, false>), grid, block, slds, s, p, epoch); break;
    }
    STEREO_HIP_CHECK(hipGetLastError());
    return;
  }
  WIDE_SWITCH(trws_wide_kernel, p)
}
void launch_wide_group(int kernel, int what, int blocks, hipStream_t s, const GroupArgs &ga, int epoch) { WIDE_SWITCH(trws_wide_group_kernel, ga) }
#undef WIDE_SWITCH

}  // namespace stereo
