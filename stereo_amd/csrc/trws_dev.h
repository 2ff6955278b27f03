// Device-side common part of the TRW-S kernels (see trws_plan.hip for the overview): launch
// parameters, hand-over accesses, wave reductions, the message update routines (certified min-plus
// fast path, second look, serial lower-envelope construction) and the descriptor decoding shared
// by the sweep kernel families (trws_generic.hip, trws_pipe.hip, trws_pipe2.hip, trws_wide.hip).
// Everything but the launch-parameter structs has internal linkage (anonymous namespace).
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>

#include "trws_graph.h"

namespace stereo {

struct DevParams {
  int K, Kp, kernel;
  double lambda;
  const double *unary;
  double *msg;
  const double *q, *qprim;          // per edge, or null when `pos` is used
  const double *pos;                // shared positions
  const uint16_t *perm_q, *perm_qp;  // per edge sort permutations (null with pos)
  const uint16_t *perm_pos;
  const double *alpha;
  const uint8_t *mdir;
  const int32_t *tail;
  const int32_t *order;
  const int32_t *fptr, *fidx, *bptr, *bidx;
  const double *gamma;
  const int32_t *lb_pos_node, *lb_pos_edge;
  double *lbterms, *eterms;
  int32_t *x;
  // persistent dataflow sweeps
  const int32_t *run_ptr[2];
  const int32_t *run_order[2];  // ticket -> run (nullptr: identity)
  int nruns[2];
  const int32_t *dep_ptr[2], *dep_rank[2];
  const int8_t *in_slot[2];
  int32_t *done;    // per rank: epoch of the last completed visit
  int32_t *ticket;  // run dispenser of the current launch
  int32_t *abort_flag;  // (d_ctl + 1; the four words behind it take the give-up report, see report_give_up)
  long long spin_ticks; // how long a visit may wait for another workgroup's flag: 100 MHz wall-clock ticks
  int n_own;            // strips: local node ids >= n_own are the halo (a neighbouring strip's nodes); else N
  int N;
  unsigned long long *fallbacks;  // messages that needed the serial envelope (diagnostics)
  int certificate;                // 0: always run the serial envelope
  int lean;                       // STEREO_TRWS_MESSAGES_MINPLUS in the wide-label regime (trws_wide_kernel's plain min-plus branch)
  unsigned long long *prof;       // optional: 8 phase-cycle accumulators (development)
  const int32_t *desc[2];         // packed node descriptors of the pipelined kernels
  int prof_run;
  int debug;  // development switches: 2 / 4 profile backward / forward sweeps only, 256 no windowed paths,
              // 512 serial envelopes by the lane-read loop instead of the mask construction, 2048 by the
              // bit-set walk instead of the closed form (build_envelope_parallel)
              // (none of them changes a result)
  unsigned long long *timeline;  // optional [2][nruns][2] wall-clock stamps (development)
  int window;  // wide kernel: sources within lambda of a destination lie within +-window indices
  double uniform_step;  // wide kernel: != 0 if pos[k+d] - pos[k] == d * step exactly for |d| <= window <= 16
  int win_ok;  // shared strictly ascending positions and window <= 16: windowed min-plus allowed
  double pos_gap;  // smallest distance of two neighbouring shared positions (ascending case)
  double pos_first, pos_last;  // ... their two ends
  // Row strips (one plan per strip, normally one per GPU): a plan dispenses only its own runs and
  // writes what the neighbouring strips read -- messages on edges that cross the boundary, the
  // completion flag and the label of a boundary node -- straight into THEIR arrays (same index
  // space on every strip; over xGMI when the neighbour is another GPU).  [0] previous, [1] next strip.
  int ntickets[2];
  // Speculative schedule of the long serial run (trws_graph.h: Sweep::Spec; trws_pipe_kernel, shared positions,
  // linear kernel).  spec_kind == nullptr: the plain chain schedule.  Flags of the scheme live behind the nodes'
  // own in `done`: done[N + s] = the runner's rows for segment s are there, done[N + nseg + s] = segment s - 1 has
  // committed (its last node's rows and label are final).
  const int32_t *spec_kind[2];        // per run: 0, or 1 + segment index
  int spec_c0[2], spec_c1[2];         // schedule positions of the cut run
  int spec_len, spec_nseg, spec_max_len;
  double *spec_rows;                  // [nseg][8][K]: row j = what the first node of segment s finds as its j-th message
  int32_t *spec_x;                    // [nseg]: label of the node in front of segment s (primal pass)
  double *spec_undo;                  // [nseg][max_len][4][K]: the rows a segment's visits overwrite, for a second walk
  unsigned long long *spec_stat;      // [0] segments walked twice, [1] segments committed, [2] runner visits
  int tl_stride;                      // runs per direction in `timeline`
  const DevParams *self;              // this block in global memory (what chain_runner reads its parameters from)
  double *peer_msg0, *peer_msg1;      // (scalars, not arrays: an index computed at run time would put
  int32_t *peer_done0, *peer_done1;   //  the whole parameter block into scratch memory)
  int32_t *peer_x0, *peer_x1;
};

// Several strips of one problem in ONE launch (row strips that share a device: logical strips, or
// a process that owns more than one band): workgroup b works for strip s with first[s] <= b <
// first[s + 1], on that strip's parameters.  One launch, so that all of them are resident
// together whatever the runtime does with streams (strips wait for each other in both directions).
constexpr int kMaxGroup = 16;
struct GroupArgs {
  const DevParams *pp;
  int n;
  int first[kMaxGroup + 1];
};

namespace {

constexpr int kWave = 64;
constexpr int kBlock = 256;
constexpr int kWavesPerBlock = kBlock / kWave;
constexpr int kWaveVecs = 5;  // K-vectors of LDS scratch per wave

// ---- hand-over accesses (sc0 sc1): data handed between workgroups inside one launch never sits
// in a per-CU L1 or a non-coherent L2 (cdna_hip_programming.md G16, R1/R2).  System scope, not
// agent scope: with row strips the other workgroup may run on the neighbouring GPU and write into
// this GPU's memory over xGMI; on one GPU both scopes cost the same (measured: 68.6 vs 68.6
// iterations/s at 450x375x60, 132.2 vs 132.2 ms at 1500x1000x256).
#ifndef STEREO_HANDOVER_SCOPE
#define STEREO_HANDOVER_SCOPE __HIP_MEMORY_SCOPE_SYSTEM
#endif
__device__ __forceinline__ double ld_sc1(const double *p) {
  return __hip_atomic_load(p, __ATOMIC_RELAXED, STEREO_HANDOVER_SCOPE);
}
__device__ __forceinline__ void st_sc1(double *p, double v) {
  __hip_atomic_store(p, v, __ATOMIC_RELAXED, STEREO_HANDOVER_SCOPE);
}
__device__ __forceinline__ int ld_sc1(const int32_t *p) {
  return __hip_atomic_load(p, __ATOMIC_RELAXED, STEREO_HANDOVER_SCOPE);
}
__device__ __forceinline__ void st_sc1(int32_t *p, int v) {
  __hip_atomic_store(p, v, __ATOMIC_RELAXED, STEREO_HANDOVER_SCOPE);
}

__device__ __forceinline__ double readlane_f64(double v, int lane) {
  int lo = __double2loint(v), hi = __double2hiint(v);
  lo = __builtin_amdgcn_readlane(lo, lane);
  hi = __builtin_amdgcn_readlane(hi, lane);
  return __hiloint2double(hi, lo);
}
// min / max of two doubles as ONE instruction.  std::fmin / fmax cost ~1.6x as much here: in IEEE
// mode the compiler puts a canonicalising v_max_f64 x, x, x in front of every v_min / v_max
// (tools/micro_valu.hip: 13 vs 8 cycles per wave instruction).  Same result for every non-NaN input
// (the sign of a zero result may differ, which no comparison or sum downstream can see).
__device__ __forceinline__ double min_raw(double a, double b) {
  double r;
  asm("v_min_f64 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
  return r;
}
__device__ __forceinline__ double max_raw(double a, double b) {
  double r;
  asm("v_max_f64 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
  return r;
}
// acc = min(acc, x) where `take` holds, else acc.  NOT a conditional expression around min_raw: the
// compiler does not execute inline assembly speculatively, so that form becomes an exec-mask region
// with a branch around one instruction -- in the pair loop of every message (found in the ISA in
// round 3: five of the ~24 instructions per source).
__device__ __forceinline__ double min_raw_if(bool take, double acc, double x) {
  return min_raw(acc, take ? x : __builtin_huge_val());
}
// wave-uniform predicate -> scalar branch
#define UNI(c) (__builtin_amdgcn_ballot_w64(c) != 0)

// ---- DPP wave reductions (gfx9 row_bcast forms): ~20 VALU instead of 12 ds_bpermute.
// The combined value ends up in lane 63 and is broadcast with v_readlane.
// (steps that write every lane from a valid source lane -- ROW_MASK 0xF: the quad / row permutations --
//  name no `old` value: the compiler then emits the bare v_mov_b32_dpp instead of copying the register
//  first, 3 instead of 5 instructions per step of a double; the row_bcast steps keep `old` = the
//  lane's own value for the rows they leave alone)
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ double dpp_f64(double v) {
  int lo = __double2loint(v), hi = __double2hiint(v);
#ifndef STEREO_DPP_OLD
  if (ROW_MASK == 0xF) {
    lo = __builtin_amdgcn_update_dpp(0, lo, CTRL, 0xF, 0xF, true);
    hi = __builtin_amdgcn_update_dpp(0, hi, CTRL, 0xF, 0xF, true);
  } else
#endif
  {
    lo = __builtin_amdgcn_update_dpp(lo, lo, CTRL, ROW_MASK, 0xF, false);
    hi = __builtin_amdgcn_update_dpp(hi, hi, CTRL, ROW_MASK, 0xF, false);
  }
  return __hiloint2double(hi, lo);
}
// (old = the lane's own value wherever the pattern has no source lane: shifts, scans)
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ double dpp_f64_keep(double v) {
  int lo = __double2loint(v), hi = __double2hiint(v);
  lo = __builtin_amdgcn_update_dpp(lo, lo, CTRL, ROW_MASK, 0xF, false);
  hi = __builtin_amdgcn_update_dpp(hi, hi, CTRL, ROW_MASK, 0xF, false);
  return __hiloint2double(hi, lo);
}
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ int dpp_i32(int v) {
#ifndef STEREO_DPP_OLD
  if (ROW_MASK == 0xF) return __builtin_amdgcn_update_dpp(0, v, CTRL, 0xF, 0xF, true);
#endif
  return __builtin_amdgcn_update_dpp(v, v, CTRL, ROW_MASK, 0xF, false);
}
#define DPP_REDUCE_STEPS(STEP) \
  STEP(0xB1, 0xF) STEP(0x4E, 0xF) STEP(0x141, 0xF) STEP(0x140, 0xF) STEP(0x142, 0xA) STEP(0x143, 0xC)
__device__ __forceinline__ double wave_min_dpp(double v) {
#define STEP(C, M) { const double o = dpp_f64<C, M>(v); v = min_raw(o, v); }
  DPP_REDUCE_STEPS(STEP)
#undef STEP
  return readlane_f64(v, 63);
}
__device__ __forceinline__ double wave_max_dpp(double v) {
#define STEP(C, M) { const double o = dpp_f64<C, M>(v); v = max_raw(o, v); }
  DPP_REDUCE_STEPS(STEP)
#undef STEP
  return readlane_f64(v, 63);
}
// minimum of `a` and maximum of `b` over the wave in one interleaved pass (two independent chains)
__device__ __forceinline__ void wave_min_max_dpp(double &a, double &b) {
#define STEP(C, M) { const double oa = dpp_f64<C, M>(a), ob = dpp_f64<C, M>(b); a = min_raw(oa, a); b = max_raw(ob, b); }
  DPP_REDUCE_STEPS(STEP)
#undef STEP
  a = readlane_f64(a, 63); b = readlane_f64(b, 63);
}
// lexicographic (value, index) minimum -> index of the FIRST minimum, uniform
__device__ __forceinline__ int wave_argmin_dpp(double v, int i) {
#define STEP(C, M) { const double ov = dpp_f64<C, M>(v); const int oi = dpp_i32<C, M>(i); \
                     if (ov < v || (ov == v && oi < i)) { v = ov; i = oi; } }
  DPP_REDUCE_STEPS(STEP)
#undef STEP
  return __builtin_amdgcn_readlane(i, 63);
}


__device__ __forceinline__ double wave_min(double v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
    double o = __shfl_xor(v, off, kWave);
    v = o < v ? o : v;
  }
  return v;
}

// (value, index) lexicographic minimum: the FIRST minimum wins
// (typeStereoLinear.h:242-249 strict '>').
__device__ __forceinline__ void wave_argmin(double &v, int &i) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
    double ov = __shfl_xor(v, off, kWave);
    int oi = __shfl_xor(i, off, kWave);
    if (ov < v || (ov == v && oi < i)) { v = ov; i = oi; }
  }
}

template <int KERNEL>
__device__ __forceinline__ double pair_cost(double alpha, double d, double h) {
  // typeStereoLinear.h:474 m_alpha*std::abs(d) + hj ; typeStereoQuadratic.h:484 m_alpha*val*val + hj
  if (KERNEL == 1) return alpha * fabs(d) + h;
  return alpha * d * d + h;
}

// Serial construction of the lower envelope, executed by ONE lane, exactly as
// typeStereoLinear.h:401-460 / typeStereoQuadratic.h:407-470 do it, including
// their tie and "numerical stability" behaviour (stale breakpoints survive pops).
// Hs/Qs: heights / positions in ascending position order.  Stack entries are
// stored by value (sh, sq) with breakpoints z.
template <int KERNEL>
__device__ void build_envelope(int K, double alpha, const double *Hs, const double *Qs,
                               double *sh, double *sq, double *z) {
  const double inf = __builtin_huge_val();
  int top = 0;
  double hj = Hs[0], qj = Qs[0], zt = -inf;
  sh[0] = hj; sq[0] = qj; z[0] = -inf; z[1] = inf;
  for (int k = 1; k < K; ++k) {
    const double hk = Hs[k], qk = Qs[k];
    for (int guard = k; guard >= 0; --guard) {
      if (KERNEL == 1) {
        const double dist = alpha * fabs(qk - qj);
        if (dist + hk < hj) {
          if (top == 0) {
            sh[0] = hk; sq[0] = qk; z[0] = -inf; z[1] = inf; hj = hk; qj = qk;
            break;  // the reference re-compares the new bottom with itself and breaks
          }
          --top; hj = sh[top]; qj = sq[top];
        } else if (dist + hj <= hk) {
          break;
        } else {
          const double s = ((hk - hj) + alpha * (qk + qj)) / (2 * alpha);
          if (s >= qk) break;
          if (s <= qj) break;
          ++top; sh[top] = hk; sq[top] = qk; z[top] = s; z[top + 1] = inf; hj = hk; qj = qk;
          break;
        }
      } else {
        if (qk - qj < 1e-8) {
          if (hj > hk) {
            if (top == 0) {
              sh[0] = hk; sq[0] = qk; z[0] = -inf; z[1] = inf; hj = hk; qj = qk; zt = -inf;
              break;
            }
            --top; hj = sh[top]; qj = sq[top]; zt = z[top];
          } else {
            break;
          }
        } else {
          const double s = ((hk + alpha * qk * qk) - (hj + alpha * qj * qj)) / (2 * alpha * (qk - qj));
          if (s <= zt) {
            --top;
            if (top < 0) { top = 0; break; }  // unreachable for finite input (z[0] = -inf)
            hj = sh[top]; qj = sq[top]; zt = z[top];
          } else {
            ++top; sh[top] = hk; sq[top] = qk; z[top] = s; z[top + 1] = inf; hj = hk; qj = qk; zt = s;
            break;
          }
        }
      }
    }
  }
}

// The same construction with the whole state in registers, for K <= 64: lane k
// holds the k-th sorted source (hs, qs); afterwards lane j holds stack slot j
// (sh, sq) and zz = z[j+1].  All arithmetic is wave-uniform (operands come from
// v_readlane), so every lane computes exactly what the reference's scalar code
// computes; branches are scalar.  Returns the highest slot ever written.
template <int KERNEL>
__device__ __forceinline__ int build_envelope_regs(int K, double alpha, double hs, double qs,
                                                   double &sh, double &sq, double &zz, int lane) {
  const double inf = __builtin_huge_val();
  int top = 0, maxtop = 0;
  double hj = readlane_f64(hs, 0), qj = readlane_f64(qs, 0), zt = -inf;
  sh = hj; sq = qj; zz = inf;
  for (int k = 1; k < K; ++k) {
    const double hk = readlane_f64(hs, k), qk = readlane_f64(qs, k);
    for (;;) {
      if (KERNEL == 1) {
        const double dist = alpha * fabs(qk - qj);
        if (UNI(dist + hk < hj)) {
          if (top == 0) {
            if (lane == 0) { sh = hk; sq = qk; zz = inf; }
            hj = hk; qj = qk;
            break;
          }
          --top; hj = readlane_f64(sh, top); qj = readlane_f64(sq, top);
        } else if (UNI(dist + hj <= hk)) {
          break;
        } else {
          const double s = ((hk - hj) + alpha * (qk + qj)) / (2 * alpha);
          if (UNI(s >= qk)) break;
          if (UNI(s <= qj)) break;
          if (lane == top) zz = s;  // z[top+1] = s
          ++top;
          if (lane == top) { sh = hk; sq = qk; zz = inf; }
          hj = hk; qj = qk;
          break;
        }
      } else {
        if (UNI(qk - qj < 1e-8)) {
          if (UNI(hj > hk)) {
            if (top == 0) {
              if (lane == 0) { sh = hk; sq = qk; zz = inf; }
              hj = hk; qj = qk; zt = -inf;
              break;
            }
            --top; hj = readlane_f64(sh, top); qj = readlane_f64(sq, top);
            zt = top == 0 ? -inf : readlane_f64(zz, top - 1);
          } else {
            break;
          }
        } else {
          const double s = ((hk + alpha * qk * qk) - (hj + alpha * qj * qj)) / (2 * alpha * (qk - qj));
          if (UNI(s <= zt)) {
            if (top == 0) break;  // unreachable for finite input (z[0] = -inf)
            --top; hj = readlane_f64(sh, top); qj = readlane_f64(sq, top);
            zt = top == 0 ? -inf : readlane_f64(zz, top - 1);
          } else {
            if (lane == top) zz = s;
            ++top;
            if (lane == top) { sh = hk; sq = qk; zz = inf; }
            hj = hk; qj = qk; zt = s;
            break;
          }
        }
      }
    }
    maxtop = top > maxtop ? top : maxtop;
  }
  return maxtop;
}

// ---- the linear-kernel construction without its inner loop -------------------------------------
// Every comparison of typeStereoLinear.h:401-460 involves the new cone k and the cone j on top of
// the stack, nothing else -- so all of them can be evaluated up front for source k against ALL
// sources j at once (lane j), giving three 64-bit masks per k, and the stack itself shrinks to a
// bit set over source indices (sources arrive in ascending position order and the stack is a
// subsequence of them: top = highest set bit, pop = clear it).  One trip per source, no dependent
// chain of lane reads and scalar branches per pop:
//   m1[j]: dist + hk <  hj   (j is dominated: pop)           typeStereoLinear.h:417-431
//   m2[j]: dist + hj <= hk   (k is dominated: drop k)         :432-435
//   m3[j]: s >= qk or s <= qj ("numerical stability": drop k) :444-449
// with s = ((hk - hj) + alpha (qk + qj)) / (2 alpha).  The two tests on s are made on the numerator:
// x -> fl(x / c) is monotone, so s >= qk <=> num >= thi(qk) with thi = the smallest double whose
// quotient reaches qk, and s <= qj <=> num <= tlo(qj) with tlo the largest one whose quotient stays
// at or below qj; both thresholds are found per lane by stepping ulp-wise from fl(q c) (a handful of
// divisions per message instead of one per pair).  Slot contents are recorded per lane exactly as
// the serial code leaves them (stale breakpoints above `top` included): lane t = slot t holds the
// source stored there and the pair whose crossing is z[t+1]; values are filled in at the end with
// one vector division.  Returns false (nothing done) on inputs outside the argument above.
__device__ __forceinline__ double ulp_up(double x) {    // next double above a finite x
  long long b = __double_as_longlong(x);
  b = x > 0 ? b + 1 : x < 0 ? b - 1 : 1;                // +-0 -> smallest positive denormal
  return __longlong_as_double(b);
}
__device__ __forceinline__ double ulp_down(double x) {  // next double below a finite x
  long long b = __double_as_longlong(x);
  b = x > 0 ? b - 1 : x < 0 ? b + 1 : (long long)0x8000000000000001ull;
  return __longlong_as_double(b);
}

// Numerator thresholds of the construction below: thi = smallest x with fl(x / c) >= qs, tlo = largest
// x with fl(x / c) <= qs (c = 2 alpha).  false: inputs outside the argument (the caller falls back).
__device__ __forceinline__ bool envelope_thresholds(int K, double alpha, double hs, double qs, double &thi_out,
                                                    double &tlo_out, int lane, bool *exact_c_out = nullptr) {
  const double inf = __builtin_huge_val();
  const bool act = lane < K;
  const double c = 2 * alpha;
  bool ok = alpha > 0 && c < inf && (!act || (fabs(hs) < inf && fabs(qs) < inf));
  // at most kSteps ulp steps from fl(q c), else give up
  constexpr int kSteps = 6;
  double thi = qs * c, tlo = thi;
  ok = ok && (!act || fabs(thi) < 1e300);
  // c a power of two (alpha = 1, 2, 1/2 ...: unit smoothness weights) and nothing near the ends of the
  // exponent range: x -> x / c is exact, so both thresholds are q c itself and no stepping is needed
  const bool exact_c = (__double_as_longlong(c) & 0x000FFFFFFFFFFFFFll) == 0 && c > 1e-100 && c < 1e100 &&
                       !UNI(act && qs != 0 && !(fabs(qs) > 1e-100 && fabs(qs) < 1e100));
  if (!UNI(!ok) && !exact_c) {
    // thi: smallest x with fl(x / c) >= qs
    bool settled = !act;
    {
      const bool above = thi / c >= qs;   // start inside the set: walk down to its edge, else walk up into it
      for (int i = 0; i < kSteps; ++i) {
        const double nx = above ? ulp_down(thi) : ulp_up(thi);
        const bool in = nx / c >= qs;
        if (above) { if (in && !settled) thi = nx; else settled = true; }
        else { if (!settled) thi = nx; if (in) settled = true; }
      }
      if (above) {  // settled only if the last step left the set
        settled = settled || !(ulp_down(thi) / c >= qs);
      }
    }
    ok = ok && settled;
    // tlo: largest x with fl(x / c) <= qs
    settled = !act;
    {
      const bool below = tlo / c <= qs;
      for (int i = 0; i < kSteps; ++i) {
        const double nx = below ? ulp_up(tlo) : ulp_down(tlo);
        const bool in = nx / c <= qs;
        if (below) { if (in && !settled) tlo = nx; else settled = true; }
        else { if (!settled) tlo = nx; if (in) settled = true; }
      }
      if (below) settled = settled || !(ulp_up(tlo) / c <= qs);
    }
    ok = ok && settled;
  }
  thi_out = thi; tlo_out = tlo;
  if (exact_c_out) *exact_c_out = exact_c;
  return !UNI(!ok);
}

// The three masks of source k (see above); lane j holds source j's (hs, qs, tlo).
__device__ __forceinline__ void envelope_masks_of(double alpha, double hk, double qk, double thik, double hs, double qs,
                                                  double tlo, unsigned long long &m1, unsigned long long &m2,
                                                  unsigned long long &m3) {
  const double dist = alpha * fabs(qk - qs);
  m1 = __builtin_amdgcn_ballot_w64(dist + hk < hs);
  m2 = __builtin_amdgcn_ballot_w64(dist + hs <= hk);
  const double num = (hk - hs) + alpha * (qk + qs);
  m3 = __builtin_amdgcn_ballot_w64(num >= thik || num <= tlo);
}

// State of the bit-set stack while the sources are taken in (lane t: slot t).
struct EnvelopeStack {
  unsigned long long A = 1;      // source 0 is the bottom of the stack
  int top = 0, maxtop = 0;
  int src = 0, zk = -1, zj = 0;  // lane t: slot t holds source `src`; z[t+1] = crossing of (zk, zj), inf if zk < 0
};
__device__ __forceinline__ void envelope_take(EnvelopeStack &S, int k, unsigned long long m1, unsigned long long m2,
                                              unsigned long long m3, int lane) {
  const unsigned long long B = S.A & ~m1;
  if (B == 0) {  // every cone on the stack is dominated: k becomes the bottom (typeStereoLinear.h:419-425)
    S.A = 1ull << k;
    if (lane == 0) { S.src = k; S.zk = -1; }
    S.top = 0;
    return;
  }
  const int js = 63 - __builtin_clzll(B);   // the cone k meets: the highest one it does not dominate
  S.A &= (2ull << js) - 1;                  // (js < k <= 63)
  S.top = __builtin_popcountll(S.A) - 1;
  if (((m2 | m3) >> js) & 1) return;
  if (lane == S.top) { S.zk = k; S.zj = js; }
  ++S.top;
  if (lane == S.top) { S.src = k; S.zk = -1; }
  S.A |= 1ull << k;
  S.maxtop = S.top > S.maxtop ? S.top : S.maxtop;
}
// slot contents -> (sh, sq, zz) as the serial code leaves them
__device__ __forceinline__ void envelope_fill(const EnvelopeStack &S, double alpha, double hs, double qs, double &sh,
                                              double &sq, double &zz) {
  const double inf = __builtin_huge_val();
  sh = __shfl(hs, S.src, kWave); sq = __shfl(qs, S.src, kWave);
  const double hk = __shfl(hs, S.zk < 0 ? 0 : S.zk, kWave), qk = __shfl(qs, S.zk < 0 ? 0 : S.zk, kWave);
  const double hj = __shfl(hs, S.zj, kWave), qj = __shfl(qs, S.zj, kWave);
  const double s = ((hk - hj) + alpha * (qk + qj)) / (2 * alpha);
  zz = S.zk < 0 ? inf : s;
}

__device__ __forceinline__ bool build_envelope_masks(int K, double alpha, double hs, double qs, double &sh,
                                                     double &sq, double &zz, int lane, int &maxtop_out) {
  double thi, tlo;
  if (!envelope_thresholds(K, alpha, hs, qs, thi, tlo, lane)) return false;
  EnvelopeStack S;
  for (int k = 1; k < K; ++k) {
    const double hk = readlane_f64(hs, k), qk = readlane_f64(qs, k), thik = readlane_f64(thi, k);
    unsigned long long m1, m2, m3;
    envelope_masks_of(alpha, hk, qk, thik, hs, qs, tlo, m1, m2, m3);
    envelope_take(S, k, m1, m2, m3, lane);
  }
  envelope_fill(S, alpha, hs, qs, sh, sq, zz);
  maxtop_out = S.maxtop;
  return true;
}

// ---- the linear-kernel construction without ANY serial loop (round 3) --------------------------
// The bit-set form above still walks the sources one after the other (~60 dependent steps of ~66
// instructions: 33 k cycles per message, and on volumes with flat columns one such message per row
// sits on the sweep's critical path).  Its state, however, has a closed form.  Let R1_k = {j < k :
// cone k dominates cone j} (the pop test, typeStereoLinear.h:417), one 64-bit row per source from one
// ballot.  A cone leaves the stack when the first later cone that dominates it arrives, PROVIDED the
// serial code gets to test it, i.e. provided everything above it on the stack goes at the same time.
// Assume that for now: with D_k = OR of the rows before k (a prefix OR over lanes) the stack in front
// of source k is S_k = P & ~D_k & below(k), P = the set of sources that were pushed, and k meets
// js(k) = the highest bit of S_k & ~R1_k.  Whether k is pushed is ONE pair test against js(k) (the
// drop tests :432-449), evaluated by lane k on demand, so P is the fixed point of
//     P(k) = [S_k & ~R1_k empty]  or  not dropped(k, js(k)),
// unique because P(k) only depends on P(j), j < k; iterating from "all pushed" settles at least one
// more source per round and in practice everything within a few rounds (a pushed source mostly stays
// pushed).  The assumption is then CHECKED: the cones k dominates must be a top segment of S_k (no
// bit of S_k & R1_k below js(k)); transitivity of dominance makes that hold in exact arithmetic, a
// rounding accident makes this routine return false and the bit-set walk above runs instead.  With
// the check the closed form IS the serial code's trace, by induction over k: same stack in front of
// k, same pops, same cone met, same drop tests, same push.  Slot contents as the serial code leaves
// them (stale slots above `top` included): source k lands in slot popcount(S_k & ~R1_k); lane t
// takes the LAST source written to slot t and the last breakpoint written to z[t+1] (a push at slot
// t resets it to +inf, a push at slot t+1 sets it) -- two LDS atomic maxima over (time, payload).
// tab: 64 x 4 doubles (this wave's source table), scr: 129 ints, both this wave's own LDS.
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ unsigned long long dpp_or_u64(unsigned long long v) {
  const int lo = (int)(unsigned)v, hi = (int)(unsigned)(v >> 32);
  const unsigned l2 = (unsigned)__builtin_amdgcn_update_dpp(0, lo, CTRL, ROW_MASK, 0xF, false);
  const unsigned h2 = (unsigned)__builtin_amdgcn_update_dpp(0, hi, CTRL, ROW_MASK, 0xF, false);
  return v | (((unsigned long long)h2 << 32) | l2);
}
// Rows K0 .. K0 + 3 of the pop test: lane j compares source K0 + i (from the table) with its own source
// j; the ballot is row K0 + i, kept by lane K0 + i (v_writelane, lane number as a literal: this clang has
// no writelane builtin, and a lane number in an SGPR next to the SGPR data would break gfx9's
// one-scalar-operand rule).
template <int K0>
__device__ __forceinline__ void envelope_rows4(const double *tab, double alpha, double hs, double qs, unsigned &lo,
                                               unsigned &hi) {
  double hk[4], qk[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) { hk[i] = tab[4 * (K0 + i)]; qk[i] = tab[4 * (K0 + i) + 1]; }
  unsigned long long m[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const double dist = alpha * fabs(qk[i] - qs);
    m[i] = __builtin_amdgcn_ballot_w64(dist + hk[i] < hs);
  }
  // A v_writelane must not directly follow the v_cmp that writes the SGPR pair it takes its data from:
  // the result is then occasionally stale (found by the certificate stress, 3 in 1000 constructions;
  // the compiler's hazard recogniser does not look inside inline assembly).  The scheduling barriers
  // keep all four compares in front of the four writes, which puts six instructions between the last
  // compare and the write that reads its mask (one statement: between separate ones the compiler adds s_nops).
  __builtin_amdgcn_sched_barrier(0);
  asm volatile("v_writelane_b32 %0, %2, %10\n\tv_writelane_b32 %1, %3, %10\n\t"
               "v_writelane_b32 %0, %4, %11\n\tv_writelane_b32 %1, %5, %11\n\t"
               "v_writelane_b32 %0, %6, %12\n\tv_writelane_b32 %1, %7, %12\n\t"
               "v_writelane_b32 %0, %8, %13\n\tv_writelane_b32 %1, %9, %13"
               : "+v"(lo), "+v"(hi)
               : "s"((unsigned)m[0]), "s"((unsigned)(m[0] >> 32)), "s"((unsigned)m[1]), "s"((unsigned)(m[1] >> 32)),
                 "s"((unsigned)m[2]), "s"((unsigned)(m[2] >> 32)), "s"((unsigned)m[3]), "s"((unsigned)(m[3] >> 32)),
                 "n"(K0), "n"(K0 + 1), "n"(K0 + 2), "n"(K0 + 3));
  __builtin_amdgcn_sched_barrier(0);
}

#ifdef STEREO_HIP_MESSAGE_PROFILE
// (phase times collected in registers and written at the end: an atomic per stamp costs ~1 k cycles)
#define PSTAMP(slot) do { if (prof) { const long long n_ = (long long)__builtin_readcyclecounter(); pacc[((slot) - 16) / 2] += n_ - pt0; pt0 = (long long)__builtin_readcyclecounter(); } } while (0)
#else
#define PSTAMP(slot) do { } while (0)
#endif
__device__ __forceinline__ bool build_envelope_parallel(int K, double alpha, double hs, double qs, double *tab, int *scr,
                                                        double &sh, double &sq, double &zz, int lane, int &maxtop_out,
                                                        double mag, unsigned long long *prof = nullptr) {
  const double inf = __builtin_huge_val();
  const bool act = lane < K;
#ifdef STEREO_HIP_MESSAGE_PROFILE
  long long pt0 = prof ? (long long)__builtin_readcyclecounter() : 0;  // development profile: slots 16..27
  long long pacc[4] = {0, 0, 0, 0};
#endif
  double thi, tlo;
  bool exact_c = false;
  if (!envelope_thresholds(K, alpha, hs, qs, thi, tlo, lane, &exact_c)) return false;
  PSTAMP(16);
  tab[4 * lane] = hs; tab[4 * lane + 1] = qs; tab[4 * lane + 2] = thi; tab[4 * lane + 3] = tlo;
  ((long long *)scr)[lane] = 0;
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  // rows of the pop test: lane k keeps row k.  In exact arithmetic source k dominates source j < k iff
  // v_k < v_j (v = h + alpha q), so its row is EMPTY unless v_k comes within the rounding of the largest
  // v in front of it -- decided with a margin of 1e-12 x magnitude, two thousand times the rounding of
  // either side of the comparison -- and only the other rows are computed: all of them by the unrolled
  // groups below when they are many, one by one (two per trip) when they are few.
  unsigned r1lo = 0, r1hi = 0;
  unsigned long long need;
  {
    const double v = act ? hs + alpha * qs : -inf;
    double pm = v;  // inclusive prefix maximum over the lanes, then shifted by one lane
#define STEREO_SCAN(C, M) { const double o = dpp_f64_keep<C, M>(pm); pm = max_raw(o, pm); }
    STEREO_SCAN(0x111, 0xF) STEREO_SCAN(0x112, 0xF) STEREO_SCAN(0x114, 0xF) STEREO_SCAN(0x118, 0xF)
    STEREO_SCAN(0x142, 0xA) STEREO_SCAN(0x143, 0xC)
#undef STEREO_SCAN
    pm = dpp_f64_keep<0x138, 0xF>(pm);  // wave_shr:1 (lane 0 keeps its own value and is not asked)
    need = __builtin_amdgcn_ballot_w64(act && lane > 0 && !(v - pm > 1e-12 * mag));
  }
  const int nrows = __builtin_popcountll(need);
  if (nrows > 20) {
    // (four sources per group, their table reads issued together, the lane numbers of v_writelane as
    //  literals -- hence the unrolled groups; rows of sources >= K land in lanes that take no part, row 0
    //  is masked below)
#define STEREO_ROWS4(K0) if (K > (K0)) envelope_rows4<(K0)>(tab, alpha, hs, qs, r1lo, r1hi);
    STEREO_ROWS4(0) STEREO_ROWS4(4) STEREO_ROWS4(8) STEREO_ROWS4(12) STEREO_ROWS4(16) STEREO_ROWS4(20) STEREO_ROWS4(24)
    STEREO_ROWS4(28) STEREO_ROWS4(32) STEREO_ROWS4(36) STEREO_ROWS4(40) STEREO_ROWS4(44) STEREO_ROWS4(48) STEREO_ROWS4(52)
    STEREO_ROWS4(56) STEREO_ROWS4(60)
#undef STEREO_ROWS4
  } else {
    unsigned long long todo = need;
    while (todo) {
      const int k0 = __builtin_ctzll(todo);
      todo &= todo - 1;
      const int k1 = todo ? __builtin_ctzll(todo) : k0;  // (a row written twice is the same row)
      todo &= todo - 1;
      const double hk0 = tab[4 * k0], qk0 = tab[4 * k0 + 1], hk1 = tab[4 * k1], qk1 = tab[4 * k1 + 1];
      const unsigned long long m0 = __builtin_amdgcn_ballot_w64(alpha * fabs(qk0 - qs) + hk0 < hs);
      const unsigned long long m1 = __builtin_amdgcn_ballot_w64(alpha * fabs(qk1 - qs) + hk1 < hs);
      // (lane numbers in m0 here; the s_nop keeps the first write four instructions behind the compare
      //  that produced its mask -- see envelope_rows4 -- and the second write follows another three)
      __builtin_amdgcn_sched_barrier(0);
      unsigned keep_m0;   // (m0 is the compiler's; it is put back before the statement ends)
      asm volatile("s_mov_b32 %2, m0\n\ts_mov_b32 m0, %7\n\ts_nop 2\n\tv_writelane_b32 %0, %3, m0\n\tv_writelane_b32 %1, %4, m0\n\t"
                   "s_mov_b32 m0, %8\n\ts_nop 0\n\tv_writelane_b32 %0, %5, m0\n\tv_writelane_b32 %1, %6, m0\n\ts_mov_b32 m0, %2"
                   : "+v"(r1lo), "+v"(r1hi), "=&s"(keep_m0)
                   : "s"((unsigned)m0), "s"((unsigned)(m0 >> 32)), "s"((unsigned)m1), "s"((unsigned)(m1 >> 32)), "s"(k0), "s"(k1));
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  PSTAMP(18);
  const unsigned long long below = lane == 0 ? 0ull : (~0ull >> (64 - lane));
  const unsigned long long R1 = ((((unsigned long long)r1hi) << 32) | r1lo) & below;
  // D_k: exclusive prefix OR of the rows (row_shr 1 2 4 8, row_bcast 15 / 31, then one wave shift)
  unsigned long long D = R1;
  D = dpp_or_u64<0x111, 0xF>(D); D = dpp_or_u64<0x112, 0xF>(D); D = dpp_or_u64<0x114, 0xF>(D);
  D = dpp_or_u64<0x118, 0xF>(D); D = dpp_or_u64<0x142, 0xA>(D); D = dpp_or_u64<0x143, 0xC>(D);
  {
    const int lo = (int)(unsigned)D, hi = (int)(unsigned)(D >> 32);
    const unsigned l2 = (unsigned)__builtin_amdgcn_update_dpp(0, lo, 0x138, 0xF, 0xF, false);  // wave_shr:1
    const unsigned h2 = (unsigned)__builtin_amdgcn_update_dpp(0, hi, 0x138, 0xF, 0xF, false);
    D = ((unsigned long long)h2 << 32) | l2;
  }
  const unsigned long long cand = ~(D | R1) & below;
  // The drop tests of source `lane` against the cone it meets (typeStereoLinear.h:432-449), evaluated
  // up front against its six highest candidates (three at a time: one table read latency for each three): a round of
  // the fixed point below is then a handful of bit operations.  `known` / `dropm`: candidates tested so
  // far / those that drop this source; a round that meets another candidate tests it then.
  // (per-lane bit sets as two 32-bit words: 64-bit shifts and compares are several times as expensive as
  //  32-bit instructions here, and a round of the fixed point is little else)
  const unsigned cand_lo = (unsigned)cand, cand_hi = (unsigned)(cand >> 32);
  unsigned known_lo = 0, known_hi = 0, dropm_lo = 0, dropm_hi = 0;
  auto top_bit = [](unsigned lo, unsigned hi) {   // index of the highest set bit (0 for the empty set: test lo | hi)
    const int a = 31 - __builtin_clz(lo | 1u), b2 = 63 - __builtin_clz(hi | 1u);
    return hi ? b2 : a;
  };
  // tests the (up to) three highest sources of `from` that have not been tested yet
  auto test3 = [&](unsigned from_lo, unsigned from_hi) {
    unsigned r_lo = from_lo & ~known_lo, r_hi = from_hi & ~known_hi;
    int c[3];
    bool real[3];
    double hj[3], qj[3], tj[3];
#pragma unroll
    for (int i = 0; i < 3; ++i) {
      real[i] = (r_lo | r_hi) != 0;
      c[i] = top_bit(r_lo, r_hi);
      const unsigned bit = 1u << (c[i] & 31);
      r_hi &= c[i] >= 32 ? ~bit : ~0u; r_lo &= c[i] >= 32 ? ~0u : ~bit;
      hj[i] = tab[4 * c[i]]; qj[i] = tab[4 * c[i] + 1]; tj[i] = tab[4 * c[i] + 3];
    }
#pragma unroll
    for (int i = 0; i < 3; ++i) {
      const double dist = alpha * fabs(qs - qj[i]);
      const double num = (hs - hj[i]) + alpha * (qs + qj[i]);
      // (| and selects, not || and branches: the compiler turns short-circuit conditions on per-lane
      //  values into nested exec-mask regions, three per candidate here)
      const bool dropped = (dist + hj[i] <= hs) | (num >= thi) | (num <= tj[i]);
      const unsigned bit = real[i] ? 1u << (c[i] & 31) : 0u, dbit = dropped ? bit : 0u;
      const bool up = c[i] >= 32;
      known_hi |= up ? bit : 0u; known_lo |= up ? 0u : bit;
      dropm_hi |= up ? dbit : 0u; dropm_lo |= up ? 0u : dbit;
    }
  };
#ifdef STEREO_HIP_MESSAGE_PROFILE
  const long long pt_a = (long long)__builtin_readcyclecounter();
#endif
  test3(cand_lo, cand_hi);
  test3(cand_lo, cand_hi);   // (six up front: on noisy ramps the cone a source meets is rarely its nearest candidate)
#ifdef STEREO_HIP_MESSAGE_PROFILE
  const long long pt_b = (long long)__builtin_readcyclecounter();
#endif
  unsigned long long P = K >= 64 ? ~0ull : ((1ull << K) - 1);
  unsigned B_lo = 0, B_hi = 0;
  int js = 0;
  bool settled = false;
  int extra_rounds = 0, late_tests = 0;
  for (int round = 0; round <= K + 8; ++round) {
    B_lo = (unsigned)P & cand_lo; B_hi = (unsigned)(P >> 32) & cand_hi;
    const bool empty = (B_lo | B_hi) == 0;
    js = top_bit(B_lo, B_hi);
    const unsigned kw = js >= 32 ? known_hi : known_lo, dw = js >= 32 ? dropm_hi : dropm_lo;
    const bool tested = (kw >> (js & 31)) & 1, drops = (dw >> (js & 31)) & 1;
    // a source meets a cone it has not been tested against (its higher candidates were all dropped):
    // that one and the next two that are still in P, which is where the following rounds tend to land
    // (both masks are formed before either is looked at: one trip of the vector results to the scalar side per round)
    const unsigned long long late = __builtin_amdgcn_ballot_w64(act & !empty & !tested);
    const unsigned long long np = __builtin_amdgcn_ballot_w64(act & (empty | !drops)) | 1ull;
    if (late) { test3(B_lo, B_hi); ++late_tests; continue; }
    if (np == P) { settled = true; break; }
    P = np;
    ++extra_rounds;
  }
  const unsigned long long B = ((unsigned long long)B_hi << 32) | B_lo;
  (void)extra_rounds; (void)nrows; (void)late_tests;  // (read by the profile flavour only)
  PSTAMP(20);
#ifdef STEREO_HIP_MESSAGE_PROFILE
  if (prof && lane == 0) {
    atomicAdd(prof + 24, (unsigned long long)extra_rounds);                    // extra rounds of the fixed point
    atomicAdd(prof + 26, (unsigned long long)__builtin_popcountll(P));         // pushed sources
    atomicAdd(prof + 27, (unsigned long long)nrows);                           // rows of the pop test that were computed
    atomicAdd(prof + 28, (unsigned long long)late_tests);                      // rounds that had to test more candidates
    atomicAdd(prof + 29, (unsigned long long)(pt_b - pt_a));                   // cycles of the six up-front drop tests
    atomicAdd(prof + 30, (unsigned long long)((long long)__builtin_readcyclecounter() - pt_b));  // ... of the rounds
  }
#endif
  if (!settled) return false;
  // the cones a source dominates must be what the serial code pops: a top segment of its stack
  {
    const unsigned long long S = P & ~D & below;
    const bool bad = act && B != 0 && (S & R1 & ((1ull << js) - 1)) != 0;
    if (UNI(bad)) {
#ifdef STEREO_HIP_MESSAGE_PROFILE
      if (prof && lane == 0) atomicAdd(prof + 25, 1ull);  // top-segment check failed
#endif
      return false;
    }
  }
  const bool pushed = act && ((P >> lane) & 1);
  const int t = B ? __builtin_popcountll(B) : 0;
  if (pushed) {
    if (lane > 0) atomicMax(scr + 2 * t, lane << 8);
    if (B) atomicMax(scr + 2 * (t - 1) + 1, (lane << 8) | js);
  }
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
  __builtin_amdgcn_wave_barrier();
  const int src = scr[2 * lane] >> 8, zw = scr[2 * lane + 1];
  const bool have_z = (zw >> 8) > src;
  const int zk = have_z ? zw >> 8 : 0, zj = zw & 255;
  sh = tab[4 * src]; sq = tab[4 * src + 1];
  const double hk = tab[4 * zk], qk = tab[4 * zk + 1];
  const double hj = tab[4 * zj], qj = tab[4 * zj + 1];
  // (2 alpha a power of two: the quotient by multiplication, exact either way)
  const double num = (hk - hj) + alpha * (qk + qj);
  const double s = exact_c ? num * (1.0 / (2 * alpha)) : num / (2 * alpha);
  zz = have_z ? s : inf;
  // slots fill up from 0: the highest slot ever written is the highest one that holds a source
  maxtop_out = 63 - __builtin_clzll(__builtin_amdgcn_ballot_w64(src != 0) | 1ull);
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
  __builtin_amdgcn_wave_barrier();
  PSTAMP(22);
#ifdef STEREO_HIP_MESSAGE_PROFILE
  if (prof && lane == 0)
    for (int i = 0; i < 4; ++i) { atomicAdd(prof + 16 + 2 * i, (unsigned long long)pacc[i]); atomicAdd(prof + 17 + 2 * i, 1ull); }
#endif
  return true;
}
#undef PSTAMP

// Certified fast path of the truncated QUADRATIC message (typeStereoQuadratic.h:329-501), K <= 64,
// lane = source and destination label.  The reference builds the lower envelope of the parabolas
// alpha (t - q_s)^2 + h_s as the lower convex hull of the points (q_s, g_s = h_s + alpha q_s^2) --
// its breakpoint s = (g_k - g_j) / (2 alpha (q_k - q_j)) is the hull slope over 2 alpha -- by a
// monotone-chain scan, then picks for destination t the stack slot with z[slot] < t <= z[slot+1].
// Suppose that at t the smallest cost c_j(t) is separated from every other source's cost by more
// than delta.  Since c_s(t) - c_j(t) = 2 alpha (q_s - q_j) (sigma(j,s) - t), every exact slope from
// j to a later source exceeds t + delta / (2 alpha Q) and every slope from an earlier source to j is
// below t - delta / (2 alpha Q) (Q = span of the source positions).  If the rounding error of any
// computed breakpoint that involves j (<= ~1e-14 G / (alpha gap), G >= |g|, gap = distance from
// a useful source to the nearest other source) is smaller than that margin, then (i) j is pushed
// when its turn comes (no near-duplicate position: gap > 1e-8 regime), (ii) no later source pops it
// (its breakpoint against j stays above j's own), and (iii) j's two breakpoints on the final stack
// bracket t; breakpoints increase strictly along the stack by construction (a push requires
// s > z[top]), so the walk stops at j: the reference returns exactly alpha (t-q_j)^2 + h_j, the
// plain min-plus value.  Sources with h >= vTrunc cost >= vTrunc everywhere: they are covered by
// the margin to vTrunc.  Destinations whose minimum is >= vTrunc return vTrunc whatever is picked.
// Returns "needs the serial construction"; m1 = min-plus value over the useful sources.
__device__ __forceinline__ bool message_quad_fast(double lambda, int K, double alpha, double h, double qsrc,
                                                  double t, double vtrunc, int lane, const double *hq,
                                                  double &m1_out, int window = -1, double shared_gap = 0) {
  const double inf = __builtin_huge_val();
  const bool act = lane < K;
  double scale = act ? fabs(h) + alpha * qsrc * qsrc + alpha * t * t : 0.0;  // >= |g| and >= cost / 4
  scale = wave_max_dpp(scale);
  double qlo = act ? qsrc : inf, qhi = act ? qsrc : -inf;
  wave_min_max_dpp(qlo, qhi);  // smallest and largest source position
  const double delta = 1e-9 * (scale + fabs(alpha * lambda) + fabs(vtrunc));
  unsigned long long mask = __builtin_amdgcn_ballot_w64(act && h < vtrunc);
  double m1 = inf, m2 = inf;
  // `gap` must not become a wave-uniform constant: this compiler (AMD clang 22, gfx950) then merges it
  // with the uniform `shared_gap` in scalar registers and emits s_mov_b64 with a 64-bit literal, which
  // the encoder truncates to its low half (+inf -> 0.0).  build.sh greps the ISA for that pattern.
  double gap = inf;
  asm volatile("" : "+v"(gap));
  if (window >= 0 && __builtin_popcountll(mask) > 2 * window + 1) {
    // shared strictly ascending positions (padded table): a source more than `window` indices away
    // lies farther than sqrt(lambda (1 + 1e-9)) and costs >= vTrunc bit for bit (alpha > 0; the two
    // roundings of alpha d d lose less than the 1e-9), so it is covered by the margin to vTrunc
    for (int d = -window; d <= window; ++d) {
      const double hj = hq[4 * (lane + d)], qj = hq[4 * (lane + d) + 1];
      const double c = pair_cost<2>(alpha, t - qj, hj);
      const double lo = min_raw(m1, c), hi = max_raw(m1, c);
      m2 = min_raw(m2, hi);
      m1 = lo;
    }
    gap = shared_gap;
    mask = 0;
  }
  while (mask) {
    const int j = __builtin_ctzll(mask);
    mask &= mask - 1;
    double hj, qj;
    if (hq) { hj = hq[4 * j]; qj = hq[4 * j + 1]; }
    else { hj = readlane_f64(h, j); qj = readlane_f64(qsrc, j); }
    const double c = pair_cost<2>(alpha, t - qj, hj);
    const double lo = min_raw(m1, c), hi = max_raw(m1, c);
    m2 = min_raw(m2, hi);  // second smallest, equal costs of two sources count
    m1 = lo;
    const double dq = fabs(qsrc - qj);
    gap = min_raw_if(lane != j, gap, dq);
  }
  gap = wave_min_dpp(act ? gap : inf);
  bool bad = !(delta < inf) || !(alpha > 0) || !(gap > 4e-8);
  // breakpoint error <= ~7 eps G / (alpha gap) must stay below the slope margin delta / (2 alpha Q):
  // delta gap > 1.6e-15 G Q, tested with a factor 60 in hand
  bad = bad || !(1e-13 * scale * (qhi - qlo) < delta * gap);
  bad = bad || (m1 < vtrunc && !(m2 - m1 > delta && vtrunc - m1 > delta));
  m1_out = m1;
  return UNI(act && bad);
}

constexpr int kMaxSlots = TrwsGraph::kMaxSlots;
constexpr int kSpinLimit = 1 << 22;  // polls before a wait INSIDE a workgroup (LDS flags) gives up
constexpr int kCoopSpinLimit = 1 << 24;  // ... before a wave stops waiting for its partner's half of a message (a bug, not a state: the
                                         // partner runs the same code on the same inputs); the result is then wrong and the parity tests say so

// Waiting for ANOTHER workgroup -- with row strips possibly another process on another GPU, whose
// launch may start late -- is bounded by wall-clock time (DevParams::spin_ticks of the 100 MHz
// s_memrealtime counter; STEREO_HIP_TRWS_SPIN_SECONDS), not by a poll count: one step of such a wait.
// Every 1024 polls it looks at the abort flag (somebody else gave up) and at the clock.
// `halo`: the wait is for a node of a neighbouring strip.  Waits for the strip's own nodes get twice the
// time: when a neighbour is missing, the visit that waits for IT gives up first and names the cause,
// the visits queued up behind it inside the strip see the abort flag and leave quietly.
// (the poll counter wraps inside [0, 1024): nothing that could overflow however long the bound is; the
//  clock is read at every wrap, the first reading -- t0 == 0: not started -- starts the bound, so the
//  first 1024 polls, about a millisecond, come on top of it and a wait that ends earlier never reads it)
__device__ __forceinline__ bool keep_waiting(const DevParams &p, int &spins, long long &t0, bool halo) {
  __builtin_amdgcn_s_sleep(1);
  spins = (spins + 1) & 1023;
  if (spins != 0) return true;
  if (ld_sc1(p.abort_flag)) return false;
  const long long now = (long long)wall_clock64();
  if (t0 == 0) { t0 = now | 1; return true; }
  return now - t0 < (halo ? p.spin_ticks : 2 * p.spin_ticks);
}
// The first visit that gives up says what it was waiting for (the host turns it into the error text):
// abort_flag[1..4] = visiting rank, awaited rank, value seen in its flag, epoch expected.
__device__ __forceinline__ void report_give_up(const DevParams &p, int visiting_rank, int awaited_rank, int seen, int epoch) {
  if (atomicCAS(p.abort_flag, 0, 1) == 0) {
    st_sc1(p.abort_flag + 1, visiting_rank); st_sc1(p.abort_flag + 2, awaited_rank);
    st_sc1(p.abort_flag + 3, seen); st_sc1(p.abort_flag + 4, epoch);
  }
}

// ---- fast persistent sweep: K <= 64, <= 4 edges per list, <= 2 foreign dependencies ----
// Same dataflow schedule as trws_persistent_kernel, restructured so that in
// steady state a node visit touches no memory on its critical path:
//  * lane = label; every wave keeps D, the outgoing-list messages and the
//    incoming messages of the node in registers and forms Di redundantly, so the
//    only workgroup traffic is the LDS hand-over of the new messages;
//  * a packed 128-byte descriptor per processing position replaces the chains of
//    dependent index loads; descriptor, unary and previous-sweep messages of the
//    NEXT node are fetched while the current node's messages are computed, and
//    so are the foreign incoming messages once their flags are seen raised;
//  * a node's completion flag is raised in the middle of the next visit, when
//    its write-through stores have long drained, so no store latency is exposed.
struct NodeDesc {
  int node, rank, nout, nin, ndep, md, lbn, urgent, remote, epos;
  int e[8], slot[8], dep[4], lbe[8], xn[8];
  int re[8], pn[2];  // strips: edge / node ids in the neighbouring strip's numbering (only the storers read them)
};
#define RLI(v, i) __builtin_amdgcn_readlane((v), (i))
__device__ __forceinline__ NodeDesc decode_desc(int w) {
  NodeDesc d;
  d.node = RLI(w, 0); d.rank = RLI(w, 1);
  const int f = RLI(w, 2);
  d.nout = f & 15; d.nin = (f >> 4) & 15; d.ndep = (f >> 8) & 15; d.md = (f >> 16) & 255;
  d.lbn = RLI(w, 3);
  d.e[0] = RLI(w, 4); d.e[1] = RLI(w, 5); d.e[2] = RLI(w, 6); d.e[3] = RLI(w, 7);
  d.e[4] = RLI(w, 8); d.e[5] = RLI(w, 9); d.e[6] = RLI(w, 10); d.e[7] = RLI(w, 11);
  d.slot[0] = RLI(w, 12); d.slot[1] = RLI(w, 13); d.slot[2] = RLI(w, 14); d.slot[3] = RLI(w, 15);
  d.slot[4] = RLI(w, 16); d.slot[5] = RLI(w, 17); d.slot[6] = RLI(w, 18); d.slot[7] = RLI(w, 19);
  d.dep[0] = RLI(w, 20); d.dep[1] = RLI(w, 21); d.dep[2] = RLI(w, 22); d.dep[3] = RLI(w, 23);
  d.lbe[0] = RLI(w, 24); d.lbe[1] = RLI(w, 25); d.lbe[2] = RLI(w, 26); d.lbe[3] = RLI(w, 27);
  d.lbe[4] = RLI(w, 28); d.lbe[5] = RLI(w, 29); d.lbe[6] = RLI(w, 30); d.lbe[7] = RLI(w, 31);
  d.xn[0] = RLI(w, 32); d.xn[1] = RLI(w, 33); d.xn[2] = RLI(w, 34); d.xn[3] = RLI(w, 35);
  d.xn[4] = RLI(w, 36); d.xn[5] = RLI(w, 37); d.xn[6] = RLI(w, 38); d.xn[7] = RLI(w, 39);
  d.urgent = RLI(w, 40);
  d.remote = RLI(w, kDescRemote); d.epos = RLI(w, kDescEpos);
  d.re[0] = RLI(w, 45); d.re[1] = RLI(w, 46); d.re[2] = RLI(w, 47); d.re[3] = RLI(w, 48);
  d.re[4] = RLI(w, 49); d.re[5] = RLI(w, 50); d.re[6] = RLI(w, 51); d.re[7] = RLI(w, 52);
  d.pn[0] = RLI(w, 53); d.pn[1] = RLI(w, 54);
  return d;
}

// bounded wait for one completion flag; returns false if the launch must give up
__device__ __forceinline__ bool wait_flag(const DevParams &p, int rank, int epoch, int visiting_rank = -1) {
  const int32_t *flag = p.done + rank;
  int spins = 0, v;
  long long t0 = 0;
  while ((v = ld_sc1(flag)) < epoch) {
    if (!keep_waiting(p, spins, t0, rank >= p.n_own)) {
      report_give_up(p, visiting_rank, rank, v, epoch);
      return false;
    }
  }
  return true;
}

// The loaders' wait for a node's foreign dependencies: lane j watches the completion flag of
// dependency j (all flags polled together), bounded by the wall clock (keep_waiting); a wait that
// gives up reports what it waited for and raises the workgroup's abort word.
// (the descriptor's fields by value: a NodeDesc passed by reference is materialised in scratch memory)
__device__ __forceinline__ void wait_for_dependencies(const DevParams &p, int ndep, int dep0, int dep1, int dep2, int dep3,
                                                      int visiting_rank, int epoch, int lane, int *abort_word) {
  if (ndep <= 0) return;
  const int myrank = lane == 1 ? dep1 : lane == 2 ? dep2 : lane == 3 ? dep3 : dep0;
  const bool watching = lane < ndep;
  int spins = 0;
  long long t0 = 0;
  for (;;) {
    const int v = watching ? ld_sc1(p.done + myrank) : epoch;
    if (!UNI(v < epoch)) break;
    const unsigned long long late = __builtin_amdgcn_ballot_w64(v < epoch),
                             late_halo = __builtin_amdgcn_ballot_w64(v < epoch && myrank >= p.n_own);
    if (!keep_waiting(p, spins, t0, late_halo != 0)) {   // wall-clock bound, or somebody else gave up
      if (lane == __builtin_ctzll(late_halo ? late_halo : late)) report_give_up(p, visiting_rank, myrank, v, epoch);
      if (lane == 0) *abort_word = 1;
      return;
    }
  }
  // The flags are up: what the caller reads next (the neighbours' messages and labels, sc1 loads that
  // go to memory) must not be moved in front of the flag loads by the compiler.  The hardware issues a
  // wave's loads in order and the branch above has waited for the flags' values, so a wavefront-scope
  // acquire -- a compiler barrier, no cache operation -- completes the hand-over's consumer side; the
  // producer drains its write-through stores (s_waitcnt vmcnt(0)) before it stores the flag.
  // CONTRACT (this is weaker than the memory model's agent- / system-scope acquire, which would emit a
  // buffer_inv sc1 -- ~1.7 us per wait on this part, MI355X_MICROARCH.md -- on every visit of every row):
  // every load of data ANOTHER workgroup / GPU wrote during this launch goes through ld_sc1 (sc0 sc1:
  // served from memory, never from this CU's L1 or a non-coherent L2 line).  The loaders keep to it by
  // construction -- loader B (this routine's only caller besides wait_flag) reads nothing but foreign
  // rows and labels, all with ld_sc1; loader A reads only data no other workgroup writes in the launch
  // (unary, positions, weights, the node's OWN previous-sweep rows) with plain loads -- and
  // tools/stress_trws.py / the strips tests compare whole solves bit for bit on every commit that
  // touches a loader.  A new post-wait read of handed-over data with a plain load would break it silently.
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// The same wait with the lane's own dependency already in the lane (`myrank`: lane j < ndep watches
// dependency j -- the caller takes it from the descriptor word with one ds_bpermute instead of four scalar
// reads and a chain of selects).
__device__ __forceinline__ void wait_for_dependencies_w(const DevParams &p, int ndep, int myrank, int visiting_rank, int epoch,
                                                        int lane, int *abort_word) {
  const bool watching = lane < ndep;
  int spins = 0;
  long long t0 = 0;
  for (;;) {
    const int v = watching ? ld_sc1(p.done + myrank) : epoch;
    if (!UNI(v < epoch)) break;
    const unsigned long long late = __builtin_amdgcn_ballot_w64(v < epoch),
                             late_halo = __builtin_amdgcn_ballot_w64(v < epoch && myrank >= p.n_own);
    if (!keep_waiting(p, spins, t0, late_halo != 0)) {   // wall-clock bound, or somebody else gave up
      if (lane == __builtin_ctzll(late_halo ? late_halo : late)) report_give_up(p, visiting_rank, myrank, v, epoch);
      if (lane == 0) *abort_word = 1;
      return;
    }
  }
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");   // (contract: see wait_for_dependencies)
}

// ---- lane exchange lane ^ S without an address register where the hardware offers one: a DPP move for 1, 2 and 8,
// two of them for 4 (row_half_mirror is i -> i ^ 7 inside every eight lanes, the reversed quad i -> i ^ 3), the LDS
// crossbar for 16 and 32.  (mov_dpp with bound_ctrl and all rows / banks enabled: one instruction, no copy of the source.)
template <int S>
__device__ __forceinline__ unsigned xor_lane_u32(unsigned v) {
  if (S == 1) return (unsigned)__builtin_amdgcn_mov_dpp((int)v, 0xB1, 0xF, 0xF, true);   // quad_perm [1,0,3,2]
  if (S == 2) return (unsigned)__builtin_amdgcn_mov_dpp((int)v, 0x4E, 0xF, 0xF, true);   // quad_perm [2,3,0,1]
  if (S == 8) return (unsigned)__builtin_amdgcn_mov_dpp((int)v, 0x128, 0xF, 0xF, true);  // row_ror:8
  if (S == 4) return (unsigned)__builtin_amdgcn_mov_dpp(__builtin_amdgcn_mov_dpp((int)v, 0x141, 0xF, 0xF, true), 0x1B, 0xF, 0xF, true);
  if (S == 16) return (unsigned)__builtin_amdgcn_ds_swizzle((int)v, 0x1F | (S << 10));   // bit mode: xor S
  return (unsigned)__shfl_xor((int)v, S, kWave);
}
// Bitonic sort of two independent sets of 64 unsigned keys (one key of each per lane), ascending by lane.
// A compare-exchange keeps the smaller or the larger of (own, partner's) -- which one is a property of the lane and
// the step -- and both are the MEDIAN of (own, partner's, c) with c = 0 resp. 0xffffffff: one v_med3_u32 per key (the
// compiler matches max(min(a, b), min(max(a, b), c))) instead of min + max + select on a 64-bit lane mask per step
// (21 masks: 42 scalar registers, spilled and read back lane by lane).  c comes out of one per-lane word of 21 bits,
// one signed bit-field extract per step.  5 instructions per step and pair of keys, 12 before.
__device__ __forceinline__ unsigned med3_u32(unsigned a, unsigned b, unsigned c) {
  const unsigned lo = a < b ? a : b, hi = a < b ? b : a;
  const unsigned t = hi < c ? hi : c;
  return lo > t ? lo : t;
}
// bit k (k - 1) / 2 + j of the word: the step with blocks of 2^k and stride 2^j keeps the LARGER key in this lane
__device__ __forceinline__ int bitonic_keep_bits(int lane) {
  int km = 0;
#pragma unroll
  for (int k = 1; k <= 6; ++k) {
    const int w = lane ^ -((lane >> k) & 1);   // bit j: lane bit j != lane bit k
    km |= (w & ((1 << k) - 1)) << (k * (k - 1) / 2);
  }
  return km;
}
template <int LOG_SIZE, int LOG_STRIDE>
__device__ __forceinline__ void bitonic_step2(unsigned &a, unsigned &b, int km) {
  const unsigned c = (unsigned)__builtin_amdgcn_sbfe(km, LOG_SIZE * (LOG_SIZE - 1) / 2 + LOG_STRIDE, 1);
  if (LOG_STRIDE >= 4) {
    // strides 16 and 32: gfx950's v_permlane16_swap / v_permlane32_swap exchange the odd rows (the upper half) of one
    // register with the even rows (the lower half) of another; fed the key twice they leave (own, partner's) in every
    // lane of the pair -- in which order does not matter to a median -- without the LDS crossbar's round trip
    const auto ra = LOG_STRIDE == 4 ? __builtin_amdgcn_permlane16_swap(a, a, false, false) : __builtin_amdgcn_permlane32_swap(a, a, false, false);
    const auto rb = LOG_STRIDE == 4 ? __builtin_amdgcn_permlane16_swap(b, b, false, false) : __builtin_amdgcn_permlane32_swap(b, b, false, false);
    a = med3_u32(ra[0], ra[1], c);
    b = med3_u32(rb[0], rb[1], c);
    return;
  }
  const unsigned oa = xor_lane_u32<(1 << LOG_STRIDE)>(a), ob = xor_lane_u32<(1 << LOG_STRIDE)>(b);
  a = med3_u32(a, oa, c);
  b = med3_u32(b, ob, c);
}
__device__ __forceinline__ void wave_sort2(unsigned &a, unsigned &b, int lane) {
  const int km = bitonic_keep_bits(lane);
  bitonic_step2<1, 0>(a, b, km);
  bitonic_step2<2, 1>(a, b, km); bitonic_step2<2, 0>(a, b, km);
  bitonic_step2<3, 2>(a, b, km); bitonic_step2<3, 1>(a, b, km); bitonic_step2<3, 0>(a, b, km);
  bitonic_step2<4, 3>(a, b, km); bitonic_step2<4, 2>(a, b, km); bitonic_step2<4, 1>(a, b, km); bitonic_step2<4, 0>(a, b, km);
  bitonic_step2<5, 4>(a, b, km); bitonic_step2<5, 3>(a, b, km); bitonic_step2<5, 2>(a, b, km); bitonic_step2<5, 1>(a, b, km);
  bitonic_step2<5, 0>(a, b, km);
  bitonic_step2<6, 5>(a, b, km); bitonic_step2<6, 4>(a, b, km); bitonic_step2<6, 3>(a, b, km); bitonic_step2<6, 2>(a, b, km);
  bitonic_step2<6, 1>(a, b, km); bitonic_step2<6, 0>(a, b, km);
}

// Second look at a message whose certificate failed (cold path, kept out of line so that it costs the
// hot path no registers).  A cone whose apex lies above vTrunc by more than alpha times the whole
// position range cannot touch a useful cone -- every useful cone dominates it with that margin
// wherever they meet -- so it neither counts for the magnitude behind delta nor for the tangency
// test.  Out-of-range plane proposals (unary ~ 4e7, dispmap_ncc.m:245) would otherwise inflate delta
// and send almost every message of such a fusion to the serial construction.  Returns "still bad";
// m1 = min-plus value over the useful sources.
// (the cheap part -- does leaving out the far cones shrink delta at all? -- is inline at the call site:
//  a call costs this kernel ~10 k cycles of register traffic, and on volumes whose failures are exact
//  ties, e.g. the flat columns of an NCC volume, it never does)
__device__ __forceinline__ bool second_look_applies(double lambda, int K, double alpha, double h, double qsrc, double t,
                                                    double vtrunc, double delta, int lane, double &delta2_out, bool &rel_out) {
  const bool act = lane < K;
  const double aq = alpha * qsrc;
  const double qabs = wave_max_dpp(act ? max_raw(fabs(qsrc), fabs(t)) : 0.0);
  const bool rel = act && h <= vtrunc + 2.000002 * fabs(alpha) * qabs;
  const double mag2 = max_raw(wave_max_dpp(act ? (rel ? fabs(h) : 0.0) + fabs(aq) + alpha * fabs(t) : 0.0), fabs(vtrunc));
  const double delta2 = 1e-9 * (mag2 + fabs(alpha * lambda));
  delta2_out = delta2; rel_out = rel;
  return delta2 < delta;
}
__device__ __attribute__((noinline)) bool message_second_look(int K, double alpha, double h, double qsrc, double t,
                                                              double vtrunc, double delta2, bool rel, int lane,
                                                              double &m1_out) {
  const double inf = __builtin_huge_val();
  const bool act = lane < K;
  const double aq = alpha * qsrc;
  const double ui = h - aq, vi = h + aq;
  unsigned long long mask = __builtin_amdgcn_ballot_w64(act && h < vtrunc);
  double m1 = inf, m2 = inf;
  bool bad = false;
  while (mask) {
    const int j = __builtin_ctzll(mask);
    mask &= mask - 1;
    const double hj = readlane_f64(h, j), qj = readlane_f64(qsrc, j);
    const double c = pair_cost<1>(alpha, t - qj, hj);
    const double lo = min_raw(m1, c), hi = max_raw(m1, c);
    m2 = min_raw_if(hi > lo, m2, hi);
    m1 = lo;
    const double aqj = alpha * qj;
    const bool near = (fabs(ui - (hj - aqj)) <= delta2) || (fabs(vi - (hj + aqj)) <= delta2);
    bad = bad || (near && qsrc != qj && rel);
  }
  bad = bad || (m1 < vtrunc && !(m2 - m1 > delta2 && vtrunc - m1 > delta2));
  m1_out = m1;
  return UNI(act && bad);
}

// Evaluation of a constructed envelope at destination t (typeStereoLinear.h:462-479): the reference
// walks up the stack while z[j+1] < t, so the slot it stops at is the FIRST one whose upper breakpoint
// is not below t (stale slots above `top` included, hence up to the highest slot ever written) --
// which is also the first slot where the RUNNING MAXIMUM of the breakpoints reaches t, and a running
// maximum can be bisected: a 64-lane max-scan and six lane reads per destination instead of one pass
// per stack slot (the stack holds most of the K cones on ramp-like data).  Slots from maxtop on count
// as +inf, so the search ends at maxtop when nothing below stops it.
template <int KERNEL>
__device__ __forceinline__ double envelope_value(const DevParams &p, double alpha, double t, double vtrunc, double sh,
                                                 double sq, double zz, int maxtop, int lane, double *tab = nullptr) {
  const double inf = __builtin_huge_val();
  int slot;
  double ch, cq;
  if (p.debug & 1024) {
    slot = maxtop;
    for (int j = maxtop - 1; j >= 0; --j) {
      const double zj1 = readlane_f64(zz, j);
      slot = !(zj1 < t) ? j : slot;
    }
    ch = __shfl(sh, slot, kWave); cq = __shfl(sq, slot, kWave);
  } else {
    double zm = (lane >= maxtop || zz != zz) ? inf : zz;  // (a NaN breakpoint stops the walk like +inf does)
    if (tab) {
      // running maximum by DPP (no LDS round trips), then the slots' (running maximum, height, position)
      // go to this wave's table and the bisection reads them from there: one ds_read per probe instead of
      // two ds_bpermute, and one for the answer
#define STEREO_SCAN(C, M) { const double o = dpp_f64_keep<C, M>(zm); zm = max_raw(o, zm); }
      STEREO_SCAN(0x111, 0xF) STEREO_SCAN(0x112, 0xF) STEREO_SCAN(0x114, 0xF) STEREO_SCAN(0x118, 0xF)
      STEREO_SCAN(0x142, 0xA) STEREO_SCAN(0x143, 0xC)
#undef STEREO_SCAN
      tab[4 * lane] = zm; tab[4 * lane + 1] = sh; tab[4 * lane + 2] = sq;
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      // (three probes per round, requested together: 64 -> 16 -> 4 -> 1 candidates in three LDS latencies instead
      //  of six; the running maximum is non-decreasing, so "how many of the three probes lie below t" is the step)
      slot = 0;
#pragma unroll
      for (int step = kWave / 4; step >= 1; step >>= 2) {
        const double pr1 = tab[4 * (slot + step - 1)], pr2 = tab[4 * (slot + 2 * step - 1)], pr3 = tab[4 * (slot + 3 * step - 1)];
        slot += ((pr1 < t ? 1 : 0) + (pr2 < t ? 1 : 0) + (pr3 < t ? 1 : 0)) * step;
      }
      ch = tab[4 * slot + 1]; cq = tab[4 * slot + 2];
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
      __builtin_amdgcn_wave_barrier();
    } else {
#pragma unroll
      for (int o = 1; o < kWave; o <<= 1) {
        const double up = __shfl_up(zm, o, kWave);
        zm = (lane >= o && up > zm) ? up : zm;
      }
      slot = 0;
#pragma unroll
      for (int step = kWave / 2; step >= 1; step >>= 1) {
        const double probe = __shfl(zm, slot + step - 1, kWave);
        slot = probe < t ? slot + step : slot;
      }
      ch = __shfl(sh, slot, kWave); cq = __shfl(sq, slot, kWave);
    }
  }
  const double c = pair_cost<KERNEL>(alpha, t - cq, ch);
  return c < vtrunc ? c : vtrunc;
}

// Two waves, one message (trws_pipe_kernel, shared positions).  The reference's neighbourhood holds every pair of
// pixels as TWO directed edges; with equal weights and one positions vector both carry the same message (same Di,
// same gamma, same old message by induction from zero), so the second wave of such a pair -- on another SIMD --
// takes half of the useful-source loop (or the window loop while the first sorts the tangency keys) instead of
// computing the same message again.  Everything up to the loop is computed by both (same inputs: same bits, same
// decisions, hence no coordination), the second wave leaves its partial minima / match counts in LDS behind a flag,
// the first combines, judges the certificate, falls back to the serial construction if it must, and writes both rows.
// The visit's barrier separates one exchange from the next.
constexpr int kPipeCompute = 8;  // one compute wave per outgoing message (<= 8 per node)
constexpr int kPipeWaves = kPipeCompute + 4;  // loader (data behind flags), storer, loader A (own data, two visits
                                              // deep; trws_pipe_kernel only), primal
constexpr int kPipeThreads = kPipeWaves * kWave;
// LDS stage layout (doubles): D[64] m[8][64] qv[8][64] qpv[8][64] | a[8] gamma pad | ints: desc[64] px[8] row[8]
constexpr int kStD = 0, kStM = 64, kStQ = 64 + 512, kStQP = 64 + 1024, kStA = 64 + 1536;
constexpr int kStG = kStA + 8;                    // gamma = 1 / max(n_out, n_in) of the node (the loader's division)
constexpr int kStI = kStA + 10;                   // int area starts here (as doubles)
constexpr int kStageDoubles = kStI + 40;          // ints: desc[64] px[8] row[8] (where Di's k-th message row lives in LDS)
constexpr int kScalDoubles = 16;                  // newv[8], node_vmin, prim_e, x (as int)
constexpr int kPipePad = 16;                      // source tables are padded by this many (+inf) entries on both sides
constexpr int kPipeScr = 66;                      // doubles of scratch behind a wave's table: 129 ints of build_envelope_parallel
constexpr int kPipeTab = 4 * (kWave + 2 * kPipePad) + kPipeScr;  // doubles per compute wave: (h, q, u, v) x 96 + scratch
constexpr int kPipeXchg = 2 * kWave + kWave / 2;  // doubles of a helper's exchange area: m1[64], m2[64], 64 ints (CoopPart)

// where the exchange areas and their flags live in trws_pipe_kernel's LDS (doubles from the start of the dynamic
// allocation; pipe_body carves it in this order): the message routine forms the addresses from these constants and ONE
// scalar word instead of carrying pointers through its whole length (scalar registers spilled into VGPR lanes cost an
// instruction per use on the critical path of every visit)
constexpr int kPipeCtlOff = 2 * kStageDoubles + 4 * 8 * kWave + 2 * kScalDoubles + kPipeCompute * kPipeTab;   // ctl words: [0] run, [1] abort, [2] verdict
constexpr int kPipeXchgOff = kPipeCtlOff + 2 + 3 * kWave / 2 + kWave + 16;
constexpr int kPipeXflagOff = kPipeXchgOff + kPipeCompute * kPipeXchg;
constexpr int kPipeLdsDoubles = kPipeXflagOff + kPipeCompute / 2;   // (the runner of the speculative schedule, trws_spec.h, has its LDS behind)

// Min-plus of a destination over the table entries lane + d0 .. lane + d1 (the truncation window or a part of it; the
// table is padded with +inf entries): four entries per trip, requested together -- one LDS latency per trip, not per entry.
// m1 / m2: smallest and second smallest DISTINCT cost so far.
__device__ __forceinline__ void window_minplus(const double *hq, double alpha, double t, int lane, int d0, int d1, double &m1, double &m2) {
#define STEREO_WIN(HJ, QJ)                                         \
  {                                                                \
    const double c = pair_cost<1>(alpha, t - (QJ), (HJ));          \
    const double lo = min_raw(m1, c), hi = max_raw(m1, c);         \
    m2 = min_raw_if(hi > lo, m2, hi);                              \
    m1 = lo;                                                       \
  }
  int d = d0;
  for (; d + 3 <= d1; d += 4) {
    const double *e = hq + 4 * (lane + d);
    const double h0 = e[0], q0 = e[1], h1 = e[4], q1 = e[5], h2 = e[8], q2 = e[9], h3 = e[12], q3 = e[13];
    STEREO_WIN(h0, q0) STEREO_WIN(h1, q1) STEREO_WIN(h2, q2) STEREO_WIN(h3, q3)
  }
  for (; d <= d1; ++d) {
    const double hj = hq[4 * (lane + d)], qj = hq[4 * (lane + d) + 1];
    STEREO_WIN(hj, qj)
  }
#undef STEREO_WIN
}

// more useful sources than this on shared ascending positions with a window: key sort + window loop instead of the pair loop
#ifndef STEREO_FLAT_FROM
#define STEREO_FLAT_FROM 32
#endif
#ifndef STEREO_FLAT_OWN
#define STEREO_FLAT_OWN 0
#endif
constexpr int kFlatFrom = STEREO_FLAT_FROM;
struct CoopPart {
  int word = 0;   // bit 0: sharing; bit 1: this wave is the helper; bits 4-7: the helper's wave; bits 8-: what the flag must show
  __device__ __forceinline__ bool active() const { return word & 1; }
  __device__ __forceinline__ bool part() const { return (word >> 1) & 1; }
};
__device__ __forceinline__ void coop_publish(const CoopPart &c, double m1, double m2, int cnt, int lane) {
  extern __shared__ __attribute__((aligned(16))) double coop_lds[];
  double *xd = coop_lds + kPipeXchgOff + ((c.word >> 4) & 15) * kPipeXchg;
  xd[lane] = m1; xd[kWave + lane] = m2; ((int *)(xd + 2 * kWave))[lane] = cnt;
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
  if (lane == 0) __hip_atomic_store((int *)(coop_lds + kPipeXflagOff) + ((c.word >> 4) & 15), c.word >> 8, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}
// (the second smallest DISTINCT cost of the union: the part whose minimum is the overall minimum contributes its
//  second smallest, the other its minimum)
__device__ __forceinline__ void coop_collect(const CoopPart &c, double &m1, double &m2, int &cnt, int lane) {
  extern __shared__ __attribute__((aligned(16))) double coop_lds[];
  const double *xd = coop_lds + kPipeXchgOff + ((c.word >> 4) & 15) * kPipeXchg;
  int *flag = (int *)(coop_lds + kPipeXflagOff) + ((c.word >> 4) & 15);
  int spins = 0;
  while (__hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) != (c.word >> 8) && ++spins < kCoopSpinLimit) { }
  // (a partner that never publishes is a bug, not a state; what would be merged then is not a message: the launch gives up
  //  through the workgroup's abort word, like a dependency wait that ran out of time)
  if (spins >= kCoopSpinLimit && lane == 0) __hip_atomic_store((int *)(coop_lds + kPipeCtlOff) + 1, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
  const double b1 = xd[lane], b2 = xd[kWave + lane];
  cnt += ((const int *)(xd + 2 * kWave))[lane];
  const double lo = min_raw(m1, b1);
  const double ca = m1 > lo ? m1 : m2, cb = b1 > lo ? b1 : b2;
  m2 = min_raw(ca, cb);
  m1 = lo;
}

// Second look at a certificate whose tangency test found a cone within delta of an arm of a useful cone j
// (shared strictly ascending positions: lane t = destination t = cone t).  One kind of tangency is decided, not
// ambiguous: cone t's apex EXACTLY on the RIGHT arm of an earlier cone j -- fl(alpha |t - q_j| + h_j) == h_t, the
// very expression of the reference's test `dist + hj <= hk` (typeStereoLinear.h:432-435), which therefore holds and
// drops cone t whenever t meets j, while `dist + hk < hj` (:417) cannot: the trace of the serial construction is that
// of the same input with h_t raised by an infinitesimal amount, where cone t is strictly dominated by j and
// contributes nothing.  Whether the two cones give the same BITS where they coincide is the margin test's business
// (equal costs count once; costs one ulp apart fail it).  The mirror image -- an apex exactly on a LEFT arm -- is the
// reference's "s <= q_j" quirk (:444-449: the new cone is dropped although it wins to its right) and stays a failure,
// like every inexact near-tangency.  The masked columns of an NCC volume (dispmap_ncc.m:190-191) produce such exact
// right-arm ramps by the thousand: messages there are truncated cones, so the next node's H carries arms of slope
// exactly alpha (84 % of the Teddy volume's failed certificates, none of them with a result other than min-plus).
// Table: entry i at hq[stride * i] (h) and hq[stride * i + 1] (q): the first `count` entries (compacted table), or
// the entries named by `mask` (count == 0).  Returns "still tangent" per lane.
__device__ __forceinline__ bool harmless_ties_only(double alpha, double h, double t, double delta, bool useful, const double *hq,
                                                   int stride, int count, unsigned long long mask) {
  int cnt = 0;
  if (count > 0) {
    for (int i = 0; i < count; ++i) {
      const double hj = hq[stride * i], qj = hq[stride * i + 1];
      const double dc = (alpha * fabs(t - qj) + hj) - h;
      cnt += ((fabs(dc) <= delta) & !((dc == 0) & (qj < t))) ? 1 : 0;
    }
  } else {
    while (mask) {
      const int j = __builtin_ctzll(mask);
      mask &= mask - 1;
      const double hj = hq[stride * j], qj = hq[stride * j + 1];
      const double dc = (alpha * fabs(t - qj) + hj) - h;
      cnt += ((fabs(dc) <= delta) & !((dc == 0) & (qj < t))) ? 1 : 0;
    }
  }
  return cnt != (useful ? 1 : 0);
}

// Message update with everything in registers (K <= 64): h = gamma*Di - old message,
// qsrc / t = source / destination positions, perm = ascending order of the sources
// (only touched by the serial fallback).  Returns the normalised message in `out`.
// SHAREDPOS: source and destination positions are the same shared vector (lane k: qsrc == t ==
// position k) and `hq` is given; where it is strictly ascending (p.pos_gap > 0) the certified path
// walks a COMPACTED table of the useful sources (below).
template <int KERNEL, bool SHAREDPOS = false>
__device__ __forceinline__ double message_regs(const DevParams &p, int K, double alpha, double h,
                                               double qsrc, double t, const uint16_t *perm,
                                               double &outmsg, int lane, double *hq = nullptr,
                                               int window = -1, int *look_streak = nullptr,
                                               unsigned long long *vprof = nullptr, int perm_here = -1,
                                               const CoopPart coop = CoopPart()) {
  const double inf = __builtin_huge_val();
  const bool act = lane < K;
#ifdef STEREO_HIP_VISIT_PROFILE
  // phases of a message for the visit profile (cheap stamps, no atomics): [0] reduction + table,
  // [1] pair loop / flat path, [2] margins + second look, [3] serial construction + walk, [4] minimum
  long long vm_ = (long long)__builtin_readcyclecounter();
#define VMSTAMP(i) do { if (vprof) { const long long n_ = (long long)__builtin_readcyclecounter(); vprof[i] += (unsigned long long)(n_ - vm_); vm_ = n_; } } while (0)
#else
#define VMSTAMP(i) do { } while (0)
#endif
  // hmin and the magnitude behind delta in one interleaved reduction
  const double aq = alpha * qsrc;
  double hmin = h, mag = act ? fabs(h) + fabs(aq) + alpha * fabs(t) : 0.0;  // inactive lanes hold h = +inf
  wave_min_max_dpp(hmin, mag);
  double out, vmin;
  if (coop.part() && (UNI(alpha == 0) || KERNEL != 1 || !p.certificate)) { outmsg = 0; return 0; }   // (nothing there is shared)
  if (UNI(alpha == 0)) {
    out = hmin; vmin = hmin;  // typeStereoLinear.h:390-396
  } else {
    const double vtrunc = hmin + alpha * p.lambda;
    bool need_serial = true;
    out = vtrunc;
// (per-phase cycle counters of a message: compiled in only with -DSTEREO_HIP_MESSAGE_PROFILE -- even
// the untaken branches cost 5 % of a Teddy iteration: 15.4 vs 14.55 ms)
#ifndef STEREO_HIP_MESSAGE_PROFILE
#define MSTAMP(slot) do { } while (0)
#else
    long long tm0 = p.prof ? (long long)__builtin_readcyclecounter() : 0;  // development profile: slots 8..15
#define MSTAMP(slot) do { if (p.prof) { const long long n_ = (long long)__builtin_readcyclecounter(); if (lane == 0) { atomicAdd(p.prof + (slot), (unsigned long long)(n_ - tm0)); atomicAdd(p.prof + (slot) + 1, 1ull); } tm0 = n_; } } while (0)
#endif
    // (Going straight to the serial construction after four failures in a row, with a look at the certificate at
    //  every eighth message, was measured in round 4: 23-41 % more serial constructions -- the failures of a wave do
    //  not come in runs long enough -- and 14.21 -> 14.31 ms per iteration on the Teddy pair.  Not kept.)
    if (KERNEL == 1 && p.certificate) {
      // Fast path (DESIGN.md "message certificate").  Only "useful" sources, those with
      // h < vTrunc, can produce a value below the truncation level: cost >= h for every
      // other source.  Min-plus over the useful sources is therefore the plain min-plus
      // result; the certificate demands (i) every pair of cones with distinct apex positions
      // of which at least one is useful is delta-separated from tangency (|u_i-u_j| > delta
      // and |v_i-v_j| > delta with u = h - alpha q, v = h + alpha q), so each comparison
      // the reference's serial envelope construction makes on a useful cone is decided as
      // in real arithmetic, and (ii) at every destination whose minimum beats vTrunc the
      // minimum is delta-separated from the next larger cost and from vTrunc, so rounding
      // in the envelope's breakpoints cannot pick another value.  Otherwise: serial path.
      const double ui = h - aq, vi = h + aq;
      const double delta = 1e-9 * (mag + fabs(alpha * p.lambda));
      const bool useful = act && h < vtrunc;
      unsigned long long mask = __builtin_amdgcn_ballot_w64(useful);
      double m1 = inf, m2 = inf;
      bool bad = !(delta < inf);
      if (SHAREDPOS && p.pos_gap > 0 && !(window >= 0 && __builtin_popcountll(mask) > kFlatFrom)) {
        // Shared strictly ascending positions, at most 32 useful sources (or no window): the useful
        // sources' (h, q) are COMPACTED into the wave's table -- a useful lane writes entry number
        // "useful lanes below me" -- and walked four per trip with uniform reads at constant offsets:
        // no mask arithmetic, no per-source address, nothing in scalar registers inside the loop (a
        // v_cmp whose mask an s_and / s_or consumes stalls the wave for the VALU's latency; the trip of
        // the masked loop below is 20 scalar + 30 vector instructions for two sources, this one ~13
        // vector instructions per source).  The table ends with seven inert entries (h = +inf: cost
        // +inf, changes neither minimum, matches nothing), so a trip never has to be cut short.
        // Tangency as in the wide kernel (4.4): with destination t = position of lane t, the cost
        // c_j(t) = alpha |t - q_j| + h_j of useful source j equals u_j - u_t + h_t right of j and
        // v_j - v_t + h_t left of it, so |c_j(t) - h_t| <= delta IS "cone t lies within delta of an arm
        // of cone j"; lane t counts its matches, and a useful cone matches itself exactly once
        // (distance 0: c = h_t bit for bit), so "count != [t useful]" means a near tangency.  The pair
        // test the masked loop makes in addition -- the apex of a USEFUL cone j on an arm of a useless
        // cone t -- needs h_j >= h_t + alpha |q_t - q_j| - delta >= vTrunc + alpha gap - delta, which
        // h_j < vTrunc rules out as soon as alpha gap > 2 delta: checked (uniform), serial path otherwise.
        const int nuse = __builtin_popcountll(mask);
        bad = bad || !(alpha * p.pos_gap > 2 * delta);
        {
          const int slot = __builtin_amdgcn_mbcnt_hi((unsigned)(mask >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)mask, 0));
          if (useful) { hq[2 * slot] = h; hq[2 * slot + 1] = qsrc; }
          if (lane < 7) { hq[2 * (nuse + lane)] = inf; hq[2 * (nuse + lane) + 1] = 0; }
          __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
          __builtin_amdgcn_wave_barrier();
        }
        VMSTAMP(0);
        int cnt = 0;
        // (the match count as compare + add-with-carry; left to itself the compiler packs the four
        //  compare results of a trip into a bit field and counts its bits: 14 instructions instead of 8.
        //  s_nop 1: an fp64 v_cmp's VCC needs two wait states before a VALU reads it as a carry)
#define STEREO_ACC_C(HJ, QJ)                                                         \
  {                                                                                  \
    const double c = pair_cost<1>(alpha, t - (QJ), (HJ));                            \
    const double lo = min_raw(m1, c), hi = max_raw(m1, c);                           \
    const double dc = c - h;                                                         \
    m2 = min_raw_if(hi > lo, m2, hi);                                                \
    m1 = lo;                                                                         \
    asm("v_cmp_le_f64_e64 vcc, |%1|, %2\n\ts_nop 1\n\tv_addc_co_u32_e32 %0, vcc, 0, %0, vcc" \
        : "+v"(cnt) : "v"(dc), "v"(delta) : "vcc");                                  \
  }
        // (two waves: only worth the exchange with more than eight sources; the helper takes the upper half)
        const bool split = coop.active() && nuse > 8;
        if (nuse <= 8) {
          if (coop.part()) { outmsg = 0; return 0; }   // (nothing to share: the first wave does it alone)
          // (up to eight sources -- nearly every message of a noisy volume --: all eight entries are
          //  requested together, one LDS latency instead of one per trip; first in the chain of cases: the common one)
          const double h0 = hq[0], q0 = hq[1], h1 = hq[2], q1 = hq[3], h2 = hq[4], q2 = hq[5], h3 = hq[6], q3 = hq[7];
          const double h4 = hq[8], q4 = hq[9], h5 = hq[10], q5 = hq[11], h6 = hq[12], q6 = hq[13], h7 = hq[14], q7 = hq[15];
          STEREO_ACC_C(h0, q0) STEREO_ACC_C(h1, q1) STEREO_ACC_C(h2, q2) STEREO_ACC_C(h3, q3)
          if (nuse > 4) {
            STEREO_ACC_C(h4, q4) STEREO_ACC_C(h5, q5)
            if (nuse > 6) { STEREO_ACC_C(h6, q6) STEREO_ACC_C(h7, q7) }
          }
        } else
        if (split) {
          const int half = ((nuse + 7) >> 3) << 2;   // a multiple of four >= nuse / 2
          const int i0 = coop.part() ? half : 0, i1 = coop.part() ? nuse : half;
          for (int i = i0; i < i1; i += 4) {
            const double h0 = hq[2 * i], q0 = hq[2 * i + 1], h1 = hq[2 * i + 2], q1 = hq[2 * i + 3];
            const double h2 = hq[2 * i + 4], q2 = hq[2 * i + 5], h3 = hq[2 * i + 6], q3 = hq[2 * i + 7];
            STEREO_ACC_C(h0, q0) STEREO_ACC_C(h1, q1) STEREO_ACC_C(h2, q2) STEREO_ACC_C(h3, q3)
          }
          if (coop.part()) { coop_publish(coop, m1, m2, cnt, lane); outmsg = 0; return 0; }
          coop_collect(coop, m1, m2, cnt, lane);
        } else if (coop.part()) {
          outmsg = 0; return 0;   // (nothing to share: the first wave does it alone)
        } else {
          for (int i = 0; i < nuse; i += 4) {
            const double h0 = hq[2 * i], q0 = hq[2 * i + 1], h1 = hq[2 * i + 2], q1 = hq[2 * i + 3];
            const double h2 = hq[2 * i + 4], q2 = hq[2 * i + 5], h3 = hq[2 * i + 6], q3 = hq[2 * i + 7];
            STEREO_ACC_C(h0, q0) STEREO_ACC_C(h1, q1) STEREO_ACC_C(h2, q2) STEREO_ACC_C(h3, q3)
          }
        }
#undef STEREO_ACC_C
        bool tangent = cnt != (useful ? 1 : 0);
        if (UNI(act & tangent) && !UNI(act & bad)) tangent = harmless_ties_only(alpha, h, t, delta, useful, hq, 2, nuse, 0ull);
        bad = bad | tangent;
        VMSTAMP(1);
      } else {
      // The sources are broadcast from a per-wave LDS table (one ds_read_b128 per source instead of
      // eight v_readlane); two sources per trip keep two independent dependency chains in flight.
      // m1 / m2 = smallest and second smallest DISTINCT cost seen so far.
      // table entry of a source: (h, q, u, v) -- the tangency test then needs no arithmetic on the source
      const bool flat = window >= 0 && __builtin_popcountll(mask) > kFlatFrom;
      if (coop.part() && !flat) { outmsg = 0; return 0; }   // (the masked loop is not shared: the first wave does it alone)
      if (hq) {
        hq[4 * lane] = h; hq[4 * lane + 1] = qsrc;
        if (!flat) { hq[4 * lane + 2] = ui; hq[4 * lane + 3] = vi; }   // (the flat path reads (h, q) only)
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
      }
      VMSTAMP(0);
#define STEREO_SRC(J, HJ, QJ)                                                        \
  double HJ, QJ, HJ##u, HJ##v;                                                       \
  if (hq) { HJ = hq[4 * (J)]; QJ = hq[4 * (J) + 1]; HJ##u = hq[4 * (J) + 2]; HJ##v = hq[4 * (J) + 3]; } \
  else {                                                                             \
    HJ = readlane_f64(h, (J)); QJ = readlane_f64(qsrc, (J));                         \
    const double aqj_ = alpha * QJ;                                                  \
    HJ##u = HJ - aqj_; HJ##v = HJ + aqj_;                                            \
  }
#define STEREO_ACC(HJ, QJ)                                                           \
  {                                                                                  \
    const double c = pair_cost<1>(alpha, t - QJ, HJ);                                \
    const double lo = min_raw(m1, c), hi = max_raw(m1, c);                           \
    m2 = min_raw_if(hi > lo, m2, hi);                                                \
    m1 = lo;                                                                         \
    const bool near = (fabs(ui - HJ##u) <= delta) || (fabs(vi - HJ##v) <= delta);    \
    bad = bad || (near && qsrc != QJ);                                               \
  }
#ifdef STEREO_HIP_MESSAGE_PROFILE
      if (p.prof && lane == 0) { const int nu_ = __builtin_popcountll(mask); atomicAdd(p.prof + 56 + (nu_ > 32 ? 3 : nu_ > 16 ? 2 : nu_ > 8 ? 1 : 0), 1ull); }
#endif
      if (flat) {
        // Flat h (the zig-zag rows: gamma = 1/6 .. 1/8 makes almost every source useful) on shared
        // strictly ascending positions.  The pair loop below would be K^2; instead
        //  * tangency for ALL pairs by sorting u and v (conservative superset of the useful pairs):
        //    keys quantised to 32 bits over a range that certainly contains them, "within delta"
        //    tested as "within delta / resolution + 2 units";
        //  * min-plus only over the sources inside the truncation window of each destination (a source
        //    farther than lambda costs >= vTrunc exactly); the table is padded with +inf entries.
        // (two waves: the helper walks the window while the first wave sorts the keys)
        // (... all of it but the last kFlatOwn entries, which the first wave takes behind its sort)
        constexpr int kFlatOwn = STEREO_FLAT_OWN;
        const int wsplit = window >= kFlatOwn ? window - kFlatOwn : window;
        if (coop.part()) {
          window_minplus(hq, alpha, t, lane, -window, wsplit, m1, m2);
          coop_publish(coop, m1, m2, 0, lane);
          outmsg = 0;
          return 0;
        }
        // (keys of cones that cannot meet a useful one are taken at a CAP: a cone within delta of a useful cone's arm has
        //  h <= vTrunc + |alpha| (position range) + delta, so whatever lies above the cap is out of every tangency a useful
        //  cone takes part in; capped, the keys' range is known without a reduction for max H -- this wave is the one
        //  the visit waits for -- and a capped key that happens to land next to another one only costs the second look,
        //  which judges pairs by their true h.  The scale by v_rcp_f64 and a safety factor: any scale that keeps the keys
        //  below 2^32 will do, the threshold is made with the same one.)
        const double ap0 = alpha * p.pos_first, ap1 = alpha * p.pos_last;
        const double aplo = min_raw(ap0, ap1), aphi = max_raw(ap0, ap1);
        const double hcap = vtrunc + 1.000001 * (aphi - aplo) + 4 * delta;
        const double span = (hcap - hmin) + (aphi - aplo);  // >= max u - min u and >= max v - min v of the capped cones
        const double scale = (4294967040.0 * 0.99999) * __builtin_amdgcn_rcp(span);   // <= (2^32 - 256) / span
        bad = bad || !(span < inf) || !(span > 0) || !(delta * scale < 1e9);
        unsigned ku = 0xFFFFFFFFu, kv = 0xFFFFFFFFu;
        if (act && !bad) {
          const double hc = min_raw(h, hcap);
          ku = (unsigned)(((hc - aq) - (hmin - aphi)) * scale);
          kv = (unsigned)(((hc + aq) - (hmin + aplo)) * scale);
        }
        wave_sort2(ku, kv, lane);
        // (the key of the next lane: wave_shl:1, one DPP move each)
        const unsigned un = (unsigned)__builtin_amdgcn_mov_dpp((int)ku, 0x130, 0xF, 0xF, true),
                       vn = (unsigned)__builtin_amdgcn_mov_dpp((int)kv, 0x130, 0xF, 0xF, true);
        const unsigned thr = bad ? 0u : (unsigned)(delta * scale) + 2u;
        bool tangent = lane + 1 < K && (un - ku <= thr || vn - kv <= thr);
        // (the sorted keys see ALL pairs and cannot tell an exact tie from a near one: a hit is looked at again pair by pair
        //  -- the pairs with a USEFUL member, which is all the certificate asks for; the one pair kind that loop cannot see,
        //  a useful apex on an arm of a useless cone, is excluded by alpha gap > 2 delta as in the compacted loop above)
        if (UNI(act & tangent) && !UNI(act & bad) && alpha * p.pos_gap > 2 * delta)
          tangent = harmless_ties_only(alpha, h, t, delta, useful, hq, 4, 0, mask);
        bad = bad || tangent;
        if (coop.active()) {
          int none = 0;
          window_minplus(hq, alpha, t, lane, wsplit + 1, window, m1, m2);
          coop_collect(coop, m1, m2, none, lane);
        } else {
          window_minplus(hq, alpha, t, lane, -window, window, m1, m2);
        }
      } else {
      // (a wave whose last certificate failed checks after every trip whether this one has failed
      //  already: in the flat columns of an NCC volume the failures come in runs)
      //  -- its own copy of the loop, so that the check costs the other waves nothing)
#define STEREO_TRIP                                                                   \
        const int j0 = __builtin_ctzll(mask);                                          \
        mask &= mask - 1;                                                              \
        const int j1 = mask ? __builtin_ctzll(mask) : j0; /* a source visited twice changes nothing */ \
        mask &= mask - 1;                                                              \
        STEREO_SRC(j0, hj0, qj0)                                                       \
        STEREO_SRC(j1, hj1, qj1)                                                       \
        STEREO_ACC(hj0, qj0)                                                           \
        STEREO_ACC(hj1, qj1)
      if (look_streak && *look_streak > 0) {
        while (mask) {
          if (UNI(act && bad)) break;
          STEREO_TRIP
        }
      } else {
        while (mask) { STEREO_TRIP }
      }
#undef STEREO_TRIP
      }
#undef STEREO_SRC
#undef STEREO_ACC
      VMSTAMP(1);
      }
      // (| and &, not || and &&: short-circuit conditions on per-lane fp64 tests become nested exec-mask regions)
      bad = bad | ((m1 < vtrunc) & !((m2 - m1 > delta) & (vtrunc - m1 > delta)));
      need_serial = UNI(act & bad);
      MSTAMP(8);
      // The second look (8.6 k cycles) rescues messages whose useful cones are merely close; on volumes
      // with exact ties (the flat columns of an NCC volume: 442 k failed certificates per 10 iterations
      // of the Teddy-sized workload, not one rescued) it only delays the serial construction.  Either
      // way the result is the reference's, so whether to look is the caller's running bet: after four
      // failures in a row only every 32nd failed certificate is looked at again.
      if (need_serial) {
        const int streak = look_streak ? *look_streak : 0;
        if (streak < 4 || (streak & 31) == 0) {
          double delta2;
          bool rel;
          if (second_look_applies(p.lambda, K, alpha, h, qsrc, t, vtrunc, delta, lane, delta2, rel))
            need_serial = message_second_look(K, alpha, h, qsrc, t, vtrunc, delta2, rel, lane, m1);
          MSTAMP(10);
          if (look_streak) *look_streak = need_serial ? streak + 1 : 0;
        } else {
          *look_streak = streak + 1;
        }
      } else if (look_streak) {
        *look_streak = 0;
      }
      out = min_raw(m1, vtrunc);   // m1 < vtrunc ? m1 : vtrunc (no NaN on either side)
      if (need_serial && lane == 0 && p.fallbacks) atomicAdd(p.fallbacks, 1);
    }
    if (KERNEL == 2 && p.certificate) {
      if (hq) {
        hq[4 * lane] = h; hq[4 * lane + 1] = qsrc;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
      }
      double m1;
      need_serial = message_quad_fast(p.lambda, K, alpha, h, qsrc, t, vtrunc, lane, hq, m1, hq ? window : -1, p.pos_gap);
      out = m1 < vtrunc ? m1 : vtrunc;
      if (need_serial && lane == 0 && p.fallbacks) atomicAdd(p.fallbacks, 1);
    }
    VMSTAMP(2);
    if (need_serial) {
      // (perm_here >= 0: the caller holds perm[lane] in a register -- shared positions, the same for every
      //  message: no load on the serial path)
      const int idx = perm_here >= 0 ? perm_here : act ? perm[lane] : lane;
      const double hs = __shfl(h, idx, kWave), qs = __shfl(qsrc, idx, kWave);
      double sh = 0, sq = 0, zz = 0;
      int maxtop = 0;
      bool built = false;
      if (KERNEL == 1 && !(p.debug & 512)) {
        // (hq + 4 (64 + pad): the scratch words behind this wave's table, see kPipeTab)
        if (hq && !(p.debug & 2048)) built = build_envelope_parallel(K, alpha, hs, qs, hq, (int *)(hq + 4 * (kWave + 16)), sh, sq, zz, lane, maxtop, mag, p.prof);
        if (!built) built = build_envelope_masks(K, alpha, hs, qs, sh, sq, zz, lane, maxtop);
      }
      if (!built) maxtop = build_envelope_regs<KERNEL>(K, alpha, hs, qs, sh, sq, zz, lane);
      MSTAMP(12);
      out = envelope_value<KERNEL>(p, alpha, t, vtrunc, sh, sq, zz, maxtop, lane, hq);
      MSTAMP(14);
    }
    VMSTAMP(3);
#undef MSTAMP
    // Smallest entry of the message.  With the same positions on both sides (lane k: source k and
    // destination k) a certified linear message has its minimum at min H exactly: destination t sees
    // its own source at distance 0 (h_t + alpha 0 = h_t), every other term is some h_s plus a
    // non-negative cost, and vTrunc = min H + alpha lambda is no smaller -- no reduction needed.
    if (KERNEL == 1 && p.certificate && !need_serial && (SHAREDPOS || !UNI(act && qsrc != t))) vmin = hmin;
    else vmin = wave_min_dpp(act ? out : inf);
  }
  VMSTAMP(4);
#undef VMSTAMP
  outmsg = out - vmin;
  return vmin;
}

#undef RLI

__device__ __forceinline__ int group_strip(const GroupArgs &ga) {
  int s = 0;
#pragma unroll
  for (int i = 1; i < kMaxGroup; ++i) s = (i < ga.n && (int)blockIdx.x >= ga.first[i]) ? i : s;  // static indices only
  return s;
}

}  // namespace
}  // namespace stereo
