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
// completion flags between runs.  Five implementations, identical results
// (stereo_trws_plan_path): trws_pipe_kernel (K <= 64, role-specialised waves, both smoothness
// kernels), trws_pipe2_kernel (64 < K <= 128, two labels per lane, linear kernel),
// trws_wide_kernel (64 < K <= 256, shared ascending positions, linear kernel),
// trws_persistent_kernel (everything else), trws_sweep_kernel (one launch per DAG level; kept
// for comparison).  The three descriptor-driven kernels walk the chain schedule of
// trws_graph.h; messages take a certified min-plus fast path and fall back to
// the reference's serial envelope construction when the certificate fails.
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

namespace stereo {

std::string &last_error() {
  static thread_local std::string s;
  return s;
}

namespace {

constexpr int kWave = 64;
constexpr int kBlock = 256;
constexpr int kWavesPerBlock = kBlock / kWave;
constexpr int kWaveVecs = 5;  // K-vectors of LDS scratch per wave

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
  int32_t *abort_flag;
  int N;
  unsigned long long *fallbacks;  // messages that needed the serial envelope (diagnostics)
  int certificate;                // 0: always run the serial envelope
  unsigned long long *prof;       // optional: 8 phase-cycle accumulators (development)
  const int32_t *desc[2];         // packed node descriptors of the pipelined kernels
  int prof_run;
  int debug;  // development switches: 2 / 4 profile backward / forward sweeps only, 256 no windowed paths,
              // 512 serial envelopes by the lane-read loop instead of the mask construction
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
  double *peer_msg0, *peer_msg1;      // (scalars, not arrays: an index computed at run time would put
  int32_t *peer_done0, *peer_done1;   //  the whole parameter block into scratch memory)
  int32_t *peer_x0, *peer_x1;
};

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
// wave-uniform predicate -> scalar branch
#define UNI(c) (__builtin_amdgcn_ballot_w64(c) != 0)

// ---- DPP wave reductions (gfx9 row_bcast forms): ~20 VALU instead of 12 ds_bpermute.
// The combined value ends up in lane 63 and is broadcast with v_readlane.
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ double dpp_f64(double v) {
  int lo = __double2loint(v), hi = __double2hiint(v);
  lo = __builtin_amdgcn_update_dpp(lo, lo, CTRL, ROW_MASK, 0xF, false);
  hi = __builtin_amdgcn_update_dpp(hi, hi, CTRL, ROW_MASK, 0xF, false);
  return __hiloint2double(hi, lo);
}
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ int dpp_i32(int v) {
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

__device__ __forceinline__ bool build_envelope_masks(int K, double alpha, double hs, double qs, double &sh,
                                                     double &sq, double &zz, int lane, int &maxtop_out) {
  const double inf = __builtin_huge_val();
  const bool act = lane < K;
  const double c = 2 * alpha;
  bool ok = alpha > 0 && c < inf && (!act || (fabs(hs) < inf && fabs(qs) < inf));
  // thresholds on the numerator (see above); at most kSteps ulp steps from fl(q c), else give up
  constexpr int kSteps = 6;
  double thi = qs * c, tlo = thi;
  ok = ok && (!act || fabs(thi) < 1e300);
  if (!UNI(!ok)) {
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
  if (UNI(!ok)) return false;
  unsigned long long A = 1;      // source 0 is the bottom of the stack
  int top = 0, maxtop = 0;
  int src = 0, zk = -1, zj = 0;  // lane t: slot t holds source `src`; z[t+1] = crossing of (zk, zj), inf if zk < 0
  for (int k = 1; k < K; ++k) {
    const double hk = readlane_f64(hs, k), qk = readlane_f64(qs, k), thik = readlane_f64(thi, k);
    const double dist = alpha * fabs(qk - qs);
    const unsigned long long m1 = __builtin_amdgcn_ballot_w64(dist + hk < hs);
    const unsigned long long m2 = __builtin_amdgcn_ballot_w64(dist + hs <= hk);
    const double num = (hk - hs) + alpha * (qk + qs);
    const unsigned long long m3 = __builtin_amdgcn_ballot_w64(num >= thik || num <= tlo);
    const unsigned long long B = A & ~m1;
    if (B == 0) {  // every cone on the stack is dominated: k becomes the bottom (typeStereoLinear.h:419-425)
      A = 1ull << k;
      if (lane == 0) { src = k; zk = -1; }
      top = 0;
      continue;
    }
    const int js = 63 - __builtin_clzll(B);   // the cone k meets: the highest one it does not dominate
    A &= (2ull << js) - 1;                    // (js < k <= 63)
    top = __builtin_popcountll(A) - 1;
    if (((m2 | m3) >> js) & 1) continue;
    if (lane == top) { zk = k; zj = js; }
    ++top;
    if (lane == top) { src = k; zk = -1; }
    A |= 1ull << k;
    maxtop = top > maxtop ? top : maxtop;
  }
  sh = __shfl(hs, src, kWave); sq = __shfl(qs, src, kWave);
  const double hk = __shfl(hs, zk < 0 ? 0 : zk, kWave), qk = __shfl(qs, zk < 0 ? 0 : zk, kWave);
  const double hj = __shfl(hs, zj, kWave), qj = __shfl(qs, zj, kWave);
  const double s = ((hk - hj) + alpha * (qk + qj)) / (2 * alpha);
  zz = zk < 0 ? inf : s;
  maxtop_out = maxtop;
  return true;
}

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
    gap = lane != j ? min_raw(gap, dq) : gap;
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

// One message update by one wave (typeStereo*.h UpdateMessage).  Di lives in
// LDS.  Returns vMin (identical in all lanes).  SC1: the new message is stored
// write-through at agent scope (it is consumed by another workgroup inside the
// same launch); `handoff` (LDS, may be null) additionally receives it for the
// next node of the same run.
template <int KERNEL, bool BACKWARD, int MODE, bool SC1>
__device__ double update_message(const DevParams &p, int e, const double *Di, double gamma,
                                 double *scratch, double *handoff, int lane) {
  const int K = p.K, Kp = p.Kp;
  const double inf = __builtin_huge_val();
  double *m = p.msg + (size_t)e * K;
  const double alpha = p.alpha[e];
  const int mdir = p.mdir[e];
  const int dir = BACKWARD ? 1 : 0;
  // typeStereoLinear.h:343-357: dir == m_dir -> sources sit in the qprim half
  const bool src_is_qprim = (dir == mdir);
  const double *src, *dst;
  const uint16_t *perm;
  if (p.pos) {
    src = dst = p.pos; perm = p.perm_pos;
  } else {
    const size_t off = (size_t)e * K;
    src = (src_is_qprim ? p.qprim : p.q) + off;
    dst = (src_is_qprim ? p.q : p.qprim) + off;
    perm = (src_is_qprim ? p.perm_qp : p.perm_q) + off;
  }
  if (MODE == STEREO_TRWS_MESSAGES_EXACT && K <= kWave) {
    // ---- register path, lane = label
    double h = inf, qsrc = 0, t = 0;
    if (lane < K) {
      h = gamma * Di[lane] - m[lane];
      qsrc = src[lane];
      t = dst[lane];
    }
    const double hmin = wave_min(h);
    double out, vmin;
    if (UNI(alpha == 0)) {
      out = hmin; vmin = hmin;  // typeStereoLinear.h:390-396
    } else {
      const double vtrunc = hmin + alpha * p.lambda;
      bool need_serial = true;
      out = vtrunc;
      if (KERNEL == 1 && p.certificate) {
        // Fast path: plain min-plus over all sources plus a certificate that the
        // reference's serial envelope construction yields the very same bits
        // (DESIGN.md "message certificate"): (i) no cone apex lies within delta of
        // another cone (u = h - alpha q and v = h + alpha q pairwise delta-separated
        // for distinct positions), so every comparison the serial algorithm makes
        // is decided as in real arithmetic and it builds the true lower envelope;
        // (ii) the minimum over the cones is delta-separated from the next larger
        // cost at every destination whose minimum beats the truncation value, so
        // rounding in the envelope's breakpoints cannot select a different value.
        const double aq = alpha * qsrc;
        const double ui = h - aq, vi = h + aq;
        double mag = lane < K ? fabs(h) + fabs(aq) + alpha * fabs(t) : 0.0;
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {
          const double o = __shfl_xor(mag, off, kWave);
          mag = o > mag ? o : mag;
        }
        const double delta = 1e-9 * (mag + fabs(alpha * p.lambda));
        double m1 = inf, m2 = inf;
        bool bad = !(delta < inf);
        for (int j = 0; j < K; ++j) {
          const double hj = readlane_f64(h, j), qj = readlane_f64(qsrc, j);
          const double c = pair_cost<1>(alpha, t - qj, hj);
          if (c < m1) { m2 = m1; m1 = c; } else if (c > m1 && c < m2) { m2 = c; }
          const double aqj = alpha * qj;
          const double uj = hj - aqj, vj = hj + aqj;
          const bool near = (fabs(ui - uj) <= delta) || (fabs(vi - vj) <= delta);
          bad = bad || (near && qsrc != qj);
        }
        bad = bad || (m1 < vtrunc && !(m2 - m1 > delta));
        need_serial = UNI(lane < K && bad);
        out = m1 < vtrunc ? m1 : vtrunc;
        if (need_serial && lane == 0 && p.fallbacks) atomicAdd(p.fallbacks, 1);
      }
      if (KERNEL == 2 && p.certificate) {
        double m1;
        need_serial = message_quad_fast(p.lambda, K, alpha, h, qsrc, t, vtrunc, lane, nullptr, m1);
        out = m1 < vtrunc ? m1 : vtrunc;
        if (need_serial && lane == 0 && p.fallbacks) atomicAdd(p.fallbacks, 1);
      }
      if (need_serial) {
        // the reference's serial envelope, lane k = k-th source in ascending position order
        const int idx = lane < K ? perm[lane] : lane;
        const double hs = __shfl(h, idx, kWave), qs = __shfl(qsrc, idx, kWave);
        double sh, sq, zz;
        const int maxtop = build_envelope_regs<KERNEL>(K, alpha, hs, qs, sh, sq, zz, lane);
        // while (z[j+1] < t) ++j, walked over the slots with uniform reads
        double ch = 0, cq = 0;
        bool walking = true;
        for (int j = 0; j <= maxtop; ++j) {
          const double shj = readlane_f64(sh, j), sqj = readlane_f64(sq, j), zj1 = readlane_f64(zz, j);
          if (walking) { ch = shj; cq = sqj; walking = zj1 < t; }
        }
        const double c = pair_cost<KERNEL>(alpha, t - cq, ch);
        out = c < vtrunc ? c : vtrunc;
      }
      vmin = wave_min(lane < K ? out : inf);
    }
    if (lane < K) {
      const double v = out - vmin;
      if (SC1) st_sc1(m + lane, v); else m[lane] = v;
      if (handoff) handoff[lane] = v;
    }
    return vmin;
  }
  // ---- LDS path (K > 64, or plain min-plus)
  double *A = scratch;           // exact: Hs   | minplus: H
  double *B = scratch + Kp;      // exact: Qs   | minplus: S
  double *sh = scratch + 2 * Kp;
  double *sq = scratch + 3 * Kp;
  double *z = scratch + 4 * Kp;  // Kp + 2 entries (allocation has slack)
  double hmin = inf;
  if (MODE == STEREO_TRWS_MESSAGES_EXACT) {
    for (int k = lane; k < K; k += kWave) {
      const int idx = perm[k];
      const double h = gamma * Di[idx] - m[idx];
      A[k] = h; B[k] = src[idx];
      hmin = h < hmin ? h : hmin;
    }
  } else {
    for (int k = lane; k < K; k += kWave) {
      const double h = gamma * Di[k] - m[k];
      A[k] = h; B[k] = src[k];
      hmin = h < hmin ? h : hmin;
    }
  }
  hmin = wave_min(hmin);
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  double vmin = inf;
  double outv[8];  // K <= 8*64
  if (alpha == 0) {
    // typeStereoLinear.h:390-396
#pragma unroll
    for (int it = 0; it < 8; ++it) outv[it] = hmin;
    vmin = hmin;
  } else {
    const double vtrunc = hmin + alpha * p.lambda;
    bool certified = false;
    if (MODE == STEREO_TRWS_MESSAGES_EXACT && KERNEL == 1 && p.certificate && K <= 2 * kWave) {
      // Certified fast path for 64 < K <= 128 (two labels per lane), as in the register path:
      // min-plus over the useful sources (h < vTrunc) and the tangency / margin certificate
      // (DESIGN.md "message certificate"); the serial construction below only runs if it fails.
      // A / B hold the sources in ascending position order; a lane owns cones and destinations
      // lane and lane + 64.  The list of useful sources is compacted into `sh` (free until then).
      int *ul = (int *)sh;
      int nu = 0;
      double ck_h[2], ck_q[2], ck_u[2], ck_v[2], tt[2], m1[2] = {inf, inf}, m2[2] = {inf, inf};
      double mag = 0;
#pragma unroll
      for (int it = 0; it < 2; ++it) {
        const int k = lane + it * kWave;
        const bool on = k < K;
        ck_h[it] = on ? A[k] : inf; ck_q[it] = on ? B[k] : 0.0; tt[it] = on ? dst[k] : 0.0;
        const double aq = alpha * ck_q[it];
        ck_u[it] = ck_h[it] - aq; ck_v[it] = ck_h[it] + aq;
        if (on) {
          const double mg = fabs(ck_h[it]) + fabs(aq) + alpha * fabs(tt[it]);
          mag = mg > mag ? mg : mag;
        }
        const bool useful = on && ck_h[it] < vtrunc;
        const unsigned long long um = __builtin_amdgcn_ballot_w64(useful);
        if (useful) ul[nu + __builtin_popcountll(um & ((1ull << lane) - 1))] = k;
        nu += __builtin_popcountll(um);
      }
#pragma unroll
      for (int off = 32; off > 0; off >>= 1) {
        const double o = __shfl_xor(mag, off, kWave);
        mag = o > mag ? o : mag;
      }
      double delta = 1e-9 * (mag + fabs(alpha * p.lambda));
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      bool bad = false;
      bool rel[2] = {true, true};  // cones that take part in the tangency test
      for (int attempt = 0;; ++attempt) {
        bad = !(delta < inf);
        vmin = inf;
#pragma unroll
        for (int it = 0; it < 2; ++it) { m1[it] = inf; m2[it] = inf; }
        for (int jj = 0; jj < nu; ++jj) {
          const int j = ul[jj];
          const double hj = A[j], qj = B[j];
          const double aqj = alpha * qj;
          const double uj = hj - aqj, vj = hj + aqj;
#pragma unroll
          for (int it = 0; it < 2; ++it) {
            const double c = pair_cost<1>(alpha, tt[it] - qj, hj);
            const double lo = min_raw(m1[it], c), hi = max_raw(m1[it], c);
            m2[it] = hi > lo ? min_raw(m2[it], hi) : m2[it];
            m1[it] = lo;
            const bool near = (fabs(ck_u[it] - uj) <= delta) || (fabs(ck_v[it] - vj) <= delta);
            bad = bad || (near && ck_q[it] != qj && rel[it] && lane + it * kWave < K);
          }
        }
#pragma unroll
        for (int it = 0; it < 2; ++it) {
          if (lane + it * kWave < K) {
            bad = bad || (m1[it] < vtrunc && !(m2[it] - m1[it] > delta && vtrunc - m1[it] > delta));
            outv[it] = m1[it] < vtrunc ? m1[it] : vtrunc;
            vmin = outv[it] < vmin ? outv[it] : vmin;
          }
        }
        if (!UNI(bad) || attempt == 1) break;
        // Second look: a cone whose apex lies above vTrunc by more than alpha times the whole position
        // range cannot touch a useful cone (every useful cone dominates it with that margin wherever
        // they meet), so it neither counts for the magnitude behind delta nor for the tangency test.
        // Out-of-range plane proposals (unary ~ 4e7, dispmap_ncc.m:245) would otherwise inflate delta.
        double qabs = 0;
#pragma unroll
        for (int it = 0; it < 2; ++it)
          if (lane + it * kWave < K) qabs = max_raw(qabs, max_raw(fabs(ck_q[it]), fabs(tt[it])));
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) qabs = max_raw(qabs, __shfl_xor(qabs, off, kWave));
        const double hbig = vtrunc + 2.000002 * fabs(alpha) * qabs;
        double mag2 = 0;
#pragma unroll
        for (int it = 0; it < 2; ++it) {
          rel[it] = ck_h[it] <= hbig;
          if (lane + it * kWave < K)
            mag2 = max_raw(mag2, (rel[it] ? fabs(ck_h[it]) : 0.0) + fabs(alpha * ck_q[it]) + alpha * fabs(tt[it]));
        }
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) mag2 = max_raw(mag2, __shfl_xor(mag2, off, kWave));
        mag2 = max_raw(mag2, fabs(vtrunc));
        const double delta2 = 1e-9 * (mag2 + fabs(alpha * p.lambda));
        if (!(delta2 < delta)) break;
        delta = delta2;
      }
      certified = !UNI(bad);
      if (!certified) {
        vmin = inf;
        if (lane == 0 && p.fallbacks) atomicAdd(p.fallbacks, 1);
        __builtin_amdgcn_wave_barrier();  // everyone is done with the list in `sh`
      }
    }
    if (MODE == STEREO_TRWS_MESSAGES_EXACT && KERNEL == 2 && p.certificate && K <= 4 * kWave) {
      // Quadratic kernel, 64 < K <= 256 (up to four labels per lane): plain min-plus over the useful
      // sources with the destination-margin certificate of message_quad_fast.  A / B
      // hold the sources in ascending position order, so the smallest distance between two source
      // positions is the smallest gap of two neighbours (all sources: more than the proof needs).
      constexpr int NI = 4;
      int *ul = (int *)sh;
      int nu = 0;
      double tt[NI], m1[NI], m2[NI];
      double scale = 0, qlo = inf, qhi = -inf, gap = inf;
#pragma unroll
      for (int it = 0; it < NI; ++it) {
        const int k = lane + it * kWave;
        const bool on = k < K;
        const double hk = on ? A[k] : inf, qk = on ? B[k] : 0.0;
        tt[it] = on ? dst[k] : 0.0;
        m1[it] = inf; m2[it] = inf;
        if (on) {
          scale = max_raw(scale, fabs(hk) + alpha * qk * qk + alpha * tt[it] * tt[it]);
          qlo = min_raw(qlo, qk); qhi = max_raw(qhi, qk);
          if (k + 1 < K) gap = min_raw(gap, B[k + 1] - qk);
        }
        const bool useful = on && hk < vtrunc;
        const unsigned long long um = __builtin_amdgcn_ballot_w64(useful);
        if (useful) ul[nu + __builtin_popcountll(um & ((1ull << lane) - 1))] = k;
        nu += __builtin_popcountll(um);
      }
#pragma unroll
      for (int off = 32; off > 0; off >>= 1) {
        scale = max_raw(scale, __shfl_xor(scale, off, kWave));
        qlo = min_raw(qlo, __shfl_xor(qlo, off, kWave));
        qhi = max_raw(qhi, __shfl_xor(qhi, off, kWave));
        gap = min_raw(gap, __shfl_xor(gap, off, kWave));
      }
      const double delta = 1e-9 * (scale + fabs(alpha * p.lambda) + fabs(vtrunc));
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      for (int jj = 0; jj < nu; ++jj) {
        const int j = ul[jj];
        const double hj = A[j], qj = B[j];
#pragma unroll
        for (int it = 0; it < NI; ++it) {
          const double c = pair_cost<2>(alpha, tt[it] - qj, hj);
          const double lo = min_raw(m1[it], c), hi = max_raw(m1[it], c);
          m2[it] = min_raw(m2[it], hi);
          m1[it] = lo;
        }
      }
      bool bad = !(delta < inf) || !(alpha > 0) || !(gap > 4e-8) || !(1e-13 * scale * (qhi - qlo) < delta * gap);
      vmin = inf;
#pragma unroll
      for (int it = 0; it < NI; ++it) {
        if (lane + it * kWave < K) {
          bad = bad || (m1[it] < vtrunc && !(m2[it] - m1[it] > delta && vtrunc - m1[it] > delta));
          outv[it] = m1[it] < vtrunc ? m1[it] : vtrunc;
          vmin = outv[it] < vmin ? outv[it] : vmin;
        }
      }
      certified = !UNI(bad);
      if (!certified) {
        vmin = inf;
        if (lane == 0 && p.fallbacks) atomicAdd(p.fallbacks, 1);
        __builtin_amdgcn_wave_barrier();  // everyone is done with the list in `sh`
      }
    }
    if (certified) {
      // outv / vmin are set
    } else if (MODE == STEREO_TRWS_MESSAGES_EXACT) {
      if (lane == 0) build_envelope<KERNEL>(K, alpha, A, B, sh, sq, z);
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
#pragma unroll
      for (int it = 0; it < 8; ++it) {
        const int kd = lane + it * kWave;
        if (kd < K) {
          const double t = dst[kd];
          int j = 0;
          while (z[j + 1] < t) ++j;
          const double c = pair_cost<KERNEL>(alpha, t - sq[j], sh[j]);
          const double v = c < vtrunc ? c : vtrunc;
          outv[it] = v;
          vmin = v < vmin ? v : vmin;
        }
      }
    } else {
#pragma unroll
      for (int it = 0; it < 8; ++it) {
        const int kd = lane + it * kWave;
        if (kd < K) {
          const double t = dst[kd];
          double best = vtrunc;
          for (int ks = 0; ks < K; ++ks) {
            const double c = pair_cost<KERNEL>(alpha, t - B[ks], A[ks]);
            best = c < best ? c : best;
          }
          outv[it] = best;
          vmin = best < vmin ? best : vmin;
        }
      }
    }
    vmin = wave_min(vmin);
  }
#pragma unroll
  for (int it = 0; it < 8; ++it) {
    const int kd = lane + it * kWave;
    if (kd < K) {
      const double v = outv[it] - vmin;
      if (SC1) st_sc1(m + kd, v); else m[kd] = v;
      if (handoff) handoff[kd] = v;
    }
  }
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  return vmin;
}

// One DAG level of a sweep: minimize.cpp:36-62 (forward) / :67-95 (backward).
template <int KERNEL, bool BACKWARD, int MODE>
__global__ __launch_bounds__(kBlock) void trws_sweep_kernel(DevParams p, const int32_t *ranks) {
  extern __shared__ __attribute__((aligned(16))) double lds[];
  const int K = p.K, Kp = p.Kp;
  double *Di = lds;
  double *red = lds + Kp;  // kWavesPerBlock doubles
  const int tid = threadIdx.x, lane = tid & (kWave - 1), wave = tid / kWave;
  double *scratch = lds + Kp + 8 + (size_t)wave * (kWaveVecs * Kp + 8);
  const int r = ranks[blockIdx.x];
  const int node = p.order[r];
  const int f0 = p.fptr[r], f1 = p.fptr[r + 1], b0 = p.bptr[r], b1 = p.bptr[r + 1];
  // Di = D + messages, summed in the reference's list order
  double vloc = __builtin_huge_val();
  for (int k = tid; k < K; k += kBlock) {
    double acc = p.unary[(size_t)node * K + k];
    if (!BACKWARD) {
      for (int i = f0; i < f1; ++i) acc += p.msg[(size_t)p.fidx[i] * K + k];
      for (int i = b0; i < b1; ++i) acc += p.msg[(size_t)p.bidx[i] * K + k];
    } else {
      for (int i = b0; i < b1; ++i) acc += p.msg[(size_t)p.bidx[i] * K + k];
      for (int i = f0; i < f1; ++i) acc += p.msg[(size_t)p.fidx[i] * K + k];
    }
    Di[k] = acc;
    vloc = acc < vloc ? acc : vloc;
  }
  if (BACKWARD) {
    vloc = wave_min(vloc);
    if (lane == 0) red[wave] = vloc;
    __syncthreads();
    double vmin = red[0];
#pragma unroll
    for (int w = 1; w < kWavesPerBlock; ++w) vmin = red[w] < vmin ? red[w] : vmin;
    for (int k = tid; k < K; k += kBlock) Di[k] -= vmin;
    if (tid == 0) p.lbterms[p.lb_pos_node[r]] = vmin;
  }
  __syncthreads();
  const double gamma = p.gamma[r];
  const int e0 = BACKWARD ? b0 : f0, e1 = BACKWARD ? b1 : f1;
  const int32_t *elist = BACKWARD ? p.bidx : p.fidx;
  for (int i = e0 + wave; i < e1; i += kWavesPerBlock) {
    const int e = elist[i];
    const double v = update_message<KERNEL, BACKWARD, MODE, false>(p, e, Di, gamma, scratch, nullptr, lane);
    if (BACKWARD && lane == 0) p.lbterms[p.lb_pos_edge[e]] = v;
  }
}

// One DAG level of ComputeSolutionAndEnergy (minimize.cpp:223-264).
template <int KERNEL>
__global__ __launch_bounds__(kBlock) void trws_primal_kernel(DevParams p, const int32_t *ranks) {
  extern __shared__ __attribute__((aligned(16))) double lds[];
  const int K = p.K;
  double *redv = lds;
  int *redi = (int *)(lds + kWavesPerBlock);
  double *Dbs = lds + 2 * kWavesPerBlock;  // K doubles: DiBackward
  const int tid = threadIdx.x, lane = tid & (kWave - 1), wave = tid / kWave;
  const int r = ranks[blockIdx.x];
  const int node = p.order[r];
  const int f0 = p.fptr[r], f1 = p.fptr[r + 1], b0 = p.bptr[r], b1 = p.bptr[r + 1];
  double bestv = __builtin_huge_val();
  int besti = 0x7fffffff;
  for (int k = tid; k < K; k += kBlock) {
    double db = p.unary[(size_t)node * K + k];
    for (int i = b0; i < b1; ++i) {
      const int e = p.bidx[i];
      const int ks = p.x[p.tail[e]];
      const double alpha = p.alpha[e];
      double d;
      if (p.pos) {
        d = p.mdir[e] == 0 ? p.pos[ks] - p.pos[k] : p.pos[k] - p.pos[ks];
      } else {
        const size_t off = (size_t)e * K;
        // typeStereoLinear.h:505-517: AddColumn(dir = 0)
        d = p.mdir[e] == 0 ? p.qprim[off + ks] - p.q[off + k] : p.qprim[off + k] - p.q[off + ks];
      }
      const double v = KERNEL == 1 ? fabs(d) : d * d;
      db += alpha * (v < p.lambda ? v : p.lambda);
    }
    Dbs[k] = db;
    double di = db;
    for (int i = f0; i < f1; ++i) di += p.msg[(size_t)p.fidx[i] * K + k];
    if (di < bestv) { bestv = di; besti = k; }  // ascending k per thread: first minimum kept
  }
  wave_argmin(bestv, besti);
  if (lane == 0) { redv[wave] = bestv; redi[wave] = besti; }
  __syncthreads();
  if (tid == 0) {
    double v = redv[0];
    int bi = redi[0];
    for (int w = 1; w < kWavesPerBlock; ++w)
      if (redv[w] < v || (redv[w] == v && redi[w] < bi)) { v = redv[w]; bi = redi[w]; }
    p.x[node] = bi;
    p.eterms[r] = Dbs[bi];
  }
}

// ---- persistent dataflow sweep -------------------------------------------------
// One launch = one whole sweep (minimize.cpp:36-62 or :67-95), optionally fused
// with the primal pass of the previous iteration (minimize.cpp:223-264; both
// visit the nodes in the same order and the primal only needs the forward
// messages as they are BEFORE this visit overwrites them).  Workgroups draw
// runs (grid rows, the border chain) from a ticket counter in processing order
// and walk them node by node; a node starts when the completion flags of its
// other incoming neighbours carry this launch's epoch.  Runs only ever wait on
// runs with a smaller ticket, which are already held by resident workgroups, so
// any grid size makes progress.  Messages produced in this launch travel
// write-through (sc1 store -> vmcnt(0) -> barrier -> sc1 flag; consumer: sc1
// poll -> sc1 loads), the hand-over to the next node of the same run goes
// through LDS.
constexpr int kMaxSlots = TrwsGraph::kMaxSlots;
constexpr int kSpinLimit = 1 << 22;  // polls before a launch gives up (bounded spin)

template <int KERNEL, bool BACKWARD, int MODE, bool PRIMAL, bool UPDATE>
__global__ __launch_bounds__(kBlock) void trws_persistent_kernel(DevParams p, int epoch) {
  extern __shared__ __attribute__((aligned(16))) double lds[];
  const int K = p.K, Kp = p.Kp;
  double *Di = lds;                       // Kp
  double *red = lds + Kp;                 // 8
  double *hand = lds + Kp + 8;            // kMaxSlots * Kp : messages for the next node of the run
  double *Dbs = hand + kMaxSlots * Kp;    // Kp : DiBackward of the primal pass
  double *wscratch = Dbs + Kp;            // per-wave scratch of the LDS message path
  int *s_run = (int *)(red + 6);
  const int tid = threadIdx.x, lane = tid & (kWave - 1), wave = tid / kWave;
  double *scratch = wscratch + (size_t)wave * (kWaveVecs * Kp + 8);
  constexpr int D = BACKWARD ? 1 : 0;
  const int32_t *optr = BACKWARD ? p.bptr : p.fptr, *oidx = BACKWARD ? p.bidx : p.fidx;
  const int32_t *iptr = BACKWARD ? p.fptr : p.bptr, *iidx = BACKWARD ? p.fidx : p.bidx;
  const int8_t *in_slot = p.in_slot[D];
  const int N = p.N;
  for (;;) {
    if (tid == 0) { const int t_ = atomicAdd(p.ticket, 1); *s_run = t_ < p.ntickets[BACKWARD ? 1 : 0] ? (p.run_order[BACKWARD ? 1 : 0] ? p.run_order[BACKWARD ? 1 : 0][t_] : t_) : p.nruns[BACKWARD ? 1 : 0]; }
    __syncthreads();
    const int run = *s_run;
    __syncthreads();
    if (run >= p.nruns[D]) break;
    const int p0 = p.run_ptr[D][run], p1 = p.run_ptr[D][run + 1];
    long long tprev = p.prof ? (long long)__builtin_readcyclecounter() : 0;
#define PROF(slot)                                                                 \
  if (p.prof && tid == 0 && run == 0) {                                            \
    const long long tn = (long long)__builtin_readcyclecounter();                  \
    atomicAdd(p.prof + (slot), (unsigned long long)(tn - tprev));                  \
    tprev = tn;                                                                    \
  }
    for (int pos = p0; pos < p1; ++pos) {
      const int r = BACKWARD ? N - 1 - pos : pos;
      const int node = p.order[r];
      const int o0 = optr[r], o1 = optr[r + 1], i0 = iptr[r], i1 = iptr[r + 1];
      // ---- wait for the incoming neighbours that other workgroups own
      const int d0 = p.dep_ptr[D][r], nd = p.dep_ptr[D][r + 1] - d0;
      int gave_up = 0;
      if (tid < nd) {
        const int32_t *flag = p.done + p.dep_rank[D][d0 + tid];
        int spins = 0;
        while (ld_sc1(flag) < epoch) {
          __builtin_amdgcn_s_sleep(1);
          if (++spins > kSpinLimit || ((spins & 1023) == 0 && ld_sc1(p.abort_flag))) {
            st_sc1(p.abort_flag, 1);
            gave_up = 1;
            break;
          }
        }
      }
      if (__syncthreads_or(gave_up)) return;  // bounded spin: the host reports the failure
      PROF(0)
      // ---- primal of the previous iteration (needs the outgoing messages before the update)
      if (PRIMAL) {
        double bestv = __builtin_huge_val();
        int besti = 0x7fffffff;
        for (int k = tid; k < K; k += kBlock) {
          double db = p.unary[(size_t)node * K + k];
          // incoming list of the forward order = backward edges (minimize.cpp:240-247)
          for (int i = i0; i < i1; ++i) {
            const int e = iidx[i];
            const int ks = ld_sc1(p.x + p.tail[e]);
            const double alpha = p.alpha[e];
            double d;
            if (p.pos) {
              d = p.mdir[e] == 0 ? p.pos[ks] - p.pos[k] : p.pos[k] - p.pos[ks];
            } else {
              const size_t off = (size_t)e * K;
              d = p.mdir[e] == 0 ? p.qprim[off + ks] - p.q[off + k] : p.qprim[off + k] - p.q[off + ks];
            }
            const double v = KERNEL == 1 ? fabs(d) : d * d;
            db += alpha * (v < p.lambda ? v : p.lambda);
          }
          Dbs[k] = db;
          double di = db;
          for (int i = o0; i < o1; ++i) di += p.msg[(size_t)oidx[i] * K + k];
          if (di < bestv) { bestv = di; besti = k; }
        }
        wave_argmin(bestv, besti);
        if (lane == 0) { red[wave] = bestv; ((int *)(red + 4))[wave] = besti; }
        __syncthreads();
        if (tid == 0) {
          double v = red[0];
          int bi = ((int *)(red + 4))[0];
          for (int w = 1; w < kWavesPerBlock; ++w) {
            const double rv = red[w];
            const int ri = ((int *)(red + 4))[w];
            if (rv < v || (rv == v && ri < bi)) { v = rv; bi = ri; }
          }
          st_sc1(p.x + node, bi);
          p.eterms[r] = Dbs[bi];
        }
        __syncthreads();
        PROF(1)
      }
      if (UPDATE) {
        // ---- Di = D + outgoing-list messages (from the previous sweep) + incoming ones
        double vloc = __builtin_huge_val();
        for (int k = tid; k < K; k += kBlock) {
          double acc = p.unary[(size_t)node * K + k];
          for (int i = o0; i < o1; ++i) acc += p.msg[(size_t)oidx[i] * K + k];
          for (int i = i0; i < i1; ++i) {
            const int sl = in_slot[i];  // 0..7: previous visit (LDS); 8..15: two visits back -> HBM
            acc += (sl >= 0 && sl < 8) ? hand[sl * Kp + k] : ld_sc1(p.msg + (size_t)iidx[i] * K + k);
          }
          Di[k] = acc;
          vloc = acc < vloc ? acc : vloc;
        }
        if (BACKWARD) {
          vloc = wave_min(vloc);
          if (lane == 0) red[wave] = vloc;
          __syncthreads();
          double vmin = red[0];
#pragma unroll
          for (int w = 1; w < kWavesPerBlock; ++w) vmin = red[w] < vmin ? red[w] : vmin;
          for (int k = tid; k < K; k += kBlock) Di[k] -= vmin;
          if (tid == 0) p.lbterms[p.lb_pos_node[r]] = vmin;
        }
        __syncthreads();  // Di complete, previous hand-over consumed
        PROF(2)
        const double gamma = p.gamma[r];
        for (int i = o0 + wave; i < o1; i += kWavesPerBlock) {
          const int e = oidx[i];
          const int sl = i - o0;
          const double v = update_message<KERNEL, BACKWARD, MODE, true>(
              p, e, Di, gamma, scratch, sl < kMaxSlots ? hand + sl * Kp : nullptr, lane);
          if (BACKWARD && lane == 0) p.lbterms[p.lb_pos_edge[e]] = v;
        }
        PROF(3)
      }
      // ---- publish: every storing wave drains, then one lane raises the flag
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
      if (tid == 0) st_sc1(p.done + r, epoch);
      PROF(4)
    }
#undef PROF
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
  return d;
}

// bounded wait for one completion flag; returns false if the launch must give up
__device__ __forceinline__ bool wait_flag(const DevParams &p, int rank, int epoch) {
  const int32_t *flag = p.done + rank;
  int spins = 0;
  while (ld_sc1(flag) < epoch) {
    __builtin_amdgcn_s_sleep(1);
    if (++spins > kSpinLimit || ((spins & 1023) == 0 && ld_sc1(p.abort_flag))) {
      st_sc1(p.abort_flag, 1);
      return false;
    }
  }
  return true;
}

// ---- lane exchange lane ^ S without an address register where the hardware offers one
template <int S>
__device__ __forceinline__ unsigned xor_lane_u32(unsigned v) {
  if (S == 1) return (unsigned)__builtin_amdgcn_update_dpp((int)v, (int)v, 0xB1, 0xF, 0xF, false);   // quad_perm [1,0,3,2]
  if (S == 2) return (unsigned)__builtin_amdgcn_update_dpp((int)v, (int)v, 0x4E, 0xF, 0xF, false);   // quad_perm [2,3,0,1]
  if (S == 8) return (unsigned)__builtin_amdgcn_update_dpp((int)v, (int)v, 0x128, 0xF, 0xF, false);  // row_ror:8
  if (S == 4 || S == 16) return (unsigned)__builtin_amdgcn_ds_swizzle((int)v, 0x1F | (S << 10));    // bit mode: xor S
  return (unsigned)__shfl_xor((int)v, S, kWave);
}
// Bitonic sort of two independent sets of 64 unsigned keys (one key of each per lane), ascending by lane.
template <int SIZE, int STRIDE>
__device__ __forceinline__ void bitonic_step2(unsigned &a, unsigned &b, int lane) {
  const unsigned oa = xor_lane_u32<STRIDE>(a), ob = xor_lane_u32<STRIDE>(b);
  const bool keep_min = ((lane & STRIDE) == 0) == ((lane & SIZE) == 0);
  a = keep_min ? (a < oa ? a : oa) : (a > oa ? a : oa);
  b = keep_min ? (b < ob ? b : ob) : (b > ob ? b : ob);
}
__device__ __forceinline__ void wave_sort2(unsigned &a, unsigned &b, int lane) {
  bitonic_step2<2, 1>(a, b, lane);
  bitonic_step2<4, 2>(a, b, lane); bitonic_step2<4, 1>(a, b, lane);
  bitonic_step2<8, 4>(a, b, lane); bitonic_step2<8, 2>(a, b, lane); bitonic_step2<8, 1>(a, b, lane);
  bitonic_step2<16, 8>(a, b, lane); bitonic_step2<16, 4>(a, b, lane); bitonic_step2<16, 2>(a, b, lane);
  bitonic_step2<16, 1>(a, b, lane);
  bitonic_step2<32, 16>(a, b, lane); bitonic_step2<32, 8>(a, b, lane); bitonic_step2<32, 4>(a, b, lane);
  bitonic_step2<32, 2>(a, b, lane); bitonic_step2<32, 1>(a, b, lane);
  bitonic_step2<64, 32>(a, b, lane); bitonic_step2<64, 16>(a, b, lane); bitonic_step2<64, 8>(a, b, lane);
  bitonic_step2<64, 4>(a, b, lane); bitonic_step2<64, 2>(a, b, lane); bitonic_step2<64, 1>(a, b, lane);
}

// Second look at a message whose certificate failed (cold path, kept out of line so that it costs the
// hot path no registers).  A cone whose apex lies above vTrunc by more than alpha times the whole
// position range cannot touch a useful cone -- every useful cone dominates it with that margin
// wherever they meet -- so it neither counts for the magnitude behind delta nor for the tangency
// test.  Out-of-range plane proposals (unary ~ 4e7, dispmap_ncc.m:245) would otherwise inflate delta
// and send almost every message of such a fusion to the serial construction.  Returns "still bad";
// m1 = min-plus value over the useful sources.
__device__ __attribute__((noinline)) bool message_second_look(double lambda, int K, double alpha, double h,
                                                              double qsrc, double t, double vtrunc,
                                                              double delta, int lane, double &m1_out) {
  const double inf = __builtin_huge_val();
  const bool act = lane < K;
  const double aq = alpha * qsrc;
  const double qabs = wave_max_dpp(act ? max_raw(fabs(qsrc), fabs(t)) : 0.0);
  const bool rel = act && h <= vtrunc + 2.000002 * fabs(alpha) * qabs;
  const double mag2 = max_raw(wave_max_dpp(act ? (rel ? fabs(h) : 0.0) + fabs(aq) + alpha * fabs(t) : 0.0), fabs(vtrunc));
  const double delta2 = 1e-9 * (mag2 + fabs(alpha * lambda));
  if (!(delta2 < delta)) return true;
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
    m2 = hi > lo ? min_raw(m2, hi) : m2;
    m1 = lo;
    const double aqj = alpha * qj;
    const bool near = (fabs(ui - (hj - aqj)) <= delta2) || (fabs(vi - (hj + aqj)) <= delta2);
    bad = bad || (near && qsrc != qj && rel);
  }
  bad = bad || (m1 < vtrunc && !(m2 - m1 > delta2 && vtrunc - m1 > delta2));
  m1_out = m1;
  return UNI(act && bad);
}

// Message update with everything in registers (K <= 64): h = gamma*Di - old message,
// qsrc / t = source / destination positions, perm = ascending order of the sources
// (only touched by the serial fallback).  Returns the normalised message in `out`.
template <int KERNEL>
__device__ __forceinline__ double message_regs(const DevParams &p, int K, double alpha, double h,
                                               double qsrc, double t, const uint16_t *perm,
                                               double &outmsg, int lane, double *hq = nullptr,
                                               int window = -1) {
  const double inf = __builtin_huge_val();
  const bool act = lane < K;
  // hmin and the magnitude behind delta in one interleaved reduction
  const double aq = alpha * qsrc;
  double hmin = h, mag = act ? fabs(h) + fabs(aq) + alpha * fabs(t) : 0.0;  // inactive lanes hold h = +inf
  wave_min_max_dpp(hmin, mag);
  double out, vmin;
  if (UNI(alpha == 0)) {
    out = hmin; vmin = hmin;  // typeStereoLinear.h:390-396
  } else {
    const double vtrunc = hmin + alpha * p.lambda;
    bool need_serial = true;
    out = vtrunc;
    long long tm0 = p.prof ? (long long)__builtin_readcyclecounter() : 0;  // development profile: slots 8..15
#define MSTAMP(slot) do { if (p.prof) { const long long n_ = (long long)__builtin_readcyclecounter(); if (lane == 0) { atomicAdd(p.prof + (slot), (unsigned long long)(n_ - tm0)); atomicAdd(p.prof + (slot) + 1, 1ull); } tm0 = n_; } } while (0)
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
      // The sources are broadcast from a per-wave LDS table (one ds_read_b128 per source instead of
      // eight v_readlane); two sources per trip keep two independent dependency chains in flight.
      // m1 / m2 = smallest and second smallest DISTINCT cost seen so far.
      // table entry of a source: (h, q, u, v) -- the tangency test then needs no arithmetic on the source
      if (hq) {
        hq[4 * lane] = h; hq[4 * lane + 1] = qsrc; hq[4 * lane + 2] = ui; hq[4 * lane + 3] = vi;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
      }
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
    m2 = hi > lo ? min_raw(m2, hi) : m2;                                             \
    m1 = lo;                                                                         \
    const bool near = (fabs(ui - HJ##u) <= delta) || (fabs(vi - HJ##v) <= delta);    \
    bad = bad || (near && qsrc != QJ);                                               \
  }
      if (window >= 0 && __builtin_popcountll(mask) > 32) {
        // Flat h (the zig-zag rows: gamma = 1/6 .. 1/8 makes almost every source useful) on shared
        // strictly ascending positions.  The pair loop below would be K^2; instead
        //  * tangency for ALL pairs by sorting u and v (conservative superset of the useful pairs):
        //    keys quantised to 32 bits over a range that certainly contains them, "within delta"
        //    tested as "within delta / resolution + 2 units";
        //  * min-plus only over the sources inside the truncation window of each destination (a source
        //    farther than lambda costs >= vTrunc exactly); the table is padded with +inf entries.
        const double hmax = wave_max_dpp(act ? h : -inf);
        const double ap0 = alpha * p.pos_first, ap1 = alpha * p.pos_last;
        const double aplo = min_raw(ap0, ap1), aphi = max_raw(ap0, ap1);
        const double span = (hmax - hmin) + (aphi - aplo);  // >= max u - min u and >= max v - min v
        const double scale = 4294967040.0 / span;           // (2^32 - 256) / span
        bad = bad || !(span < inf) || !(span > 0) || !(delta * scale < 1e9);
        unsigned ku = 0xFFFFFFFFu, kv = 0xFFFFFFFFu;
        if (act && !bad) {
          ku = (unsigned)((ui - (hmin - aphi)) * scale);
          kv = (unsigned)((vi - (hmin + aplo)) * scale);
        }
        wave_sort2(ku, kv, lane);
        const unsigned un = (unsigned)__shfl_down((int)ku, 1, kWave), vn = (unsigned)__shfl_down((int)kv, 1, kWave);
        const unsigned thr = bad ? 0u : (unsigned)(delta * scale) + 2u;
        bad = bad || (lane + 1 < K && (un - ku <= thr || vn - kv <= thr));
        for (int d = -window; d <= window; ++d) {
          const double hj = hq[4 * (lane + d)], qj = hq[4 * (lane + d) + 1];
          const double c = pair_cost<1>(alpha, t - qj, hj);
          const double lo = min_raw(m1, c), hi = max_raw(m1, c);
          m2 = hi > lo ? min_raw(m2, hi) : m2;
          m1 = lo;
        }
      } else {
      while (mask) {
        const int j0 = __builtin_ctzll(mask);
        mask &= mask - 1;
        const int j1 = mask ? __builtin_ctzll(mask) : j0;  // a source visited twice changes nothing
        mask &= mask - 1;
        STEREO_SRC(j0, hj0, qj0)
        STEREO_SRC(j1, hj1, qj1)
        STEREO_ACC(hj0, qj0)
        STEREO_ACC(hj1, qj1)
      }
      }
#undef STEREO_SRC
#undef STEREO_ACC
      bad = bad || (m1 < vtrunc && !(m2 - m1 > delta && vtrunc - m1 > delta));
      need_serial = UNI(act && bad);
      MSTAMP(8);
      if (need_serial) { need_serial = message_second_look(p.lambda, K, alpha, h, qsrc, t, vtrunc, delta, lane, m1); MSTAMP(10); }
      out = m1 < vtrunc ? m1 : vtrunc;
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
    if (need_serial) {
      const int idx = act ? perm[lane] : lane;
      const double hs = __shfl(h, idx, kWave), qs = __shfl(qsrc, idx, kWave);
      double sh, sq, zz;
      int maxtop = 0;
      bool built = false;
      if (KERNEL == 1 && !(p.debug & 512)) built = build_envelope_masks(K, alpha, hs, qs, sh, sq, zz, lane, maxtop);
      if (!built) maxtop = build_envelope_regs<KERNEL>(K, alpha, hs, qs, sh, sq, zz, lane);
      MSTAMP(12);
      // the reference walks up the stack while z[j+1] < t (typeStereoLinear.h:462-479): the slot it
      // stops at is the FIRST one whose upper breakpoint is not below t (stale slots above `top`
      // included, hence up to the highest slot ever written); found top-down so that the lowest wins
      int slot = maxtop;
      for (int j = maxtop - 1; j >= 0; --j) {
        const double zj1 = readlane_f64(zz, j);
        slot = !(zj1 < t) ? j : slot;
      }
      const double ch = __shfl(sh, slot, kWave), cq = __shfl(sq, slot, kWave);
      const double c = pair_cost<KERNEL>(alpha, t - cq, ch);
      out = c < vtrunc ? c : vtrunc;
      MSTAMP(14);
    }
#undef MSTAMP
    vmin = wave_min_dpp(act ? out : inf);
  }
  outmsg = out - vmin;
  return vmin;
}

#undef RLI

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
constexpr int kPipeCompute = 8;  // one compute wave per outgoing message (<= 8 per node)
constexpr int kPipeWaves = kPipeCompute + 4;  // loader, storer, (idle), primal: the primal wave lands on SIMD 3,
                                              // which otherwise hosts one compute wave only
constexpr int kPipeThreads = kPipeWaves * kWave;
// LDS stage layout (doubles): D[64] m[8][64] qv[8][64] qpv[8][64] | a[8] | ints: desc[64] px[8]
constexpr int kStD = 0, kStM = 64, kStQ = 64 + 512, kStQP = 64 + 1024, kStA = 64 + 1536;
constexpr int kStI = kStA + 8;                    // int area starts here (as doubles)
constexpr int kStageDoubles = kStI + 36;          // 64 + 8 ints = 36 doubles
constexpr int kScalDoubles = 16;                  // newv[8], node_vmin, prim_e, x (as int)
constexpr int kPipePad = 16;                      // source tables are padded by this many (+inf) entries on both sides
constexpr int kPipeTab = 4 * (kWave + 2 * kPipePad);  // doubles per compute wave: (h, q, u, v) x 96

template <int KERNEL, bool BACKWARD, bool PRIMAL, bool UPDATE, bool SHARED>
__device__ __forceinline__ void pipe_body(DevParams p, int epoch) {
  extern __shared__ __attribute__((aligned(16))) double lds[];
  double *stage0 = lds;                                   // 2 stages
  double *hand = lds + 2 * kStageDoubles;                 // ring of 4 x 8 x 64: the last visits' new messages
  double *scal = hand + 4 * 8 * kWave;                    // 2 x kScalDoubles
  double *hqtab = scal + 2 * kScalDoubles;                // per compute wave: 64 x (h, q, u, v)
  int *ctl = (int *)(hqtab + kPipeCompute * kPipeTab);    // [0] run, [1] abort
  int *dring = ctl + 4;                                   // the last three descriptors (the storer's comes from here, not from HBM)
  const int K = p.K;
  const double inf = __builtin_huge_val();
  const int tid = threadIdx.x, lane = tid & (kWave - 1);
  const int wave = __builtin_amdgcn_readfirstlane(tid / kWave);
  constexpr int D = BACKWARD ? 1 : 0;
  constexpr int DW = TrwsGraph::kDescWords;
  const int32_t *desc = p.desc[D];
  const bool act = lane < K;
  const double posk = (SHARED && act) ? p.pos[lane] : 0.0;
  if (tid == 0) ctl[1] = 0;
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
    if (p.timeline && tid == 0) p.timeline[((size_t)D * p.nruns[0] + run) * 2] = wall_clock64();
    unsigned long long busy = 0;

    for (int pos = p0 - 1; pos <= p1; ++pos) {
      const long long tstart = p.prof ? (long long)__builtin_readcyclecounter() : 0;
      double *st = stage0 + (pos & 1) * kStageDoubles;          // node `pos`
      double *stn = stage0 + ((pos + 1) & 1) * kStageDoubles;   // node `pos + 1`
      double *hcur = hand + (pos & 3) * 8 * kWave, *hprev = hand + ((pos - 1) & 3) * 8 * kWave;
      double *hprev2 = hand + ((pos - 2) & 3) * 8 * kWave;
      double *sc = scal + (pos & 1) * kScalDoubles;
      const bool have_node = pos >= p0 && pos < p1;

      if (wave < kPipeCompute) {
        // ------------------------------------------------------------ compute
        if (UPDATE && have_node) {
          const int *sti = (const int *)(st + kStI);
          const int f = __builtin_amdgcn_readfirstlane(sti[2]);
          const int nout = f & 15, nin = (f >> 4) & 15, md = (f >> 16) & 255, ntot = nout + nin;
          const unsigned slA = (unsigned)__builtin_amdgcn_readfirstlane(sti[41]);
          const unsigned slB = (unsigned)__builtin_amdgcn_readfirstlane(sti[42]);
          if (wave < nout || (BACKWARD && wave == 0)) {  // waves without a message stay out of the way
          double Di = act ? st[kStD + lane] : 0.0;
          double mown = 0;
#pragma unroll
          for (int j = 0; j < 8; ++j) {
            if (j < ntot) {
              double v;
              const int sl = j >= nout ? (int)(signed char)(((j < 4 ? slA : slB) >> (8 * (j & 3))) & 255) : -1;
              if (sl >= 8) v = hprev2[(sl - 8) * kWave + lane];
              else if (sl >= 0) v = hprev[sl * kWave + lane];
              else v = st[kStM + j * kWave + lane];
              Di += v;
              if (j == wave && j < nout) mown = v;
            }
          }
          double node_vmin = 0;
          if (BACKWARD) {
            node_vmin = wave_min_dpp(act ? Di : inf);
            Di -= node_vmin;
            if (tid == 0) sc[8] = node_vmin;
          }
#pragma unroll
          for (int j = 0; j < 8; ++j) {
            if (j == wave && j < nout) {
              const double gamma = (double)1 / (double)(nout > nin ? nout : nin);
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
              double newm = 0;
              const double v = message_regs<KERNEL>(p, K, alpha, h, qsrc, qdst, perm, newm, lane,
                                                    hqtab + wave * kPipeTab + 4 * kPipePad, (SHARED && p.win_ok) ? p.window : -1);
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
          const int ntot = nx.nout + nx.nin;
          // everything that does not depend on other workgroups is requested first ...
          double dk = 0, mv[8], qv[8], qpv[8];
          if (act) dk = p.unary[(size_t)nx.node * K + lane];
#pragma unroll
          for (int j = 0; j < 8; ++j) {
            mv[j] = 0; qv[j] = 0; qpv[j] = 0;
            if (j < ntot && act) {
              const size_t off = (size_t)nx.e[j] * K + lane;
              if (j < nx.nout && (UPDATE || PRIMAL)) mv[j] = p.msg[off];
              if (!SHARED) { qv[j] = p.q[off]; qpv[j] = p.qprim[off]; }
            }
          }
          double av = 0;
          int pxv = 0, xn = 0, sl = 0;
          if (lane < ntot) {
            int ej = 0;  // lane j fetches the scalars of edge j
#pragma unroll
            for (int j = 0; j < 8; ++j)
              if (lane == j) { ej = nx.e[j]; xn = nx.xn[j]; sl = nx.slot[j]; }
            av = p.alpha[ej];
          }
          // ... then the completion flags of the foreign neighbours, then their data
          // (all flags are polled together: lane j watches dependency j)
          if (nx.ndep > 0) {
            int myrank = nx.dep[0];
#pragma unroll
            for (int j = 1; j < 4; ++j)
              if (lane == j) myrank = nx.dep[j];
            const bool watching = lane < nx.ndep;
            int spins = 0;
            bool ok = true;
            for (;;) {
              const int v = watching ? ld_sc1(p.done + myrank) : epoch;
              if (!UNI(v < epoch)) break;
              __builtin_amdgcn_s_sleep(1);
              if (++spins > kSpinLimit || ((spins & 1023) == 0 && ld_sc1(p.abort_flag))) { ok = false; break; }
            }
            if (!ok && lane == 0) { st_sc1(p.abort_flag, 1); ctl[1] = 1; }
          }
#pragma unroll
          for (int j = 0; j < 8; ++j)
            if (UPDATE && j >= nx.nout && j < ntot && nx.slot[j] < 0 && act)
              mv[j] = ld_sc1(p.msg + (size_t)nx.e[j] * K + lane);
          if (PRIMAL && lane < ntot && lane >= nx.nout && sl < 0) pxv = ld_sc1(p.x + xn);
          if (act) stn[kStD + lane] = dk;
#pragma unroll
          for (int j = 0; j < 8; ++j) {
            if (j < ntot && act) {
              stn[kStM + j * kWave + lane] = mv[j];
              if (!SHARED) { stn[kStQ + j * kWave + lane] = qv[j]; stn[kStQP + j * kWave + lane] = qpv[j]; }
            }
          }
          if (lane < 8) { stn[kStA + lane] = av; stni[64 + lane] = pxv; }
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
                if (act) st_sc1(mb + (size_t)pd.e[j] * K + lane, hprev[j * kWave + lane]);
                if (BACKWARD && lane == 0) p.lbterms[pd.lbe[j]] = scp[j];
              }
            }
            if (BACKWARD && lane == 0) p.lbterms[pd.lbn] = scp[8];
          }
          if (PRIMAL && lane == 0) {
            const int xi = ((const int *)(scp + 10))[0];
            st_sc1(p.x + pd.node, xi);
            if (pd.remote & (1 << 16)) st_sc1(p.peer_x0 + pd.node, xi);
            if (pd.remote & (1 << 17)) st_sc1(p.peer_x1 + pd.node, xi);
            p.eterms[pd.epos] = scp[9];
          }
          asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
          if (lane == 0) {
            st_sc1(p.done + pd.rank, epoch);
            if (pd.remote & (1 << 16)) st_sc1(p.peer_done0 + pd.rank, epoch);
            if (pd.remote & (1 << 17)) st_sc1(p.peer_done1 + pd.rank, epoch);
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
              db += st[kStA + j] * (v < p.lambda ? v : p.lambda);
            }
          }
          double di = db;
#pragma unroll
          for (int j = 0; j < 8; ++j)
            if (j < nout) di += st[kStM + j * kWave + lane];
          const int bi = wave_argmin_dpp(act ? di : inf, act ? lane : 0x7fffffff);
          xprev2 = xprev; xprev = bi;
          const double eb = readlane_f64(db, bi);
          if (lane == 0) { sc[9] = eb; ((int *)(sc + 10))[0] = bi; }
        }
      }
      if (p.prof) busy += (unsigned long long)((long long)__builtin_readcyclecounter() - tstart);
      __syncthreads();
      if (ctl[1]) return;  // a dependency wait gave up (bounded spin); host reports it
    }
    if (p.timeline && tid == 0) p.timeline[((size_t)D * p.nruns[0] + run) * 2 + 1] = wall_clock64();
    if (p.prof && lane == 0 && (p.prof_run < 0 || run == p.prof_run)) {
      // busy cycles before the barrier per role: compute (wave 0), loader, storer, primal; steps
      const int slot = wave == 0 ? 0 : wave == kPipeCompute ? 1 : wave == kPipeCompute + 1 ? 2 : wave == kPipeCompute + 3 ? 3 : -1;
      if (slot >= 0) atomicAdd(p.prof + slot, busy);
      if (wave == 0) atomicAdd(p.prof + 6, (unsigned long long)(p1 - p0));
    }
  }
}

template <int KERNEL, bool BACKWARD, bool PRIMAL, bool UPDATE, bool SHARED>
__global__ __launch_bounds__(kPipeThreads) void trws_pipe_kernel(DevParams p, int epoch) {
  pipe_body<KERNEL, BACKWARD, PRIMAL, UPDATE, SHARED>(p, epoch);
}

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
__device__ __forceinline__ int group_strip(const GroupArgs &ga) {
  int s = 0;
#pragma unroll
  for (int i = 1; i < kMaxGroup; ++i) s = (i < ga.n && (int)blockIdx.x >= ga.first[i]) ? i : s;  // static indices only
  return s;
}
template <int KERNEL, bool BACKWARD, bool PRIMAL, bool UPDATE, bool SHARED>
__global__ __launch_bounds__(kPipeThreads) void trws_pipe_group_kernel(GroupArgs ga, int epoch) {
  pipe_body<KERNEL, BACKWARD, PRIMAL, UPDATE, SHARED>(ga.pp[group_strip(ga)], epoch);
}

// ---- pipelined persistent sweep for 64 < K <= 128 (two labels per lane), linear kernel -------
// The same role-specialised structure as trws_pipe_kernel -- eight compute waves (one message each),
// loader, storer, primal, one barrier per visit -- with rows of 128 labels: lane owns labels lane
// and lane + 64.  General (per-edge) or shared positions; this is what a simultaneous fusion of
// 64 .. 127 proposals runs on (example_ncc.m fuses 78).  A message is min-plus over the useful
// sources from a per-wave (h, q, u, v) table in LDS plus the certificate of DESIGN.md 4.3; if that
// fails the reference's serial construction runs in the wave's own LDS scratch.
constexpr int k2W = 2 * kWave;                      // row width
constexpr int k2StD = 0, k2StM = k2W, k2StQ = k2W + 8 * k2W, k2StQP = k2W + 16 * k2W, k2StA = k2W + 24 * k2W;
constexpr int k2StI = k2StA + 8;
constexpr int k2Stage = k2StI + 36;
constexpr int k2Fb = 5 * (k2W + 2);                 // serial scratch per compute wave: sorted h, q; stack h, q; breakpoints
constexpr int k2LdsDoubles = 2 * k2Stage + 4 * 8 * k2W + 2 * kScalDoubles + kPipeCompute * 4 * k2W + kPipeCompute * k2Fb + 2;
static_assert(k2LdsDoubles * 8 <= 160 * 1024, "pipe2 kernel LDS");

template <bool BACKWARD, bool PRIMAL, bool UPDATE, bool SHARED>
__global__ __launch_bounds__(kPipeThreads) void trws_pipe2_kernel(DevParams p, int epoch) {
  extern __shared__ __attribute__((aligned(16))) double lds[];
  double *stage0 = lds;                                   // 2 stages
  double *hand = lds + 2 * k2Stage;                       // ring of 4 x 8 x 128
  double *scal = hand + 4 * 8 * k2W;                      // 2 x kScalDoubles
  double *tabs = scal + 2 * kScalDoubles;                 // per compute wave: 128 x (h, q, u, v)
  double *fbs = tabs + kPipeCompute * 4 * k2W;            // serial scratch, one per compute wave
  int *ctl = (int *)(fbs + kPipeCompute * k2Fb);          // [0] run, [1] abort
  const int K = p.K;
  const double inf = __builtin_huge_val();
  const int tid = threadIdx.x, lane = tid & (kWave - 1);
  const int wave = __builtin_amdgcn_readfirstlane(tid / kWave);
  constexpr int D = BACKWARD ? 1 : 0;
  constexpr int DW = TrwsGraph::kDescWords;
  const int32_t *desc = p.desc[D];
  const bool act[2] = {lane < K, lane + kWave < K};
  const int kk[2] = {lane, lane + kWave};
  double posk[2] = {0, 0};
  if (SHARED) {
    if (act[0]) posk[0] = p.pos[kk[0]];
    if (act[1]) posk[1] = p.pos[kk[1]];
  }
  if (tid == 0) ctl[1] = 0;

  for (;;) {
    if (tid == 0) { const int t_ = atomicAdd(p.ticket, 1); ctl[0] = t_ < p.ntickets[D] ? (p.run_order[D] ? p.run_order[D][t_] : t_) : p.nruns[D]; }
    __syncthreads();
    const int run = __builtin_amdgcn_readfirstlane(ctl[0]);
    __syncthreads();
    if (run >= p.nruns[D]) break;
    const int p0 = p.run_ptr[D][run], p1 = p.run_ptr[D][run + 1];
    int xprev = 0, xprev2 = 0;
    int wnext = 0;
    if (wave == kPipeCompute) wnext = desc[(size_t)p0 * DW + lane];
    if (p.timeline && tid == 0) p.timeline[((size_t)D * p.nruns[0] + run) * 2] = wall_clock64();

    for (int pos = p0 - 1; pos <= p1; ++pos) {
      double *st = stage0 + (pos & 1) * k2Stage;
      double *stn = stage0 + ((pos + 1) & 1) * k2Stage;
      double *hcur = hand + (pos & 3) * 8 * k2W, *hprev = hand + ((pos - 1) & 3) * 8 * k2W;
      double *hprev2 = hand + ((pos - 2) & 3) * 8 * k2W;
      double *sc = scal + (pos & 1) * kScalDoubles;
      const bool have_node = pos >= p0 && pos < p1;

      if (wave < kPipeCompute) {
        // ------------------------------------------------------------ compute
        if (UPDATE && have_node) {
          const int *sti = (const int *)(st + k2StI);
          const int f = __builtin_amdgcn_readfirstlane(sti[2]);
          const int nout = f & 15, nin = (f >> 4) & 15, md = (f >> 16) & 255, ntot = nout + nin;
          const unsigned slA = (unsigned)__builtin_amdgcn_readfirstlane(sti[41]);
          const unsigned slB = (unsigned)__builtin_amdgcn_readfirstlane(sti[42]);
          if (wave < nout || (BACKWARD && wave == 0)) {
            double Di[2] = {st[k2StD + kk[0]], st[k2StD + kk[1]]};
            double mown[2] = {0, 0};
#pragma unroll
            for (int j = 0; j < 8; ++j) {
              if (j < ntot) {
                const int sl = j >= nout ? (int)(signed char)(((j < 4 ? slA : slB) >> (8 * (j & 3))) & 255) : -1;
                const double *row = sl >= 8 ? hprev2 + (sl - 8) * k2W : sl >= 0 ? hprev + sl * k2W : st + k2StM + j * k2W;
                const double v0 = row[kk[0]], v1 = row[kk[1]];
                Di[0] += v0; Di[1] += v1;
                if (j == wave && j < nout) { mown[0] = v0; mown[1] = v1; }
              }
            }
            if (BACKWARD) {
              const double node_vmin = wave_min_dpp(min_raw(act[0] ? Di[0] : inf, act[1] ? Di[1] : inf));
              Di[0] -= node_vmin; Di[1] -= node_vmin;
              if (tid == 0) sc[8] = node_vmin;
            }
#pragma unroll
            for (int j = 0; j < 8; ++j) {
              if (j == wave && j < nout) {
                const double gamma = (double)1 / (double)(nout > nin ? nout : nin);
                const bool src_is_qprim = ((BACKWARD ? 1 : 0) == ((md >> j) & 1));
                const double alpha = st[k2StA + j];
                double h[2], qsrc[2], qdst[2];
                const uint16_t *perm = p.perm_pos;
#pragma unroll
                for (int c = 0; c < 2; ++c) {
                  h[c] = act[c] ? gamma * Di[c] - mown[c] : inf;
                  qsrc[c] = posk[c]; qdst[c] = posk[c];
                  if (!SHARED) {
                    const double a_ = act[c] ? st[k2StQ + j * k2W + kk[c]] : 0.0;
                    const double b_ = act[c] ? st[k2StQP + j * k2W + kk[c]] : 0.0;
                    qsrc[c] = src_is_qprim ? b_ : a_;
                    qdst[c] = src_is_qprim ? a_ : b_;
                  }
                }
                if (!SHARED) {
                  const int e = __builtin_amdgcn_readfirstlane(sti[4 + j]);
                  perm = (src_is_qprim ? p.perm_qp : p.perm_q) + (size_t)e * K;
                }
                // ---- message (typeStereoLinear.h:329-487)
                const double hmin = wave_min_dpp(min_raw(h[0], h[1]));
                double out[2], vmin;
                if (UNI(alpha == 0)) {
                  out[0] = out[1] = hmin; vmin = hmin;  // :390-396
                } else {
                  const double vtrunc = hmin + alpha * p.lambda;
                  bool need_serial = p.certificate == 0;
                  out[0] = out[1] = vtrunc;
                  if (!need_serial) {
                    double ui[2], vi[2], mg = 0;
#pragma unroll
                    for (int c = 0; c < 2; ++c) {
                      const double aq = alpha * qsrc[c];
                      ui[c] = h[c] - aq; vi[c] = h[c] + aq;
                      if (act[c]) mg = max_raw(mg, fabs(h[c]) + fabs(aq) + alpha * fabs(qdst[c]));
                    }
                    const double delta = 1e-9 * (wave_max_dpp(mg) + fabs(alpha * p.lambda));
                    double *tab = tabs + wave * 4 * k2W;
#pragma unroll
                    for (int c = 0; c < 2; ++c) {
                      double *e4 = tab + 4 * kk[c];
                      e4[0] = h[c]; e4[1] = qsrc[c]; e4[2] = ui[c]; e4[3] = vi[c];
                    }
                    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                    __builtin_amdgcn_wave_barrier();
                    double m1[2], m2[2];
                    // min-plus over the useful sources + certificate; rel: cones that take part in the tangency test
                    auto certify = [&](double dl, bool rel0, bool rel1) -> bool {
                      const bool rel[2] = {rel0, rel1};
                      m1[0] = m1[1] = inf; m2[0] = m2[1] = inf;
                      bool bad = !(dl < inf);
#pragma unroll
                      for (int cs = 0; cs < 2; ++cs) {
                        unsigned long long mask = __builtin_amdgcn_ballot_w64(act[cs] && h[cs] < vtrunc);
                        while (mask) {
                          const int js = __builtin_ctzll(mask) + cs * kWave;
                          mask &= mask - 1;
                          const double hj = tab[4 * js], qj = tab[4 * js + 1], uj = tab[4 * js + 2], vj = tab[4 * js + 3];
#pragma unroll
                          for (int c = 0; c < 2; ++c) {
                            const double cst = pair_cost<1>(alpha, qdst[c] - qj, hj);
                            const double lo = min_raw(m1[c], cst), hi = max_raw(m1[c], cst);
                            m2[c] = hi > lo ? min_raw(m2[c], hi) : m2[c];
                            m1[c] = lo;
                            const bool near = (fabs(ui[c] - uj) <= dl) || (fabs(vi[c] - vj) <= dl);
                            bad = bad || (near && qsrc[c] != qj && rel[c]);
                          }
                        }
                      }
#pragma unroll
                      for (int c = 0; c < 2; ++c)
                        bad = bad || (act[c] && m1[c] < vtrunc && !(m2[c] - m1[c] > dl && vtrunc - m1[c] > dl));
                      return UNI(bad);
                    };
                    // Hopeless cones (apex above vTrunc by more than alpha times the position range: they
                    // cannot touch a useful cone, see message_second_look) neither set the scale of delta
                    // nor take part in the tangency test -- decided up front here: fusions of this many
                    // proposals always contain out-of-range planes (unary ~ 4e7), and a failed first look
                    // would cost a second pass over the sources.
                    double qa = 0;
#pragma unroll
                    for (int c = 0; c < 2; ++c)
                      if (act[c]) qa = max_raw(qa, max_raw(fabs(qsrc[c]), fabs(qdst[c])));
                    const double hbig = vtrunc + 2.000002 * fabs(alpha) * wave_max_dpp(qa);
                    const bool rel0 = act[0] && h[0] <= hbig, rel1 = act[1] && h[1] <= hbig;
                    double mg2 = 0;
#pragma unroll
                    for (int c = 0; c < 2; ++c)
                      if (act[c]) mg2 = max_raw(mg2, ((c == 0 ? rel0 : rel1) ? fabs(h[c]) : 0.0) + fabs(alpha * qsrc[c]) + alpha * fabs(qdst[c]));
                    const double delta2 = 1e-9 * (max_raw(wave_max_dpp(mg2), fabs(vtrunc)) + fabs(alpha * p.lambda));
                    const bool bad = certify(delta2 < delta ? delta2 : delta, rel0, rel1);
#pragma unroll
                    for (int c = 0; c < 2; ++c) out[c] = m1[c] < vtrunc ? m1[c] : vtrunc;
                    need_serial = UNI(bad);
                    if (need_serial && lane == 0 && p.fallbacks) atomicAdd(p.fallbacks, 1);
                  }
                  if (need_serial) {
                    // the reference's serial construction in this wave's LDS scratch: sources gathered in
                    // ascending position order straight from the registers
                    double *A = fbs + wave * k2Fb, *B = A + (k2W + 2), *sh = B + (k2W + 2), *sq = sh + (k2W + 2), *z = sq + (k2W + 2);
#pragma unroll
                    for (int c = 0; c < 2; ++c) {
                      const int idx = act[c] ? perm[kk[c]] : 0;
                      const double g0h = __shfl(h[0], idx & (kWave - 1), kWave), g1h = __shfl(h[1], idx & (kWave - 1), kWave);
                      const double g0q = __shfl(qsrc[0], idx & (kWave - 1), kWave), g1q = __shfl(qsrc[1], idx & (kWave - 1), kWave);
                      if (act[c]) { A[kk[c]] = idx < kWave ? g0h : g1h; B[kk[c]] = idx < kWave ? g0q : g1q; }
                    }
                    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                    __builtin_amdgcn_wave_barrier();
                    if (lane == 0) build_envelope<1>(K, alpha, A, B, sh, sq, z);
                    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                    __builtin_amdgcn_wave_barrier();
#pragma unroll
                    for (int c = 0; c < 2; ++c) {
                      if (act[c]) {
                        int jj = 0;
                        while (z[jj + 1] < qdst[c]) ++jj;
                        const double cst = pair_cost<1>(alpha, qdst[c] - sq[jj], sh[jj]);
                        out[c] = cst < vtrunc ? cst : vtrunc;
                      }
                    }
                  }
                  vmin = wave_min_dpp(min_raw(act[0] ? out[0] : inf, act[1] ? out[1] : inf));
                }
#pragma unroll
                for (int c = 0; c < 2; ++c)
                  if (act[c]) hcur[j * k2W + kk[c]] = out[c] - vmin;
                if (BACKWARD && lane == 0) sc[j] = vmin;
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
          int *stni = (int *)(stn + k2StI);
          stni[lane] = w;
          const int ntot = nx.nout + nx.nin;
          double dk[2] = {0, 0}, mv[8][2], qv[8][2], qpv[8][2];
#pragma unroll
          for (int c = 0; c < 2; ++c)
            if (act[c]) dk[c] = p.unary[(size_t)nx.node * K + kk[c]];
#pragma unroll
          for (int j = 0; j < 8; ++j) {
#pragma unroll
            for (int c = 0; c < 2; ++c) {
              mv[j][c] = 0; qv[j][c] = 0; qpv[j][c] = 0;
              if (j < ntot && act[c]) {
                const size_t off = (size_t)nx.e[j] * K + kk[c];
                if (j < nx.nout && (UPDATE || PRIMAL)) mv[j][c] = p.msg[off];
                if (!SHARED) { qv[j][c] = p.q[off]; qpv[j][c] = p.qprim[off]; }
              }
            }
          }
          double av = 0;
          int pxv = 0, xn = 0, sl = 0;
          if (lane < ntot) {
            int ej = 0;
#pragma unroll
            for (int j = 0; j < 8; ++j)
              if (lane == j) { ej = nx.e[j]; xn = nx.xn[j]; sl = nx.slot[j]; }
            av = p.alpha[ej];
          }
          if (nx.ndep > 0) {
            int myrank = nx.dep[0];
#pragma unroll
            for (int j = 1; j < 4; ++j)
              if (lane == j) myrank = nx.dep[j];
            const bool watching = lane < nx.ndep;
            int spins = 0;
            bool ok = true;
            for (;;) {
              const int v = watching ? ld_sc1(p.done + myrank) : epoch;
              if (!UNI(v < epoch)) break;
              __builtin_amdgcn_s_sleep(1);
              if (++spins > kSpinLimit || ((spins & 1023) == 0 && ld_sc1(p.abort_flag))) { ok = false; break; }
            }
            if (!ok && lane == 0) { st_sc1(p.abort_flag, 1); ctl[1] = 1; }
          }
#pragma unroll
          for (int j = 0; j < 8; ++j) {
            if (UPDATE && j >= nx.nout && j < ntot && nx.slot[j] < 0) {
#pragma unroll
              for (int c = 0; c < 2; ++c)
                if (act[c]) mv[j][c] = ld_sc1(p.msg + (size_t)nx.e[j] * K + kk[c]);
            }
          }
          if (PRIMAL && lane < ntot && lane >= nx.nout && sl < 0) pxv = ld_sc1(p.x + xn);
#pragma unroll
          for (int c = 0; c < 2; ++c) {
            if (act[c]) {
              stn[k2StD + kk[c]] = dk[c];
#pragma unroll
              for (int j = 0; j < 8; ++j) {
                if (j < ntot) {
                  stn[k2StM + j * k2W + kk[c]] = mv[j][c];
                  if (!SHARED) { stn[k2StQ + j * k2W + kk[c]] = qv[j][c]; stn[k2StQP + j * k2W + kk[c]] = qpv[j][c]; }
                }
              }
            }
          }
          if (lane < 8) { stn[k2StA + lane] = av; stni[64 + lane] = pxv; }
        }
      } else if (wave == kPipeCompute + 1) {
        // ------------------------------------------------------------ storer: node pos - 1
        if (pos - 1 >= p0) {
          const NodeDesc pd = decode_desc(desc[(size_t)(pos - 1) * DW + lane]);
          const double *scp = scal + ((pos + 1) & 1) * kScalDoubles;
          if (UPDATE) {
#pragma unroll
            for (int j = 0; j < 8; ++j) {
              if (j < pd.nout) {
                double *mb = ((pd.remote >> j) & 1) ? (((pd.remote >> (8 + j)) & 1) ? p.peer_msg1 : p.peer_msg0) : p.msg;
#pragma unroll
                for (int c = 0; c < 2; ++c)
                  if (act[c]) st_sc1(mb + (size_t)pd.e[j] * K + kk[c], hprev[j * k2W + kk[c]]);
                if (BACKWARD && lane == 0) p.lbterms[pd.lbe[j]] = scp[j];
              }
            }
            if (BACKWARD && lane == 0) p.lbterms[pd.lbn] = scp[8];
          }
          if (PRIMAL && lane == 0) {
            const int xi = ((const int *)(scp + 10))[0];
            st_sc1(p.x + pd.node, xi);
            if (pd.remote & (1 << 16)) st_sc1(p.peer_x0 + pd.node, xi);
            if (pd.remote & (1 << 17)) st_sc1(p.peer_x1 + pd.node, xi);
            p.eterms[pd.epos] = scp[9];
          }
          asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
          if (lane == 0) {
            st_sc1(p.done + pd.rank, epoch);
            if (pd.remote & (1 << 16)) st_sc1(p.peer_done0 + pd.rank, epoch);
            if (pd.remote & (1 << 17)) st_sc1(p.peer_done1 + pd.rank, epoch);
          }
        }
      } else if (wave == kPipeCompute + 3) {
        // ------------------------------------------------------------ primal of node pos
        if (PRIMAL && have_node) {
          const int *sti = (const int *)(st + k2StI);
          const int f = __builtin_amdgcn_readfirstlane(sti[2]);
          const int nout = f & 15, nin = (f >> 4) & 15, md = (f >> 16) & 255, ntot = nout + nin;
          double db[2] = {act[0] ? st[k2StD + kk[0]] : inf, act[1] ? st[k2StD + kk[1]] : inf};
#pragma unroll
          for (int j = 0; j < 8; ++j) {
            if (j >= nout && j < ntot) {
              const int sl = __builtin_amdgcn_readfirstlane(sti[12 + j]);
              const int ks = sl >= 8 ? xprev2 : sl >= 0 ? xprev : __builtin_amdgcn_readfirstlane(sti[64 + j]);
              const int mdj = (md >> j) & 1;
              const double aj = st[k2StA + j];
#pragma unroll
              for (int c = 0; c < 2; ++c) {
                if (act[c]) {
                  double d;
                  if (SHARED) {
                    const double pks = p.pos[ks];
                    d = mdj == 0 ? pks - posk[c] : posk[c] - pks;
                  } else {
                    const double qvk = st[k2StQ + j * k2W + kk[c]], qpk = st[k2StQP + j * k2W + kk[c]];
                    d = mdj == 0 ? st[k2StQP + j * k2W + ks] - qvk : qpk - st[k2StQ + j * k2W + ks];
                  }
                  const double v = fabs(d);
                  db[c] += aj * (v < p.lambda ? v : p.lambda);
                }
              }
            }
          }
          double di[2] = {db[0], db[1]};
#pragma unroll
          for (int j = 0; j < 8; ++j) {
            if (j < nout) {
#pragma unroll
              for (int c = 0; c < 2; ++c)
                if (act[c]) di[c] += st[k2StM + j * k2W + kk[c]];
            }
          }
          // first minimum: per lane label lane before lane + 64 (strict '<'), then across lanes by (value, index)
          const bool second = act[1] && di[1] < di[0];
          const double bestv = second ? di[1] : di[0];
          const int besti = second ? kk[1] : kk[0];
          const double bestdb = second ? db[1] : db[0];
          const int bi = wave_argmin_dpp(bestv, besti);
          const double eb = readlane_f64(bestdb, bi & (kWave - 1));
          xprev2 = xprev; xprev = bi;
          if (lane == 0) { sc[9] = eb; ((int *)(sc + 10))[0] = bi; }
        }
      }
      __syncthreads();
      if (ctl[1]) {
        if (tid == 0) st_sc1(p.abort_flag, 1);
        return;
      }
    }
    if (p.timeline && tid == 0) p.timeline[((size_t)D * p.nruns[0] + run) * 2 + 1] = wall_clock64();
  }
}

// ---- wide-label pipelined sweep: 64 < K <= 256, shared strictly ascending positions -------
// The regime of the large grids (3000x2000x256).  A lane holds four labels (k = c * 64 + lane).
// Per message K^2 pair costs are too many, so
//  * min-plus only looks at the sources inside the truncation window of each destination
//    (a source farther than lambda costs >= vTrunc exactly, by monotone rounding);
//  * the certificate's tangency test ("no two u = h - alpha q, nor two v = h + alpha q, within
//    delta") is a closest-pair test by bucketing -- conservative (all pairs, not only those
//    with a useful cone), O(K) instead of O(K^2).
// Three compute waves per outgoing message, each forming Di and H = gamma Di - m itself (cheap,
// and no barrier between them): wave 3j does the windowed min-plus, waves 3j+1 / 3j+2 the u / v
// closest-pair tests and post their verdicts in LDS; wave 3j waits for the two verdicts,
// normalises and hands over.  Loader / storer / primal waves as in trws_pipe_kernel (the loader is
// split in two: data nobody else writes, and data behind completion flags, so that the two HBM
// round trips of a visit overlap); one
// hardware barrier per visit.  Nodes with more than four outgoing messages take a second round
// (message j + 4 on the same waves).  If the certificate fails, wave 3j runs the reference's
// serial envelope construction in LDS (one at a time per workgroup: shared scratch, rare).
// Kernel 1 (truncated linear) only; kernel 2 above K = 64 stays on the generic kernel.
constexpr int kWideCompute = 12;
constexpr int kWideWaves = kWideCompute + 4;  // + loader (own data), loader (foreign data), storer, primal
constexpr int kWideThreads = kWideWaves * kWave;
constexpr int kWS = 260;    // LDS row stride in doubles (>= 256 + 1 breakpoints, multiple of 4)
constexpr int kWPad = 16;   // min-plus source table is padded by this many (+inf, 0) entries on both sides
constexpr int kWScr = 2 * (256 + 2 * kWPad);  // per compute wave scratch: (h, q) source table | 256 keys + 516 ints
constexpr int kWBuckets = 512;
constexpr int kWStI = kWS + 8 * kWS + 8;            // int area of a stage (in doubles)
constexpr int kWStage = kWStI + 36;
// stage: D[kWS] m[8][kWS] a[8] | ints desc[64] px[8]

struct WidePtrs {
  double *stage0, *hand, *scr, *fb, *pos, *scal, *msc;
  int *dring, *flags, *hflag, *ctl;
};
__device__ __forceinline__ WidePtrs wide_carve(double *lds) {
  WidePtrs w;
  w.stage0 = lds;                            // 2 * kWStage
  w.hand = w.stage0 + 2 * kWStage;           // 3 * 8 * kWS : new messages of the last three visits
  w.scr = w.hand + 3 * 8 * kWS;              // kWideCompute * kWScr
  w.fb = w.scr + kWideCompute * kWScr;       // 4 * kWS : sources, stack, breakpoints of the serial construction
  w.pos = w.fb + 4 * kWS;                    // kWS
  w.scal = w.pos + kWS;                      // 2 * kScalDoubles
  w.msc = w.scal + 2 * kScalDoubles;         // [8][2]: hmin, hmax of H_j for the closest-pair waves
  w.dring = (int *)(w.msc + 16);             // 3 * 64 descriptor words (for the storer)
  w.flags = w.dring + 3 * 64;                // [8][2] verdicts of the closest-pair waves
  w.hflag = w.flags + 16;                    // [8] "H_j is in the min-plus wave's table" (visit token)
  w.ctl = w.hflag + 8;                       // [0] run, [1] abort, [2] lock of the serial scratch
  return w;
}
constexpr int kWideLdsDoubles = 2 * kWStage + 3 * 8 * kWS + kWideCompute * kWScr + 4 * kWS + kWS + 2 * kScalDoubles + 16 + 96 + 8 + 4 + 2;
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

template <int KERNEL, bool BACKWARD, bool PRIMAL, bool UPDATE>
__device__ __forceinline__ void wide_body(DevParams p, int epoch) {
  extern __shared__ __attribute__((aligned(16))) double lds[];
  const WidePtrs L = wide_carve(lds);
  const int K = p.K;
  const int C = (K + kWave - 1) / kWave;  // 64-label chunks
  const double inf = __builtin_huge_val();
  const int tid = threadIdx.x, lane = tid & (kWave - 1);
  const int wave = __builtin_amdgcn_readfirstlane(tid / kWave);
  constexpr int D = BACKWARD ? 1 : 0;
  constexpr int DW = TrwsGraph::kDescWords;
  const int32_t *desc = p.desc[D];
  for (int k = tid; k < kWS; k += kWideThreads) L.pos[k] = k < K ? p.pos[k] : inf;
  // positions on an exact arithmetic progression (checked on the host: pos[k+d] - pos[k] == d * step
  // bit for bit): the min-plus source table then holds h only and alpha |d step| is formed once per d
  const double ustep = p.uniform_step;
  const bool uniform = ustep != 0;
  if (wave < kWideCompute && wave % 3 == 0 && lane < 2 * kWPad) {
    // padding of the min-plus source tables, never overwritten afterwards
    if (uniform) (L.scr + wave * kWScr)[lane < kWPad ? lane : K + lane] = inf;
    else ((double2 *)(L.scr + wave * kWScr))[lane < kWPad ? lane : K + lane] = make_double2(inf, 0.0);
  }
  if (tid < 16) L.flags[tid] = -1;
  if (tid < 8) L.hflag[tid] = -1;
  if (tid == 0) { L.ctl[1] = 0; L.ctl[2] = 0; }
  double posr[4];  // this lane's four label positions
#pragma unroll
  for (int c = 0; c < 4; ++c) posr[c] = c * kWave + lane < K ? p.pos[c * kWave + lane] : inf;
  const double pos_first = p.pos[0], pos_last = p.pos[K - 1];
  // development profile (STEREO_HIP_TRWS_PROF): cycles of wave 0 per phase [0..15], busy cycles of
  // loader / storer / primal [16..18], hardware-barrier wait of wave 0 [19], visits [20]
  unsigned long long pacc[16] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, pbusy = 0, pwait = 0, pvis = 0;
#define WSTAMP(i) do { if (p.prof) { const long long now_ = (long long)__builtin_readcyclecounter(); pacc[i] += (unsigned long long)(now_ - tmark); tmark = now_; } } while (0)
  __syncthreads();

  for (;;) {
    if (tid == 0) { const int t_ = atomicAdd(p.ticket, 1); L.ctl[0] = t_ < p.ntickets[D] ? (p.run_order[D] ? p.run_order[D][t_] : t_) : p.nruns[D]; }
    __syncthreads();
    const int run = __builtin_amdgcn_readfirstlane(L.ctl[0]);
    __syncthreads();
    if (run >= p.nruns[D]) break;
    const int p0 = p.run_ptr[D][run], p1 = p.run_ptr[D][run + 1];
    if (p.timeline && tid == 0) p.timeline[((size_t)D * p.nruns[0] + run) * 2] = wall_clock64();

    // One visit loop per role (not one loop with a role switch inside): state carried from visit
    // to visit -- the loader's parked registers -- then occupies registers in that role only.
#define WIDE_VISITS_BEGIN     for (int pos = p0 - 1; pos <= p1; ++pos) { \
      double *st = L.stage0 + (pos & 1) * kWStage; \
      double *stn = L.stage0 + ((pos + 1) & 1) * kWStage; \
      const int hb = ((pos % 3) + 3) % 3, hb1 = (((pos - 1) % 3) + 3) % 3, hb2 = (((pos - 2) % 3) + 3) % 3; \
      double *hcur = L.hand + hb * 8 * kWS, *hprev = L.hand + hb1 * 8 * kWS, *hprev2 = L.hand + hb2 * 8 * kWS; \
      double *sc = L.scal + (pos & 1) * kScalDoubles; \
      const bool have_node = pos >= p0 && pos < p1; \
      long long tmark = p.prof ? (long long)__builtin_readcyclecounter() : 0; \
      const long long tvisit = tmark; \
      (void)st; (void)stn; (void)hcur; (void)hprev; (void)hprev2; (void)sc; (void)have_node; (void)tvisit;
#define WIDE_VISITS_END_(BARRIER)       if (p.prof) { \
        const long long now_ = (long long)__builtin_readcyclecounter(); \
        if (wave == 0) pvis += have_node ? 1 : 0; \
        pbusy += (unsigned long long)(now_ - tvisit); \
        tmark = now_; \
      } \
      BARRIER; \
      if (p.prof && wave == 0) pwait += (unsigned long long)((long long)__builtin_readcyclecounter() - tmark); \
      if (L.ctl[1]) { \
        if (tid == 0) st_sc1(p.abort_flag, 1); \
        return; \
      } \
    }
#define WIDE_VISITS_END WIDE_VISITS_END_(__syncthreads())
    if (wave < kWideCompute) {
      WIDE_VISITS_BEGIN
        // ======================================================== compute waves
        const int j0 = wave / 3, role = wave - 3 * j0;  // role 0: min-plus, 1: u test, 2: v test
        if (UPDATE && have_node) {
          const int *sti = (const int *)(st + kWStI);
          const int f = __builtin_amdgcn_readfirstlane(sti[2]);
          const int nout = f & 15, nin = (f >> 4) & 15, ntot = nout + nin;
          const bool fast_msg = KERNEL == 1 && p.certificate != 0;
          const bool working = j0 < nout && (role == 0 || fast_msg);
          if (working || (BACKWARD && wave == 0)) {
            bool valid[4];
#pragma unroll
            for (int c = 0; c < 4; ++c) valid[c] = c * kWave + lane < K;
            double di[4] = {inf, inf, inf, inf};
            if (role == 0) {
            const unsigned slA = (unsigned)__builtin_amdgcn_readfirstlane(sti[41]);
            const unsigned slB = (unsigned)__builtin_amdgcn_readfirstlane(sti[42]);
            // Di = D + messages in list order (from the ring where the neighbour was one of
            // the last two visits of this run), formed by the min-plus wave of each message
            // (reads are unconditional -- rows are padded to 256 -- and masked afterwards, so that
            // all of them are in flight together)
#pragma unroll
            for (int c = 0; c < 4; ++c) di[c] = st[c * kWave + lane];
#pragma unroll
            for (int jj = 0; jj < 8; ++jj) {
              if (jj < ntot) {
                const int sl = jj >= nout ? (int)(signed char)(((jj < 4 ? slA : slB) >> (8 * (jj & 3))) & 255) : -1;
                const double *src = sl >= 8 ? hprev2 + (sl - 8) * kWS : sl >= 0 ? hprev + sl * kWS : st + kWS + jj * kWS;
#pragma unroll
                for (int c = 0; c < 4; ++c) di[c] += src[c * kWave + lane];
              }
            }
#pragma unroll
            for (int c = 0; c < 4; ++c) di[c] = valid[c] ? di[c] : inf;
            if (BACKWARD) {
              const double dm = min_raw(min_raw(di[0], di[1]), min_raw(di[2], di[3]));
              const double node_vmin = wave_min_dpp(dm);
              if (wave == 0 && lane == 0) sc[8] = node_vmin;
#pragma unroll
              for (int c = 0; c < 4; ++c) di[c] -= node_vmin;
            }
            }
            WSTAMP(0);
            for (int j = j0; working && j < nout; j += 4) {
              const double gamma = (double)1 / (double)(nout > nin ? nout : nin);
              const double alpha = st[kWS + 8 * kWS + j];
              const bool constant = UNI(alpha == 0);
              double h[4] = {inf, inf, inf, inf}, hmin = 0, hmax = 0;
              double2 *mtab = (double2 *)(L.scr + (wave - role) * kWScr) + kWPad;  // the min-plus wave's (h, q) table
              double *htab = L.scr + (wave - role) * kWScr + kWPad;                // ... or h only (uniform positions)
              if (role == 0) {
                double hlo = inf, hhi = -inf;
#pragma unroll
                for (int c = 0; c < 4; ++c) h[c] = gamma * di[c] - st[kWS + j * kWS + c * kWave + lane];
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                  hlo = min_raw(hlo, valid[c] ? h[c] : inf); hhi = max_raw(hhi, valid[c] ? h[c] : -inf);
                  h[c] = valid[c] ? h[c] : inf;
                }
                hmin = wave_min_dpp(hlo); hmax = wave_max_dpp(hhi);
                if (fast_msg && !constant) {
                  // publish H_j for the two closest-pair waves (and as this wave's source table)
#pragma unroll
                  for (int c = 0; c < 4; ++c) {
                    if (valid[c]) {
                      if (uniform) htab[c * kWave + lane] = h[c];
                      else mtab[c * kWave + lane] = make_double2(h[c], posr[c]);
                    }
                  }
                  if (lane == 0) { L.msc[2 * j] = hmin; L.msc[2 * j + 1] = hmax; }
                  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
                  if (lane == 0) __hip_atomic_store(L.hflag + j, pos, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                }
              } else if (!constant) {
                int spins = 0;
                while (__hip_atomic_load(L.hflag + j, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) != pos) {
                  __builtin_amdgcn_s_sleep(0);
                  if (++spins > kSpinLimit) { if (lane == 0) L.ctl[1] = 1; break; }  // bounded
                }
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
                double hv[4];
#pragma unroll
                for (int c = 0; c < 4; ++c) hv[c] = uniform ? htab[c * kWave + lane] : mtab[c * kWave + lane].x;
#pragma unroll
                for (int c = 0; c < 4; ++c) h[c] = valid[c] ? hv[c] : inf;
                hmin = L.msc[2 * j]; hmax = L.msc[2 * j + 1];
              }
              const double vtrunc = hmin + alpha * p.lambda;
              const double ap0 = alpha * pos_first, ap1 = alpha * pos_last;
              const double aplo = min_raw(ap0, ap1), aphi = max_raw(ap0, ap1);
              const double mag = max_raw(fabs(hmin), fabs(hmax)) + 2 * max_raw(fabs(ap0), fabs(ap1));
              const double delta = 1e-9 * (mag + fabs(alpha * p.lambda));
              double *scr = L.scr + wave * kWScr;
              WSTAMP(1);
              if (role != 0) {
                // ---- tangency: no two u = h - alpha q (role 1) / v = h + alpha q (role 2) within delta
                if (!constant) {
                  const double sgn = role == 2 ? 1.0 : -1.0;
                  double r[4];
#pragma unroll
                  for (int c = 0; c < 4; ++c) r[c] = h[c] + sgn * (alpha * posr[c]);
                  const double mn = role == 2 ? hmin + aplo : hmin - aphi;
                  const double mx = role == 2 ? hmax + aphi : hmax - aplo;
                  bool bad = !(delta < inf);
                  if (!bad) bad = keys_within(r, K, C, delta, mn, mx - mn, scr, lane);
                  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
                  if (lane == 0)
                    __hip_atomic_store(L.flags + 2 * j + (role - 1), (pos << 1) | (bad ? 1 : 0), __ATOMIC_RELAXED,
                                       __HIP_MEMORY_SCOPE_WORKGROUP);
                }
                WSTAMP(2);
              } else {
                double out[4] = {0, 0, 0, 0}, vmin = 0;
                if (constant) {
                  // typeStereoLinear.h:390-396: message = min H everywhere, normalised to zero
#pragma unroll
                  for (int c = 0; c < 4; ++c) out[c] = hmin;
                  vmin = hmin;
                } else {
                  // source table: (h, q) pairs at index kWPad + k, (+inf, 0) padding on both sides
                  // (written above, when H_j was published)
                  double2 *tab = mtab;
                  WSYNC();
                  bool serial = !fast_msg;
                  if (fast_msg) {
                    // ---- windowed min-plus: smallest and second smallest cost per destination
                    // (equal costs from two sources count as a zero margin: serial path decides)
                    double m1[4] = {inf, inf, inf, inf}, m2[4] = {inf, inf, inf, inf};
                    const int w = p.window;
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
                    } else if (w <= kWPad) {
                      for (int d = -w; d <= w; ++d) {
                        double2 sv[4];
#pragma unroll
                        for (int c = 0; c < 4; ++c) sv[c] = tab[c * kWave + lane + d];
#pragma unroll
                        for (int c = 0; c < 4; ++c) {
                          const double cst = pair_cost<1>(alpha, posr[c] - sv[c].y, sv[c].x);
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
                          sv[c] = tab[i < 0 ? 0 : i > K - 1 ? K - 1 : i];
                        }
#pragma unroll
                        for (int c = 0; c < 4; ++c) {
                          const int i = c * kWave + lane + d;
                          double cst = pair_cost<1>(alpha, posr[c] - sv[c].y, sv[c].x);
                          cst = (i >= 0 && i < K) ? cst : inf;
                          const double lo_ = min_raw(m1[c], cst), hi_ = max_raw(m1[c], cst);
                          m2[c] = min_raw(m2[c], hi_);
                          m1[c] = lo_;
                        }
                      }
                    }
                    bool bad = false;
                    double vloc = inf;
#pragma unroll
                    for (int c = 0; c < 4; ++c) {
                      if (valid[c]) {
                        bad = bad || (m1[c] < vtrunc && !(m2[c] - m1[c] > delta && vtrunc - m1[c] > delta));
                        out[c] = m1[c] < vtrunc ? m1[c] : vtrunc;
                        vloc = min_raw(vloc, out[c]);
                      }
                    }
                    vmin = wave_min_dpp(vloc);
                    serial = UNI(bad);
                    WSTAMP(3);
                    // the verdicts of the two closest-pair waves of this message
                    {
                      int spins = 0;
                      for (;;) {
                        const int v = lane < 2 ? __hip_atomic_load(L.flags + 2 * j + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) : (pos << 1);
                        const bool ready = (v >> 1) == pos;
                        if (!UNI(!ready)) { serial = serial || UNI((v & 1) != 0); break; }
                        __builtin_amdgcn_s_sleep(0);
                        if (++spins > kSpinLimit) { if (lane == 0) L.ctl[1] = 1; break; }  // bounded
                      }
                      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
                    }
                    WSTAMP(4);
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
                      if (valid[c]) Hs[c * kWave + lane] = h[c];
                    WSYNC();
                    if (lane == 0) build_envelope<KERNEL>(K, alpha, Hs, L.pos, sh, sq, z);
                    WSYNC();
                    double vloc = inf;
#pragma unroll
                    for (int c = 0; c < 4; ++c) {
                      const int k = c * kWave + lane;
                      if (c < C && k < K) {
                        int jj = 0;
                        while (z[jj + 1] < posr[c]) ++jj;
                        const double cst = pair_cost<KERNEL>(alpha, posr[c] - sq[jj], sh[jj]);
                        out[c] = cst < vtrunc ? cst : vtrunc;
                        vloc = min_raw(vloc, out[c]);
                      }
                    }
                    WSYNC();
                    if (lane == 0) __hip_atomic_store(L.ctl + 2, 0, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
                    vmin = wave_min_dpp(vloc);
                  }
                }
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                  const int k = c * kWave + lane;
                  if (c < C && k < K) hcur[j * kWS + k] = out[c] - vmin;
                }
                if (BACKWARD && lane == 0) sc[j] = vmin;
                WSTAMP(5);
              }
            }
          }
        }
      WIDE_VISITS_END
    } else if (wave == kWideCompute) {
      int wnext = desc[(size_t)p0 * DW + lane];
      WIDE_VISITS_BEGIN
        // ======================================================== loader A: node pos + 1, own data
        if (pos + 1 >= p0 && pos + 1 < p1) {
          const int w = wnext;
          if (pos + 2 < p1) wnext = desc[(size_t)(pos + 2) * DW + lane];
          const NodeDesc nx = decode_desc(w);
          int *stni = (int *)(stn + kWStI);
          stni[lane] = w;
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
#pragma unroll
              for (int j = 0; j < 8; ++j)
                if (j < nx.nout) stn[kWS + j * kWS + c * kWave + lane] = mv[j][c];
            }
          }
          if (lane < 8) stn[kWS + 8 * kWS + lane] = av;
        }
      WIDE_VISITS_END
    } else if (wave == kWideCompute + 1) {
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
          if (nx.ndep > 0) {
            int myrank = nx.dep[0];
#pragma unroll
            for (int j = 1; j < 4; ++j)
              if (lane == j) myrank = nx.dep[j];
            const bool watching = lane < nx.ndep;
            int spins = 0;
            bool ok = true;
            for (;;) {
              const int v = watching ? ld_sc1(p.done + myrank) : epoch;
              if (!UNI(v < epoch)) break;
              __builtin_amdgcn_s_sleep(1);
              if (++spins > kSpinLimit || ((spins & 1023) == 0 && ld_sc1(p.abort_flag))) { ok = false; break; }
            }
            if (!ok && lane == 0) { st_sc1(p.abort_flag, 1); L.ctl[1] = 1; }
          }
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
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                  const int k = c * kWave + lane;
                  if (c < C && k < K) st_sc1(mb + (size_t)pd.e[j] * K + k, hprev[j * kWS + k]);
                }
                if (BACKWARD && lane == 0) p.lbterms[pd.lbe[j]] = scp[j];
              }
            }
            if (BACKWARD && lane == 0) p.lbterms[pd.lbn] = scp[8];
          }
          if (PRIMAL && lane == 0) {
            const int xi = ((const int *)(scp + 10))[0];
            st_sc1(p.x + pd.node, xi);
            if (pd.remote & (1 << 16)) st_sc1(p.peer_x0 + pd.node, xi);
            if (pd.remote & (1 << 17)) st_sc1(p.peer_x1 + pd.node, xi);
            p.eterms[pd.epos] = scp[9];
          }
          asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
          if (lane == 0) {
            st_sc1(p.done + pd.rank, epoch);
            if (pd.remote & (1 << 16)) st_sc1(p.peer_done0 + pd.rank, epoch);
            if (pd.remote & (1 << 17)) st_sc1(p.peer_done1 + pd.rank, epoch);
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
          double db[4], di[4];
#pragma unroll
          for (int c = 0; c < 4; ++c) {
            const int k = c * kWave + lane;
            db[c] = (c < C && k < K) ? st[k] : inf;
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
                  const double d = fwd ? pks - posr[c] : posr[c] - pks;
                  const double v = KERNEL == 1 ? fabs(d) : d * d;
                  db[c] += aj * (v < p.lambda ? v : p.lambda);
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
          double bestv = inf, bestdb = 0;
          int besti = 0x7fffffff;
#pragma unroll
          for (int c = 0; c < 4; ++c) {  // ascending k per lane: strict '<' keeps the first minimum
            const int k = c * kWave + lane;
            if (c < C && k < K && di[c] < bestv) { bestv = di[c]; besti = k; bestdb = db[c]; }
          }
          const int bi = wave_argmin_dpp(bestv, besti);
          const double eb = readlane_f64(bestdb, bi & (kWave - 1));  // the lane owning label bi
          xprev2 = xprev; xprev = bi;
          if (lane == 0) { sc[9] = eb; ((int *)(sc + 10))[0] = bi; }
        }
            WIDE_VISITS_END
    }
#undef WIDE_VISITS_BEGIN
#undef WIDE_VISITS_END
#undef WIDE_VISITS_END_
    if (p.timeline && tid == 0) p.timeline[((size_t)D * p.nruns[0] + run) * 2 + 1] = wall_clock64();
  }
#undef WSTAMP
  if (p.prof && lane == 0) {
    if (wave == 0) {
      for (int i = 0; i < 16; ++i) atomicAdd(p.prof + i, pacc[i]);
      atomicAdd(p.prof + 21, pwait);
      atomicAdd(p.prof + 22, pvis);
    }
    if (wave >= kWideCompute) atomicAdd(p.prof + 16 + (wave - kWideCompute), pbusy);
    if (wave == 1) for (int i = 0; i < 3; ++i) atomicAdd(p.prof + 8 + i, pacc[i]);  // a closest-pair wave
  }
}
template <int KERNEL, bool BACKWARD, bool PRIMAL, bool UPDATE>
__global__ __launch_bounds__(kWideThreads) void trws_wide_kernel(DevParams p, int epoch) {
  wide_body<KERNEL, BACKWARD, PRIMAL, UPDATE>(p, epoch);
}
template <int KERNEL, bool BACKWARD, bool PRIMAL, bool UPDATE>
__global__ __launch_bounds__(kWideThreads) void trws_wide_group_kernel(GroupArgs ga, int epoch) {
  wide_body<KERNEL, BACKWARD, PRIMAL, UPDATE>(ga.pp[group_strip(ga)], epoch);
}
#undef WSYNC

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
template <int KERNEL>
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
    const double v = message_regs<KERNEL>(p, K, alpha[m], h, qs, qt, perm + (size_t)m * K, out, lane,
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
  DevBuf<int32_t> d_tail, d_order, d_fptr, d_fidx, d_bptr, d_bidx, d_lbn, d_lbe, d_levels, d_x;
  DevBuf<uint8_t> d_mdir;
  DevBuf<double> d_gamma, d_msg, d_lbterms, d_eterms;
  // persistent sweep schedule
  DevBuf<int32_t> d_run_order[2], d_chain_run_ptr[2], d_chain_run_order[2];
  DevBuf<int32_t> d_run_ptr[2], d_dep_ptr[2], d_dep_rank[2], d_done, d_ctl;  // d_ctl: [ticket, abort]
  DevBuf<int8_t> d_in_slot[2];
  DevBuf<int32_t> d_desc[2];
  bool fast = false;
  bool wide = false;  // 64 < K <= 256 with shared strictly ascending positions: trws_wide_kernel
  bool fast2 = false; // 64 < K <= 128, linear kernel, any positions: trws_pipe2_kernel (when not wide)
  bool wide_allowed = false;
  bool pos_ascending = false;  // shared positions finite and strictly ascending
  double pos_first = 0, pos_last = 0, pos_gap = 0;
  int window = 0;
  double uniform_step = 0;
  DevBuf<unsigned long long> d_fallbacks, d_prof, d_timeline;
  bool certificate = true;
  int epoch = 0;
  bool persistent = true;
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
  double *peer_msg[2] = {nullptr, nullptr};
  int32_t *peer_done[2] = {nullptr, nullptr}, *peer_x[2] = {nullptr, nullptr};
  bool need_peer[2] = {false, false};
  void *ipc_mapped[2][3] = {{nullptr, nullptr, nullptr}, {nullptr, nullptr, nullptr}};
  DevBuf<DevParams> d_group;         // parameters of the strips launched together with this one (first plan of a group)
  PinnedBuf<DevParams> h_group;
  hipStream_t own_stream = nullptr;  // strips launch concurrently: never on the NULL stream
  bool issued = false;
  int cus = 256;
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

size_t sweep_lds_bytes(int Kp) {
  return sizeof(double) * (size_t)(Kp + 8 + kWavesPerBlock * (kWaveVecs * Kp + 8));
}
size_t primal_lds_bytes(int Kp) { return sizeof(double) * (size_t)(2 * kWavesPerBlock + Kp); }
size_t persistent_lds_bytes(int Kp) {
  return sizeof(double) * (size_t)(Kp + 8 + kMaxSlots * Kp + Kp + kWavesPerBlock * (kWaveVecs * Kp + 8));
}

DevParams make_params(stereo_trws_plan *P) {
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
    p.ntickets[d] = P->nstrips > 1 ? P->ntickets[d] : p.nruns[d];
    if (P->nstrips > 1) p.run_order[d] = P->d_tickets[d].p;
  }
  p.peer_msg0 = P->peer_msg[0]; p.peer_msg1 = P->peer_msg[1];
  p.peer_done0 = P->peer_done[0]; p.peer_done1 = P->peer_done[1];
  p.peer_x0 = P->peer_x[0]; p.peer_x1 = P->peer_x[1];
  p.done = P->d_done.p; p.ticket = P->d_ctl.p; p.abort_flag = P->d_ctl.p + 1; p.N = (int)P->N;
  p.fallbacks = P->d_fallbacks.p; p.certificate = P->certificate ? 1 : 0;
  p.prof = P->d_prof.p;
  p.timeline = P->d_timeline.p;
  p.desc[0] = P->d_desc[0].p; p.desc[1] = P->d_desc[1].p;
  p.prof_run = -1;
  p.window = P->window;
  p.uniform_step = P->uniform_step;
  p.pos_first = P->pos_first; p.pos_last = P->pos_last;
  p.debug = 0;
  if (const char *dbg = std::getenv("STEREO_HIP_TRWS_DEBUG")) p.debug = std::atoi(dbg);
  p.win_ok = (P->pos_ascending && P->window <= 16 && !(p.debug & 256)) ? 1 : 0;
  p.pos_gap = P->pos_gap;
  if (const char *pr = std::getenv("STEREO_HIP_TRWS_PROF_RUN")) p.prof_run = std::atoi(pr);
  return p;
}

// One persistent launch: 0 = forward, 1 = backward, 2 = forward + primal of the
// previous iteration, 3 = primal only.
template <int KERNEL, int MODE>
void launch_persistent(stereo_trws_plan *P, const DevParams &p, int what, hipStream_t s) {
  const size_t lds = persistent_lds_bytes(P->Kp);
  const int epoch = ++P->epoch;
  STEREO_HIP_CHECK(hipMemsetAsync(P->d_ctl.p, 0, sizeof(int32_t), s));  // ticket = 0
  const dim3 grid(P->grid_blocks), block(kBlock);
  if (P->wide) {
    const size_t wlds = sizeof(double) * kWideLdsDoubles;
    const dim3 wgrid(std::min(P->grid_blocks, P->cus)), wblock(kWideThreads);
    switch (what) {
      case 0: hipLaunchKernelGGL((trws_wide_kernel<1, false, false, true>), wgrid, wblock, wlds, s, p, epoch); break;
      case 1: hipLaunchKernelGGL((trws_wide_kernel<1, true, false, true>), wgrid, wblock, wlds, s, p, epoch); break;
      case 2: hipLaunchKernelGGL((trws_wide_kernel<1, false, true, true>), wgrid, wblock, wlds, s, p, epoch); break;
      default: hipLaunchKernelGGL((trws_wide_kernel<1, false, true, false>), wgrid, wblock, wlds, s, p, epoch); break;
    }
    STEREO_HIP_CHECK(hipGetLastError());
    if (what != 3) P->sweep_launches += 1;
    return;
  }
  if (P->fast2) {
    const size_t lds2 = sizeof(double) * k2LdsDoubles;
    const bool sh2 = P->pos != nullptr;
    const dim3 grid2(std::min(P->grid_blocks, P->cus)), block2(kPipeThreads);
#define PIPE2(BW, PR, UP)                                                                         \
  do {                                                                                            \
    if (sh2) hipLaunchKernelGGL((trws_pipe2_kernel<BW, PR, UP, true>), grid2, block2, lds2, s, p, epoch);  \
    else hipLaunchKernelGGL((trws_pipe2_kernel<BW, PR, UP, false>), grid2, block2, lds2, s, p, epoch);    \
  } while (0)
    switch (what) {
      case 0: PIPE2(false, false, true); break;
      case 1: PIPE2(true, false, true); break;
      case 2: PIPE2(false, true, true); break;
      default: PIPE2(false, true, false); break;
    }
#undef PIPE2
    STEREO_HIP_CHECK(hipGetLastError());
    if (what != 3) P->sweep_launches += 1;
    return;
  }
  if (P->fast) {
    const size_t plds = sizeof(double) * (2 * kStageDoubles + 4 * 8 * kWave + 2 * kScalDoubles + kPipeCompute * kPipeTab + 2 + 3 * kWave / 2);
    const bool sh = P->pos != nullptr;
    const dim3 pblock(kPipeThreads);
#define PIPE(BW, PR, UP)                                                                          \
  do {                                                                                            \
    if (sh) hipLaunchKernelGGL((trws_pipe_kernel<KERNEL, BW, PR, UP, true>), grid, pblock, plds, s, p, epoch); \
    else hipLaunchKernelGGL((trws_pipe_kernel<KERNEL, BW, PR, UP, false>), grid, pblock, plds, s, p, epoch);   \
  } while (0)
    switch (what) {
      case 0: PIPE(false, false, true); break;
      case 1: PIPE(true, false, true); break;
      case 2: PIPE(false, true, true); break;
      default: PIPE(false, true, false); break;
    }
#undef PIPE
    STEREO_HIP_CHECK(hipGetLastError());
    if (what != 3) P->sweep_launches += 1;
    return;
  }
  switch (what) {
    case 0: hipLaunchKernelGGL((trws_persistent_kernel<KERNEL, false, MODE, false, true>), grid, block, lds, s, p, epoch); break;
    case 1: hipLaunchKernelGGL((trws_persistent_kernel<KERNEL, true, MODE, false, true>), grid, block, lds, s, p, epoch); break;
    case 2: hipLaunchKernelGGL((trws_persistent_kernel<KERNEL, false, MODE, true, true>), grid, block, lds, s, p, epoch); break;
    default: hipLaunchKernelGGL((trws_persistent_kernel<KERNEL, false, MODE, true, false>), grid, block, lds, s, p, epoch); break;
  }
  STEREO_HIP_CHECK(hipGetLastError());
  if (what != 3) P->sweep_launches += 1;
}

template <int KERNEL, int MODE>
void persistent_iteration(stereo_trws_plan *P, const DevParams &p, hipStream_t s) {
  if (P->time_sweeps) STEREO_HIP_CHECK(hipEventRecord(P->ev0, s));
  if (!P->fwd_pending) launch_persistent<KERNEL, MODE>(P, p, 0, s);
  launch_persistent<KERNEL, MODE>(P, p, 1, s);
  // the backward sweep's lower-bound terms travel while the next launch runs
  STEREO_HIP_CHECK(hipEventRecord(P->ev_bwd, s));
  STEREO_HIP_CHECK(hipStreamWaitEvent(P->copy_stream, P->ev_bwd, 0));
  STEREO_HIP_CHECK(hipMemcpyAsync(P->h_lb.p, P->d_lbterms.p, sizeof(double) * P->n_lb, hipMemcpyDeviceToHost,
                                  P->copy_stream));
  STEREO_HIP_CHECK(hipEventRecord(P->ev_lb, P->copy_stream));
  P->lb_in_flight = true;
  // forward sweep of the NEXT iteration fused with this iteration's primal
  launch_persistent<KERNEL, MODE>(P, p, 2, s);
  P->fwd_pending = true;
  if (P->time_sweeps) STEREO_HIP_CHECK(hipEventRecord(P->ev1, s));
}

template <int KERNEL, int MODE>
void launch_iteration(stereo_trws_plan *P, const DevParams &p, hipStream_t s) {
  const TrwsGraph &g = *P->graph;
  const int L = (int)g.level_ptr.size() - 1;
  const size_t lds = sweep_lds_bytes(P->Kp), plds = primal_lds_bytes(P->Kp);
  if (P->time_sweeps) STEREO_HIP_CHECK(hipEventRecord(P->ev0, s));
  for (int l = 0; l < L; ++l) {
    const int cnt = g.level_ptr[l + 1] - g.level_ptr[l];
    hipLaunchKernelGGL((trws_sweep_kernel<KERNEL, false, MODE>), dim3(cnt), dim3(kBlock), lds, s, p,
                       P->d_levels.p + g.level_ptr[l]);
  }
  for (int l = L - 1; l >= 0; --l) {
    const int cnt = g.level_ptr[l + 1] - g.level_ptr[l];
    hipLaunchKernelGGL((trws_sweep_kernel<KERNEL, true, MODE>), dim3(cnt), dim3(kBlock), lds, s, p,
                       P->d_levels.p + g.level_ptr[l]);
  }
  if (P->time_sweeps) STEREO_HIP_CHECK(hipEventRecord(P->ev1, s));
  for (int l = 0; l < L; ++l) {
    const int cnt = g.level_ptr[l + 1] - g.level_ptr[l];
    hipLaunchKernelGGL((trws_primal_kernel<KERNEL>), dim3(cnt), dim3(kBlock), plds, s, p,
                       P->d_levels.p + g.level_ptr[l]);
  }
  STEREO_HIP_CHECK(hipGetLastError());
  P->sweep_launches += 2 * L;
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
  STEREO_HIP_CHECK(hipMemset(P->d_msg.p, 0, sizeof(double) * (size_t)P->E * P->K));
  STEREO_HIP_CHECK(hipMemset(P->d_x.p, 0, sizeof(int32_t) * P->N));
  STEREO_HIP_CHECK(hipMemset(P->d_done.p, 0, sizeof(int32_t) * P->N));
  STEREO_HIP_CHECK(hipMemset(P->d_ctl.p, 0, sizeof(int32_t) * 2));
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
    P->d_perm_q.alloc((size_t)P->E * P->K);
    P->d_perm_qp.alloc((size_t)P->E * P->K);
    run_argsort(P->q, P->d_perm_q.p, P->K, P->E, nullptr);
    run_argsort(P->qprim, P->d_perm_qp.p, P->K, P->E, nullptr);
    fix_equal_positions(P->q, P->d_perm_q.p, P->K, P->E);
    fix_equal_positions(P->qprim, P->d_perm_qp.p, P->K, P->E);
  }
  STEREO_HIP_CHECK(hipDeviceSynchronize());
  // shared positions that are finite and strictly ascending: truncation window in index steps
  // (windowed min-plus of the pipelined kernel's flat-h path; the wide-label kernel requires it)
  P->wide = false; P->uniform_step = 0; P->pos_ascending = false; P->window = 0;
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
      }
    }
  }
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
    const bool wide_candidate = kernel == 1 && K > kWave && K <= 256 && message_mode == STEREO_TRWS_MESSAGES_EXACT;
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
      static struct { int64_t N = -1, E = -1, capacity = -1, cus = -1; int nstrips = 1, ordering = 0; std::vector<uint32_t> conn; std::vector<int32_t> owner;
                      std::shared_ptr<const TrwsGraph> g; } cache;
      std::lock_guard<std::mutex> lock(cache_mutex);
      const bool hit = cache.g && cache.N == N && cache.E == E && cache.capacity == capacity && cache.cus == P->cus &&
                       cache.nstrips == nstrips && cache.ordering == ordering &&
                       std::memcmp(cache.conn.data(), conn, sizeof(uint32_t) * 2 * (size_t)E) == 0 &&
                       (nstrips == 1 || std::memcmp(cache.owner.data(), owner, sizeof(int32_t) * (size_t)N) == 0);
      if (hit) {
        P->graph = cache.g;
      } else {
        auto fresh = std::make_shared<TrwsGraph>();
        if (!build_trws_graph(N, E, conn, *fresh, gerr, capacity, nstrips > 1 ? owner : nullptr, nstrips, P->cus, ordering)) return fail(gerr, err, errcap);
        P->graph = fresh;
        if (N <= (1 << 20)) {  // (the descriptors of a 3000 x 2000 grid are 3 GB: not worth keeping)
          cache.N = N; cache.E = E; cache.capacity = capacity; cache.cus = P->cus; cache.nstrips = nstrips; cache.ordering = ordering;
          cache.conn.assign(conn, conn + 2 * (size_t)E); cache.g = fresh;
          if (nstrips > 1) cache.owner.assign(owner, owner + N); else cache.owner.clear();
        } else {
          cache.g.reset(); cache.conn.clear(); cache.owner.clear(); cache.N = -1;
        }
      }
    }
    const TrwsGraph &g = *P->graph;
    if (sweep_lds_bytes(P->Kp) > 160 * 1024) return fail("stereo_trws: K too large for LDS", err, errcap);
    P->d_tail.upload(g.tail.data(), g.tail.size());
    P->d_order.upload(g.order.data(), g.order.size());
    P->d_fptr.upload(g.fptr.data(), g.fptr.size());
    P->d_fidx.upload(g.fidx.data(), g.fidx.size());
    P->d_bptr.upload(g.bptr.data(), g.bptr.size());
    P->d_bidx.upload(g.bidx.data(), g.bidx.size());
    P->d_lbn.upload(g.lb_pos_node.data(), g.lb_pos_node.size());
    P->d_lbe.upload(g.lb_pos_edge.data(), g.lb_pos_edge.size());
    P->d_levels.upload(g.level_ranks.data(), g.level_ranks.size());
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
    P->fast = g.fast_ok && K <= kWave && message_mode == STEREO_TRWS_MESSAGES_EXACT;
    P->wide_allowed = g.fast_ok && kernel == 1 && K > kWave && K <= 256 && message_mode == STEREO_TRWS_MESSAGES_EXACT;
    P->fast2 = g.fast_ok && kernel == 1 && K > kWave && K <= 2 * kWave && message_mode == STEREO_TRWS_MESSAGES_EXACT;
    if (const char *f = std::getenv("STEREO_HIP_TRWS_FAST")) {
      P->fast = P->fast && std::string(f) != "0";
      P->wide_allowed = P->wide_allowed && std::string(f) != "0";
      P->fast2 = P->fast2 && std::string(f) != "0";
    }
    if (nstrips > 1) {
      // a strip walks the chain schedule with one of the descriptor-driven kernels
      if (!(P->fast || P->wide_allowed || P->fast2))
        return fail("stereo_trws: row strips need a graph and label count the pipelined kernels take "
                    "(<= 8 edges per node; K <= 64, or K <= 128 with the linear kernel, or K <= 256 with shared positions)", err, errcap);
      for (int d = 0; d < 2; ++d) {
        const TrwsGraph::Sweep &S = g.sweep[d];
        const int64_t R = (int64_t)S.chain_run_ptr.size() - 1;
        std::vector<int32_t> mine;
        for (int64_t t = 0; t < R; ++t) {
          const int32_t run = S.chain_run_order.empty() ? (int32_t)t : S.chain_run_order[t];
          if (S.chain_run_strip[run] == strip) mine.push_back(run);
        }
        P->ntickets[d] = (int)mine.size();
        P->d_tickets[d].upload(mine.data(), mine.size());
        // which neighbours this strip writes to (it must be connected to them before it iterates)
        for (int64_t q = 0; q < N; ++q) {
          const uint32_t rem = (uint32_t)S.desc[(size_t)q * TrwsGraph::kDescWords + kDescRemote];
          if (g.owner[g.order[S.chain_rank[q]]] != strip) continue;
          if (rem & (1u << 16)) P->need_peer[0] = true;
          if (rem & (1u << 17)) P->need_peer[1] = true;
        }
      }
    }
    if (strip_api) STEREO_HIP_CHECK(hipStreamCreateWithFlags(&P->own_stream, hipStreamNonBlocking));
    P->n_lb = nstrips > 1 ? g.strip_lb_terms[strip] : g.lb_terms;
    P->n_en = nstrips > 1 ? g.strip_nodes[strip] : N;
    P->d_done.alloc(N);
    P->d_ctl.alloc(2);
    P->d_fallbacks.alloc(1);
    STEREO_HIP_CHECK(hipMemset(P->d_fallbacks.p, 0, sizeof(unsigned long long)));
    if (const char *c = std::getenv("STEREO_HIP_TRWS_CERTIFICATE")) P->certificate = std::string(c) != "0";
    if (std::getenv("STEREO_HIP_TRWS_PROF")) { P->d_prof.alloc(32); STEREO_HIP_CHECK(hipMemset(P->d_prof.p, 0, 256)); }
    if (std::getenv("STEREO_HIP_TRWS_TIMELINE"))
      P->d_timeline.alloc(4 * std::max({g.sweep[0].run_ptr.size(), g.sweep[0].chain_run_ptr.size(), g.sweep[1].chain_run_ptr.size()}) + 4);
    STEREO_HIP_CHECK(hipMemset(P->d_done.p, 0, sizeof(int32_t) * N));
    STEREO_HIP_CHECK(hipMemset(P->d_ctl.p, 0, sizeof(int32_t) * 2));
    if (const char *sc = std::getenv("STEREO_HIP_TRWS_SCHEDULE")) P->persistent = std::string(sc) != "levels";
    {
      // one workgroup per concurrently active run, capped by what stays resident
      int64_t runs = std::max<int64_t>((int64_t)g.sweep[0].run_ptr.size() - 1, 1);
      if (g.fast_ok)
        runs = std::max<int64_t>({runs, (int64_t)g.sweep[0].chain_run_ptr.size() - 1, (int64_t)g.sweep[1].chain_run_ptr.size() - 1});
      if (nstrips > 1) runs = std::max<int64_t>({1, (int64_t)P->ntickets[0], (int64_t)P->ntickets[1]});
      P->grid_blocks = (int)std::min<int64_t>(runs, P->cus * per_cu);
      if (max_blocks > 0) P->grid_blocks = std::min(P->grid_blocks, max_blocks);
    }
    P->d_msg.alloc((size_t)E * K);
    P->d_lbterms.alloc(P->n_lb);
    P->d_eterms.alloc(P->n_en);
    P->d_x.alloc(N);
    P->h_lb.alloc(P->n_lb); P->h_en.alloc(P->n_en); P->h_x.alloc(N); P->h_ctl.alloc(2);
    P->h_ctl.p[0] = P->h_ctl.p[1] = 0;
    STEREO_HIP_CHECK(hipMemset(P->d_msg.p, 0, sizeof(double) * (size_t)E * K));
    STEREO_HIP_CHECK(hipMemset(P->d_x.p, 0, sizeof(int32_t) * N));
    STEREO_HIP_CHECK(hipEventCreate(&P->ev0));
    STEREO_HIP_CHECK(hipEventCreate(&P->ev1));
    STEREO_HIP_CHECK(hipEventCreateWithFlags(&P->ev_bwd, hipEventDisableTiming));
    STEREO_HIP_CHECK(hipEventCreateWithFlags(&P->ev_lb, hipEventDisableTiming));
    STEREO_HIP_CHECK(hipStreamCreateWithFlags(&P->copy_stream, hipStreamNonBlocking));
    STEREO_HIP_CHECK(hipDeviceSynchronize());
    // every sweep kernel may need more than the default 64 KiB of dynamic LDS
    const int lds = (int)sweep_lds_bytes(P->Kp);
#define SET_LDS(KER, BW, MD)                                                                      \
  STEREO_HIP_CHECK(hipFuncSetAttribute((const void *)trws_sweep_kernel<KER, BW, MD>,              \
                                       hipFuncAttributeMaxDynamicSharedMemorySize, lds))
    SET_LDS(1, false, 0); SET_LDS(1, true, 0); SET_LDS(2, false, 0); SET_LDS(2, true, 0);
    SET_LDS(1, false, 1); SET_LDS(1, true, 1); SET_LDS(2, false, 1); SET_LDS(2, true, 1);
#undef SET_LDS
    const int plds = (int)persistent_lds_bytes(P->Kp);
    if (plds > 160 * 1024) return fail("stereo_trws: K too large for LDS", err, errcap);
#define SET_PLDS(KER, BW, MD, PR, UP)                                                              \
  STEREO_HIP_CHECK(hipFuncSetAttribute((const void *)trws_persistent_kernel<KER, BW, MD, PR, UP>,  \
                                       hipFuncAttributeMaxDynamicSharedMemorySize, plds))
#define SET_PLDS4(KER, MD)                                                                         \
  SET_PLDS(KER, false, MD, false, true); SET_PLDS(KER, true, MD, false, true);                    \
  SET_PLDS(KER, false, MD, true, true); SET_PLDS(KER, false, MD, true, false)
    SET_PLDS4(1, 0); SET_PLDS4(1, 1); SET_PLDS4(2, 0); SET_PLDS4(2, 1);
#undef SET_PLDS4
#undef SET_PLDS
    if (P->fast2) {
      const int lds2 = (int)(sizeof(double) * k2LdsDoubles);
#define SET_LDS2(SH)                                                                               \
  STEREO_HIP_CHECK(hipFuncSetAttribute((const void *)trws_pipe2_kernel<false, false, true, SH>, hipFuncAttributeMaxDynamicSharedMemorySize, lds2)); \
  STEREO_HIP_CHECK(hipFuncSetAttribute((const void *)trws_pipe2_kernel<true, false, true, SH>, hipFuncAttributeMaxDynamicSharedMemorySize, lds2));  \
  STEREO_HIP_CHECK(hipFuncSetAttribute((const void *)trws_pipe2_kernel<false, true, true, SH>, hipFuncAttributeMaxDynamicSharedMemorySize, lds2));  \
  STEREO_HIP_CHECK(hipFuncSetAttribute((const void *)trws_pipe2_kernel<false, true, false, SH>, hipFuncAttributeMaxDynamicSharedMemorySize, lds2))
      SET_LDS2(true); SET_LDS2(false);
#undef SET_LDS2
    }
    if (P->wide_allowed) {
      const int wlds = (int)(sizeof(double) * kWideLdsDoubles);
#define SET_WLDS(KER)                                                                              \
  STEREO_HIP_CHECK(hipFuncSetAttribute((const void *)trws_wide_kernel<KER, false, false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, wlds)); \
  STEREO_HIP_CHECK(hipFuncSetAttribute((const void *)trws_wide_kernel<KER, true, false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, wlds));  \
  STEREO_HIP_CHECK(hipFuncSetAttribute((const void *)trws_wide_kernel<KER, false, true, true>, hipFuncAttributeMaxDynamicSharedMemorySize, wlds));  \
  STEREO_HIP_CHECK(hipFuncSetAttribute((const void *)trws_wide_kernel<KER, false, true, false>, hipFuncAttributeMaxDynamicSharedMemorySize, wlds))
      SET_WLDS(1);
#undef SET_WLDS
    }
    if (strip_api) {
      const int glds = (int)(sizeof(double) * kWideLdsDoubles);
#define SET_GW(BW, PR, UP) STEREO_HIP_CHECK(hipFuncSetAttribute((const void *)trws_wide_group_kernel<1, BW, PR, UP>, hipFuncAttributeMaxDynamicSharedMemorySize, glds))
      SET_GW(false, false, true); SET_GW(true, false, true); SET_GW(false, true, true); SET_GW(false, true, false);
#undef SET_GW
      const int gplds = (int)(sizeof(double) * (2 * kStageDoubles + 4 * 8 * kWave + 2 * kScalDoubles + kPipeCompute * kPipeTab + 2 + 3 * kWave / 2));
#define SET_GP(KER, BW, PR, UP)                                                                                                        \
  STEREO_HIP_CHECK(hipFuncSetAttribute((const void *)trws_pipe_group_kernel<KER, BW, PR, UP, true>, hipFuncAttributeMaxDynamicSharedMemorySize, gplds)); \
  STEREO_HIP_CHECK(hipFuncSetAttribute((const void *)trws_pipe_group_kernel<KER, BW, PR, UP, false>, hipFuncAttributeMaxDynamicSharedMemorySize, gplds))
      SET_GP(1, false, false, true); SET_GP(1, true, false, true); SET_GP(1, false, true, true); SET_GP(1, false, true, false);
      SET_GP(2, false, false, true); SET_GP(2, true, false, true); SET_GP(2, false, true, true); SET_GP(2, false, true, false);
#undef SET_GP
    }
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
  if (plan && plan->d_timeline.p) {
    const bool chain = plan->graph->fast_ok && (plan->wide || plan->fast2 || plan->fast);
    const size_t R = (chain ? plan->graph->sweep[0].chain_run_ptr.size() : plan->graph->sweep[0].run_ptr.size()) - 1;
    std::vector<unsigned long long> t(4 * (R + 1));
    if (hipMemcpy(t.data(), plan->d_timeline.p, sizeof(unsigned long long) * 4 * R, hipMemcpyDeviceToHost) == hipSuccess) {
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
    unsigned long long v[32];
    if (hipMemcpy(v, plan->d_prof.p, sizeof(v), hipMemcpyDeviceToHost) == hipSuccess) {
      if (!plan->wide)
        std::fprintf(stderr, "[stereo_hip prof] cycles: p0 %llu p1 %llu p2 %llu p3 %llu p4 %llu | p5 %llu steps %llu\n",
                     v[0], v[1], v[2], v[3], v[4], v[5], v[6]);
      if (!plan->wide && v[9])
        std::fprintf(stderr, "[stereo_hip prof messages] certified attempt %.0f cycles x %llu | second look %.0f x %llu | "
                             "serial construction %.0f x %llu | walk %.0f x %llu\n",
                     (double)v[8] / v[9], v[9], v[11] ? (double)v[10] / v[11] : 0.0, v[11], v[13] ? (double)v[12] / v[13] : 0.0,
                     v[13], v[15] ? (double)v[14] / v[15] : 0.0, v[15]);
      if (plan->wide && v[22]) {
        std::fprintf(stderr, "[stereo_hip prof wide] cycles per visit of wave 0:");
        for (int i = 0; i < 16; ++i) std::fprintf(stderr, " [%d] %.0f", i, (double)v[i] / v[22]);
        std::fprintf(stderr, " | loader A %.0f B %.0f storer %.0f primal %.0f | hw barrier wait %.0f | visits %llu\n",
                     (double)v[16] / v[22], (double)v[17] / v[22], (double)v[18] / v[22], (double)v[19] / v[22],
                     (double)v[21] / v[22], v[22]);
      }
    }
  }
  delete plan;
}

int stereo_trws_plan_upload(stereo_trws_plan *P, const double *unary, const double *q,
                            const double *qprim, const double *positions, const double *alphas,
                            double tol, char *err, size_t errcap) {
  if (!P || !unary || !alphas) return fail("stereo_trws_plan_upload: NULL argument", err, errcap);
  const bool shared = (q == nullptr && qprim == nullptr);
  if (shared && !positions) return fail("stereo_trws_plan_upload: need q/qprim or positions", err, errcap);
  if (!shared && (!q || !qprim)) return fail("stereo_trws_plan_upload: q and qprim must both be given", err, errcap);
  try {
    const size_t K = P->K;
    P->o_unary.upload(unary, (size_t)P->N * K);
    P->o_alpha.upload(alphas, (size_t)P->E);
    P->unary = P->o_unary.p; P->alpha = P->o_alpha.p;
    if (shared) {
      P->o_pos.upload(positions, K);
      P->pos = P->o_pos.p; P->q = P->qprim = nullptr;
      P->o_q.release(); P->o_qprim.release();
    } else {
      P->o_q.upload(q, (size_t)P->E * K);
      P->o_qprim.upload(qprim, (size_t)P->E * K);
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
  if (!P || !d_unary || !d_alphas) return fail("stereo_trws_plan_bind_device: NULL argument", err, errcap);
  const bool shared = (d_q == nullptr && d_qprim == nullptr);
  if (shared && !d_positions) return fail("stereo_trws_plan_bind_device: need q/qprim or positions", err, errcap);
  if (!shared && (!d_q || !d_qprim)) return fail("stereo_trws_plan_bind_device: q and qprim must both be given", err, errcap);
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

int stereo_trws_plan_reset(stereo_trws_plan *P, char *err, size_t errcap) {
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
  if (P->persistent) {
    if (P->kernel == 1) {
      if (P->mode == 0) persistent_iteration<1, 0>(P, p, s); else persistent_iteration<1, 1>(P, p, s);
    } else {
      if (P->mode == 0) persistent_iteration<2, 0>(P, p, s); else persistent_iteration<2, 1>(P, p, s);
    }
  } else {
    if (P->kernel == 1) {
      if (P->mode == 0) launch_iteration<1, 0>(P, p, s); else launch_iteration<1, 1>(P, p, s);
    } else {
      if (P->mode == 0) launch_iteration<2, 0>(P, p, s); else launch_iteration<2, 1>(P, p, s);
    }
  }
  if (!P->lb_in_flight)
    STEREO_HIP_CHECK(hipMemcpyAsync(P->h_lb.p, P->d_lbterms.p, sizeof(double) * P->n_lb, hipMemcpyDeviceToHost, s));
  STEREO_HIP_CHECK(hipMemcpyAsync(P->h_en.p, P->d_eterms.p, sizeof(double) * P->n_en, hipMemcpyDeviceToHost, s));
  if (P->persistent)
    STEREO_HIP_CHECK(hipMemcpyAsync(P->h_ctl.p, P->d_ctl.p, 2 * sizeof(int32_t), hipMemcpyDeviceToHost, s));
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
  if (P->persistent && P->h_ctl.p[1]) return false;
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

static const char *kGaveUp = "stereo_trws: a persistent sweep gave up waiting on a dependency flag";

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
    for (int it = 0; it < iters; ++it) {
      issue_iteration(P, p, s);
      double lb = 0, en = 0;
      if (!collect_iteration(P, s, &lb, &en)) return fail(kGaveUp, err, errcap);
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
    total += P0->wide ? std::min(P->grid_blocks, P->cus) : P->grid_blocks;
    if (what != 3) P->sweep_launches += 1;
  }
  ga.first[n] = total;
  if (P0->wide) {
    const size_t wlds = sizeof(double) * kWideLdsDoubles;
    const dim3 grid(total), block(kWideThreads);
    switch (what) {
      case 0: hipLaunchKernelGGL((trws_wide_group_kernel<1, false, false, true>), grid, block, wlds, s, ga, epoch); break;
      case 1: hipLaunchKernelGGL((trws_wide_group_kernel<1, true, false, true>), grid, block, wlds, s, ga, epoch); break;
      case 2: hipLaunchKernelGGL((trws_wide_group_kernel<1, false, true, true>), grid, block, wlds, s, ga, epoch); break;
      default: hipLaunchKernelGGL((trws_wide_group_kernel<1, false, true, false>), grid, block, wlds, s, ga, epoch); break;
    }
  } else {
    const size_t plds = sizeof(double) * (2 * kStageDoubles + 4 * 8 * kWave + 2 * kScalDoubles + kPipeCompute * kPipeTab + 2 + 3 * kWave / 2);
    const bool sh = P0->pos != nullptr;
    const dim3 grid(total), block(kPipeThreads);
#define GPIPE(KER, BW, PR, UP)                                                                                        \
  do {                                                                                                                \
    if (sh) hipLaunchKernelGGL((trws_pipe_group_kernel<KER, BW, PR, UP, true>), grid, block, plds, s, ga, epoch);       \
    else hipLaunchKernelGGL((trws_pipe_group_kernel<KER, BW, PR, UP, false>), grid, block, plds, s, ga, epoch);         \
  } while (0)
#define GPIPE4(KER)                                                                                                   \
  switch (what) {                                                                                                     \
    case 0: GPIPE(KER, false, false, true); break;                                                                    \
    case 1: GPIPE(KER, true, false, true); break;                                                                     \
    case 2: GPIPE(KER, false, true, true); break;                                                                     \
    default: GPIPE(KER, false, true, false); break;                                                                   \
  }
    if (P0->kernel == 1) { GPIPE4(1) } else { GPIPE4(2) }
#undef GPIPE4
#undef GPIPE
  }
  STEREO_HIP_CHECK(hipGetLastError());
}

int stereo_trws_plans_issue(stereo_trws_plan *const *plans, int n, void *stream, char *err, size_t errcap) {
  if (!plans || n < 1 || n > kMaxGroup) return fail("stereo_trws_plans_issue: need 1 .. 16 plans", err, errcap);
  for (int i = 0; i < n; ++i) {
    if (int rc = strip_ready(plans[i], "stereo_trws_plans_issue", err, errcap)) return rc;
    stereo_trws_plan *P = plans[i], *P0 = plans[0];
    if (P->issued) return fail("stereo_trws_plans_issue: the previous iteration has not been collected", err, errcap);
    if (P->device != P0->device || P->graph != P0->graph || P->K != P0->K || P->kernel != P0->kernel ||
        P->epoch != P0->epoch || P->fwd_pending != P0->fwd_pending || P->wide != P0->wide || P->fast != P0->fast ||
        (P->pos == nullptr) != (P0->pos == nullptr) || P->mode != P0->mode)
      return fail("stereo_trws_plans_issue: the plans are not strips of one problem on one device in the same state", err, errcap);
    if (!P->persistent || !(P->wide || P->fast))
      return fail("stereo_trws_plans_issue: strips run on the pipelined kernels only (K <= 64, or K <= 256 with shared "
                  "ascending positions and the linear kernel)", err, errcap);
  }
  try {
    stereo_trws_plan *P0 = plans[0];
    hipStream_t s = stream ? (hipStream_t)stream : P0->own_stream;
    if (!s) return fail("stereo_trws_plans_issue: a plain plan needs an explicit stream here", err, errcap);
    if (P0->d_group.n < (size_t)n) { P0->d_group.alloc(kMaxGroup); P0->h_group.alloc(kMaxGroup); }
    for (int i = 0; i < n; ++i) P0->h_group.p[i] = make_params(plans[i]);
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
      STEREO_HIP_CHECK(hipMemcpyAsync(P->h_ctl.p, P->d_ctl.p, 2 * sizeof(int32_t), hipMemcpyDeviceToHost, s));
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
  if (!P) return fail("stereo_trws_plan_collect: NULL plan", err, errcap);
  if (!P->issued) return fail("stereo_trws_plan_collect: nothing was issued", err, errcap);
  try {
    double lb = 0, en = 0;
    if (!collect_iteration(P, P->issue_stream, &lb, &en)) return fail(kGaveUp, err, errcap);
    if (lb_part) *lb_part = lb;
    if (energy_part) *energy_part = en;
    return 0;
  } catch (const HipError &e) {
    return fail(e.msg, err, errcap);
  }
}

int stereo_trws_plan_commit(stereo_trws_plan *P, double lower_bound, double energy, char *err, size_t errcap) {
  if (!P) return fail("stereo_trws_plan_commit: NULL plan", err, errcap);
  P->lb = lower_bound; P->energy = energy; P->iterations += 1;
  return 0;
}

int stereo_trws_plan_connect(stereo_trws_plan *P, int which, stereo_trws_plan *peer, char *err, size_t errcap) {
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
  if (!P) return 1;
  if (done && hipMemcpy(done, P->d_done.p, sizeof(int32_t) * P->N, hipMemcpyDeviceToHost) != hipSuccess) return 1;
  if (ctl && hipMemcpy(ctl, P->d_ctl.p, sizeof(int32_t) * 2, hipMemcpyDeviceToHost) != hipSuccess) return 1;
  return 0;
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
  if (!P) return fail("stereo_trws_plan_result: NULL plan", err, errcap);
  try {
    if (labelling) {
      STEREO_HIP_CHECK(hipMemcpy(P->h_x.p, P->d_x.p, sizeof(int32_t) * P->N, hipMemcpyDeviceToHost));
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
  if (!P) return 1;
  if (sweep_ms) *sweep_ms = P->sweep_ms;
  if (sweep_launches) *sweep_launches = P->sweep_launches;
  if (reset) { P->sweep_ms = 0; P->sweep_launches = 0; }
  P->time_sweeps = true;
  return 0;
}

int stereo_trws_plan_counters(stereo_trws_plan *P, int64_t *serial_messages, int reset) {
  if (!P) return 1;
  unsigned long long v = 0;
  if (hipMemcpy(&v, P->d_fallbacks.p, sizeof(v), hipMemcpyDeviceToHost) != hipSuccess) return 1;
  if (serial_messages) *serial_messages = (int64_t)v;
  if (reset && hipMemset(P->d_fallbacks.p, 0, sizeof(v)) != hipSuccess) return 1;
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
    if (kernel == 2 && shared_positions) {
      p.pos_gap = std::numeric_limits<double>::infinity();
      for (int k = 1; k < K; ++k) p.pos_gap = std::min(p.pos_gap, shared_positions[k] - shared_positions[k - 1]);
    }
    if (const char *dbg = std::getenv("STEREO_HIP_TRWS_DEBUG")) p.debug = std::atoi(dbg);
    const unsigned grid = (unsigned)std::min<int64_t>(M, 4096);
    if (kernel == 1)
      hipLaunchKernelGGL(trws_messages_kernel<1>, dim3(grid), dim3(kWave), 0, 0, p, K, M, dD.p, dg.p, dm.p, dqs.p, dqd.p,
                         da.p, dperm.p, shared_positions ? window : -1, dout.p, dv.p, dser.p, dfb.p);
    else
      hipLaunchKernelGGL(trws_messages_kernel<2>, dim3(grid), dim3(kWave), 0, 0, p, K, M, dD.p, dg.p, dm.p, dqs.p, dqd.p,
                         da.p, dperm.p, shared_positions ? window : -1, dout.p, dv.p, dser.p, dfb.p);
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
  if (!P->persistent) return 0;
  return P->wide ? 3 : P->fast2 ? 4 : P->fast ? 2 : 1;
}

int stereo_trws(int kernel, const double *unary, const uint32_t *conn, const double *q,
                const double *qprim, const double *alphas, double tol, double maxiter,
                double max_relgap, int K, int64_t N, int64_t E, double *labelling, double *energy,
                double *lower_bound, double *iterations, char *err, size_t errcap) {
  if (kernel != 1 && kernel != 2) return fail("Unsupported kernel", err, errcap);  // trws_mex.cpp:162
  if (!unary || !conn || !q || !qprim || !alphas || !labelling || !energy || !lower_bound || !iterations)
    return fail("stereo_trws: NULL argument", err, errcap);
  stereo_trws_plan *P = nullptr;
  int mode = STEREO_TRWS_MESSAGES_EXACT;
  if (const char *m = std::getenv("STEREO_HIP_TRWS_MESSAGES"))
    if (std::string(m) == "minplus") mode = STEREO_TRWS_MESSAGES_MINPLUS;
  int rc = stereo_trws_plan_create(kernel, K, N, E, conn, mode, &P, err, errcap);
  if (rc) return rc;
  rc = stereo_trws_plan_upload(P, unary, q, qprim, nullptr, alphas, tol, err, errcap);
  if (!rc) {
    // Minimize_TRW_S always runs at least one iteration (minimize.cpp:31,100-101)
    int itmax = (int)maxiter;  // trws_mex.cpp:125
    if (itmax < 1) itmax = 1;
    rc = stereo_trws_plan_iterate(P, itmax, max_relgap, nullptr, nullptr, nullptr, err, errcap);
  }
  if (!rc) rc = stereo_trws_plan_result(P, labelling, energy, lower_bound, iterations, err, errcap);
  stereo_trws_plan_destroy(P);
  return rc;
}

}  // extern "C"
