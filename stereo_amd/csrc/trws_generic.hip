// TRW-S sweep kernel for everything the pipelined kernels do not take (any graph, K up to 512, both
// message modes): one persistent launch per sweep, one wave per outgoing message, state in LDS.
// Part of libstereo_hip.so; overview in trws_plan.hip.
#include <hip/hip_runtime.h>

#include <cstdint>

#include "../../include/stereo_hip.h"
#include "common.h"
#include "trws_dev.h"
#include "trws_launch.h"

namespace stereo {
namespace {

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
            m2[it] = min_raw_if(hi > lo, m2[it], hi);
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
        int spins = 0, v;
        long long t0 = 0;
        while ((v = ld_sc1(flag)) < epoch) {
          if (!keep_waiting(p, spins, t0, p.dep_rank[D][d0 + tid] >= p.n_own)) {
            report_give_up(p, r, p.dep_rank[D][d0 + tid], v, epoch);
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

}  // namespace

size_t generic_lds_bytes(int Kp) {
  return sizeof(double) * (size_t)(Kp + 8 + kMaxSlots * Kp + Kp + kWavesPerBlock * (kWaveVecs * Kp + 8));
}

void generic_set_attributes(int lds) {
#define SET_PLDS(KER, BW, MD, PR, UP)                                                              \
  STEREO_HIP_CHECK(hipFuncSetAttribute((const void *)trws_persistent_kernel<KER, BW, MD, PR, UP>,  \
                                       hipFuncAttributeMaxDynamicSharedMemorySize, lds))
#define SET_PLDS4(KER, MD)                                                                         \
  SET_PLDS(KER, false, MD, false, true); SET_PLDS(KER, true, MD, false, true);                    \
  SET_PLDS(KER, false, MD, true, true); SET_PLDS(KER, false, MD, true, false)
  SET_PLDS4(1, 0); SET_PLDS4(1, 1); SET_PLDS4(2, 0); SET_PLDS4(2, 1);
#undef SET_PLDS4
#undef SET_PLDS
}

void launch_generic(int kernel, int mode, int what, int blocks, size_t lds, hipStream_t s, const DevParams &p, int epoch) {
  const dim3 grid(blocks), block(kBlock);
#define GEN(KER, MD)                                                                                                          \
  switch (what) {                                                                                                             \
    case 0: hipLaunchKernelGGL((trws_persistent_kernel<KER, false, MD, false, true>), grid, block, lds, s, p, epoch); break;  \
    case 1: hipLaunchKernelGGL((trws_persistent_kernel<KER, true, MD, false, true>), grid, block, lds, s, p, epoch); break;   \
    case 2: hipLaunchKernelGGL((trws_persistent_kernel<KER, false, MD, true, true>), grid, block, lds, s, p, epoch); break;   \
    default: hipLaunchKernelGGL((trws_persistent_kernel<KER, false, MD, true, false>), grid, block, lds, s, p, epoch); break; \
  }
  if (kernel == 1) { if (mode == 0) { GEN(1, 0) } else { GEN(1, 1) } }
  else { if (mode == 0) { GEN(2, 0) } else { GEN(2, 1) } }
#undef GEN
  STEREO_HIP_CHECK(hipGetLastError());
}

}  // namespace stereo
