// TRW-S pipelined sweep kernel for 64 < K <= 128 with per-edge positions (two labels per lane), both smoothness
// kernels (the quadratic one since round 5).  Part of libstereo_hip.so; overview in trws_plan.hip.
#include <hip/hip_runtime.h>

#include <cstdint>

#include "../../include/stereo_hip.h"
#include "common.h"
#include "trws_dev.h"
#include "trws_launch.h"

namespace stereo {
namespace {

// ---- pipelined persistent sweep for 64 < K <= 128 (two labels per lane), linear kernel -------
// The same role-specialised structure as trws_pipe_kernel -- eight compute waves (one message each),
// loader, storer, primal, one barrier per visit -- with rows of 128 labels: lane owns labels lane
// and lane + 64.  General (per-edge) or shared positions; this is what a simultaneous fusion of
// 64 .. 127 proposals runs on (example_ncc.m fuses 78).  A message is min-plus over the useful
// sources from a per-wave (h, q, u, v) table in LDS plus the certificate of DESIGN.md 4.3; if that
// fails the reference's serial construction runs in the wave's own LDS scratch.
constexpr int k2W = 2 * kWave;                      // row width
constexpr int k2StD = 0, k2StM = k2W, k2StQ = k2W + 8 * k2W, k2StQP = k2W + 16 * k2W, k2StA = k2W + 24 * k2W;
constexpr int k2StG = k2StA + 8;                     // gamma = 1 / max(n_out, n_in) (the loader's division)
constexpr int k2StI = k2StA + 10;
constexpr int k2Stage = k2StI + 40;                  // ints: desc[64] px[8] row[8] (LDS offsets of the node's message rows)
constexpr int k2Fb = 5 * (k2W + 2);                 // serial scratch per compute wave: sorted h, q; stack h, q; breakpoints
constexpr int k2LdsDoubles = 2 * k2Stage + 4 * 8 * k2W + 2 * kScalDoubles + kPipeCompute * 4 * k2W + kPipeCompute * k2Fb + 2;
static_assert(k2LdsDoubles * 8 <= 160 * 1024, "pipe2 kernel LDS");

template <int KERNEL, bool BACKWARD, bool PRIMAL, bool UPDATE, bool SHARED>
__device__ __forceinline__ void pipe2_body(DevParams p, int epoch) {
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
    if (p.timeline && tid == 0) p.timeline[((size_t)D * p.tl_stride + run) * 2] = wall_clock64();

    unsigned long long busy = 0;  // development profile (STEREO_HIP_TRWS_PROF): cycles from barrier to barrier arrival
    // (every role walks the run in its own loop, as in trws_pipe.hip: the same visits, the same barrier, registers per role)
    if (wave < kPipeCompute) {
    // ---- compute
    for (int pos = p0 - 1; pos <= p1; ++pos) {
      const long long tstart = p.prof ? (long long)__builtin_readcyclecounter() : 0;
      double *st = stage0 + (pos & 1) * k2Stage;
      double *stn = stage0 + ((pos + 1) & 1) * k2Stage;
      double *hcur = hand + (pos & 3) * 8 * k2W, *hprev = hand + ((pos - 1) & 3) * 8 * k2W;
      double *sc = scal + (pos & 1) * kScalDoubles;
      const bool have_node = pos >= p0 && pos < p1;
      {
        // ------------------------------------------------------------ compute
        if (UPDATE && have_node) {
          const int *sti = (const int *)(st + k2StI);
          const int f = __builtin_amdgcn_readfirstlane(sti[2]);
          const int nout = f & 15, nin = (f >> 4) & 15, md = (f >> 16) & 255, ntot = nout + nin;
          const int myrow = sti[72 + (lane & 7)];  // where the node's message rows live in LDS (from the loader)
          if (wave < nout || (BACKWARD && wave == 0)) {
            double Di[2] = {st[k2StD + kk[0]], st[k2StD + kk[1]]};
            // (this wave's own old message is read directly rather than picked out of the loop)
            const int jown = wave < nout ? wave : 0;
            const double mown[2] = {st[k2StM + jown * k2W + kk[0]], st[k2StM + jown * k2W + kk[1]]};
#pragma unroll
            for (int j = 0; j < 8; ++j) {
              if (j < ntot) {
                const double *row = lds + __builtin_amdgcn_readlane(myrow, j);
                const double v0 = row[kk[0]], v1 = row[kk[1]];
                Di[0] += v0; Di[1] += v1;
              }
            }
            if (BACKWARD) {
              const double node_vmin = wave_min_dpp(min_raw(act[0] ? Di[0] : inf, act[1] ? Di[1] : inf));
              Di[0] -= node_vmin; Di[1] -= node_vmin;
              if (tid == 0) sc[8] = node_vmin;
            }
            {
              const int j = wave;  // one compute wave per outgoing message: ONE instance of the message code
              if (j < nout) {
                const double gamma = st[k2StG];  // (double)1 / (double)max(n_out, n_in)
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
                  if (!need_serial && KERNEL == 2) {
                    // ---- truncated QUADRATIC kernel (typeStereoQuadratic.h:329-501): the hull-slope certificate of
                    // message_quad_fast (trws_dev.h) on two labels per lane, per-edge positions -- plain min-plus over the
                    // useful parabolas, smallest and second smallest cost per destination (equal costs of two sources count),
                    // gap = distance from a useful source to the nearest other source, Q = span of the source positions.
                    double sc2 = 0, qlo = inf, qhi = -inf;
#pragma unroll
                    for (int c = 0; c < 2; ++c) {
                      if (act[c]) {
                        sc2 = max_raw(sc2, fabs(h[c]) + alpha * qsrc[c] * qsrc[c] + alpha * qdst[c] * qdst[c]);
                        qlo = min_raw(qlo, qsrc[c]); qhi = max_raw(qhi, qsrc[c]);
                      }
                    }
                    sc2 = wave_max_dpp(sc2);
                    wave_min_max_dpp(qlo, qhi);
                    const double delta = 1e-9 * (sc2 + fabs(alpha * p.lambda) + fabs(vtrunc));
                    double *tab = tabs + wave * 4 * k2W;
#pragma unroll
                    for (int c = 0; c < 2; ++c) { tab[4 * kk[c]] = h[c]; tab[4 * kk[c] + 1] = qsrc[c]; }
                    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                    __builtin_amdgcn_wave_barrier();
                    double m1[2] = {inf, inf}, m2[2] = {inf, inf};
                    double gap = inf;
                    asm volatile("" : "+v"(gap));   // (not a wave-uniform constant: see message_quad_fast)
#pragma unroll
                    for (int cs = 0; cs < 2; ++cs) {
                      unsigned long long mask = __builtin_amdgcn_ballot_w64(act[cs] && h[cs] < vtrunc);
                      while (mask) {
                        const int js = __builtin_ctzll(mask) + cs * kWave;
                        mask &= mask - 1;
                        const double hj = tab[4 * js], qj = tab[4 * js + 1];
#pragma unroll
                        for (int c = 0; c < 2; ++c) {
                          const double cst = pair_cost<2>(alpha, qdst[c] - qj, hj);
                          const double lo = min_raw(m1[c], cst), hi = max_raw(m1[c], cst);
                          m2[c] = min_raw(m2[c], hi);
                          m1[c] = lo;
                          gap = min_raw_if(act[c] && kk[c] != js, gap, fabs(qsrc[c] - qj));
                        }
                      }
                    }
                    gap = wave_min_dpp(gap);
                    bool bad = !(delta < inf) || !(alpha > 0) || !(gap > 4e-8);
                    bad = bad || !(1e-13 * sc2 * (qhi - qlo) < delta * gap);
#pragma unroll
                    for (int c = 0; c < 2; ++c) {
                      bad = bad || (act[c] && m1[c] < vtrunc && !(m2[c] - m1[c] > delta && vtrunc - m1[c] > delta));
                      out[c] = m1[c] < vtrunc ? m1[c] : vtrunc;
                    }
                    need_serial = UNI(bad);
                    if (need_serial && lane == 0 && p.fallbacks) atomicAdd(p.fallbacks, 1);
                  }
                  if (!need_serial && KERNEL == 1) {
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
                            m2[c] = min_raw_if(hi > lo, m2[c], hi);
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
                    if (lane == 0) build_envelope<KERNEL>(K, alpha, A, B, sh, sq, z);
                    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                    __builtin_amdgcn_wave_barrier();
#pragma unroll
                    for (int c = 0; c < 2; ++c) {
                      if (act[c]) {
                        int jj = 0;
                        while (z[jj + 1] < qdst[c]) ++jj;
                        const double cst = pair_cost<KERNEL>(alpha, qdst[c] - sq[jj], sh[jj]);
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
      }
      if (p.prof) busy += (unsigned long long)((long long)__builtin_readcyclecounter() - tstart);
      __syncthreads();
      if (ctl[1]) {
        if (tid == 0) st_sc1(p.abort_flag, 1);
        return;
      }
    }
    } else if (wave == kPipeCompute) {
    // ---- loader: stage node pos + 1
    for (int pos = p0 - 1; pos <= p1; ++pos) {
      const long long tstart = p.prof ? (long long)__builtin_readcyclecounter() : 0;
      double *st = stage0 + (pos & 1) * k2Stage;
      double *stn = stage0 + ((pos + 1) & 1) * k2Stage;
      double *hcur = hand + (pos & 3) * 8 * k2W, *hprev = hand + ((pos - 1) & 3) * 8 * k2W;
      double *sc = scal + (pos & 1) * kScalDoubles;
      const bool have_node = pos >= p0 && pos < p1;
      {
        // ------------------------------------------------------------ loader: stage node pos + 1
        if (pos + 1 >= p0 && pos + 1 < p1) {
          const int w = wnext;
          if (pos + 2 < p1) wnext = desc[(size_t)(pos + 2) * DW + lane];
          const NodeDesc nx = decode_desc(w);
          int *stni = (int *)(stn + k2StI);
          stni[lane] = w;
          {
            int slr = -1;
#pragma unroll
            for (int j = 0; j < 8; ++j)
              if (lane == j) slr = j >= nx.nout ? nx.slot[j] : -1;
            const int row = slr >= 8 ? (int)(hand - lds) + ((pos - 1) & 3) * 8 * k2W + (slr - 8) * k2W
                          : slr >= 0 ? (int)(hand - lds) + (pos & 3) * 8 * k2W + slr * k2W
                                     : (int)(stn - lds) + k2StM + lane * k2W;
            if (lane < 8) stni[72 + lane] = row;
            if (lane == 0) stn[k2StG] = (double)1 / (double)(nx.nout > nx.nin ? nx.nout : nx.nin);
          }
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
          wait_for_dependencies(p, nx.ndep, nx.dep[0], nx.dep[1], nx.dep[2], nx.dep[3], nx.rank, epoch, lane, ctl + 1);
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
      }
      if (p.prof) busy += (unsigned long long)((long long)__builtin_readcyclecounter() - tstart);
      __syncthreads();
      if (ctl[1]) {
        if (tid == 0) st_sc1(p.abort_flag, 1);
        return;
      }
    }
    } else if (wave == kPipeCompute + 1) {
    // ---- storer: node pos - 1
    for (int pos = p0 - 1; pos <= p1; ++pos) {
      const long long tstart = p.prof ? (long long)__builtin_readcyclecounter() : 0;
      double *st = stage0 + (pos & 1) * k2Stage;
      double *stn = stage0 + ((pos + 1) & 1) * k2Stage;
      double *hcur = hand + (pos & 3) * 8 * k2W, *hprev = hand + ((pos - 1) & 3) * 8 * k2W;
      double *sc = scal + (pos & 1) * kScalDoubles;
      const bool have_node = pos >= p0 && pos < p1;
      {
        // ------------------------------------------------------------ storer: node pos - 1
        if (pos - 1 >= p0) {
          const NodeDesc pd = decode_desc(desc[(size_t)(pos - 1) * DW + lane]);
          const double *scp = scal + ((pos + 1) & 1) * kScalDoubles;
          if (UPDATE) {
#pragma unroll
            for (int j = 0; j < 8; ++j) {
              if (j < pd.nout) {
                double *mb = ((pd.remote >> j) & 1) ? (((pd.remote >> (8 + j)) & 1) ? p.peer_msg1 : p.peer_msg0) : p.msg;
                const int ej = ((pd.remote >> j) & 1) ? pd.re[j] : pd.e[j];  // the neighbour numbers the edge itself
#pragma unroll
                for (int c = 0; c < 2; ++c)
                  if (act[c]) st_sc1(mb + (size_t)ej * K + kk[c], hprev[j * k2W + kk[c]]);
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
            st_sc1(p.done + pd.rank, epoch);
            if (pd.remote & (1 << 16)) st_sc1(p.peer_done0 + pd.pn[0], epoch);
            if (pd.remote & (1 << 17)) st_sc1(p.peer_done1 + pd.pn[1], epoch);
          }
        }
      }
      if (p.prof) busy += (unsigned long long)((long long)__builtin_readcyclecounter() - tstart);
      __syncthreads();
      if (ctl[1]) {
        if (tid == 0) st_sc1(p.abort_flag, 1);
        return;
      }
    }
    } else if (wave == kPipeCompute + 3) {
    // ---- primal of node pos
    for (int pos = p0 - 1; pos <= p1; ++pos) {
      const long long tstart = p.prof ? (long long)__builtin_readcyclecounter() : 0;
      double *st = stage0 + (pos & 1) * k2Stage;
      double *stn = stage0 + ((pos + 1) & 1) * k2Stage;
      double *hcur = hand + (pos & 3) * 8 * k2W, *hprev = hand + ((pos - 1) & 3) * 8 * k2W;
      double *sc = scal + (pos & 1) * kScalDoubles;
      const bool have_node = pos >= p0 && pos < p1;
      {
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
                  const double v = KERNEL == 1 ? fabs(d) : d * d;
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
      if (p.prof) busy += (unsigned long long)((long long)__builtin_readcyclecounter() - tstart);
      __syncthreads();
      if (ctl[1]) {
        if (tid == 0) st_sc1(p.abort_flag, 1);
        return;
      }
    }
    } else {
    // ---- (idle wave)
    for (int pos = p0 - 1; pos <= p1; ++pos) {
      const long long tstart = p.prof ? (long long)__builtin_readcyclecounter() : 0;
      double *st = stage0 + (pos & 1) * k2Stage;
      double *stn = stage0 + ((pos + 1) & 1) * k2Stage;
      double *hcur = hand + (pos & 3) * 8 * k2W, *hprev = hand + ((pos - 1) & 3) * 8 * k2W;
      double *sc = scal + (pos & 1) * kScalDoubles;
      const bool have_node = pos >= p0 && pos < p1;
      {
        (void)0;
      }
      if (p.prof) busy += (unsigned long long)((long long)__builtin_readcyclecounter() - tstart);
      __syncthreads();
      if (ctl[1]) {
        if (tid == 0) st_sc1(p.abort_flag, 1);
        return;
      }
    }
    }
    if (p.timeline && tid == 0) p.timeline[((size_t)D * p.tl_stride + run) * 2 + 1] = wall_clock64();
    if (p.prof && lane == 0 && (p.prof_run < 0 || run == p.prof_run)) {
      atomicAdd(p.prof + 32 + wave, busy);
      if (wave == 0) atomicAdd(p.prof + 6, (unsigned long long)(p1 - p0));
    }
  }
}

template <int KERNEL, bool BACKWARD, bool PRIMAL, bool UPDATE, bool SHARED>
__global__ __launch_bounds__(kPipeThreads) void trws_pipe2_kernel(DevParams p, int epoch) {
  pipe2_body<KERNEL, BACKWARD, PRIMAL, UPDATE, SHARED>(p, epoch);
}

// row strips that share a device: one launch, workgroup b works for the strip group_strip() names
template <int KERNEL, bool BACKWARD, bool PRIMAL, bool UPDATE, bool SHARED>
__global__ __launch_bounds__(kPipeThreads) void trws_pipe2_group_kernel(GroupArgs ga, int epoch) {
  pipe2_body<KERNEL, BACKWARD, PRIMAL, UPDATE, SHARED>(ga.pp[group_strip(ga)], epoch);
}

}  // namespace

size_t pipe2_lds_bytes() { return sizeof(double) * k2LdsDoubles; }

void pipe2_set_attributes() {
  const int lds2 = (int)pipe2_lds_bytes();
#define SET_LDS2(NAME, KER, SH)                                                                                                            \
  STEREO_HIP_CHECK(hipFuncSetAttribute((const void *)NAME<KER, false, false, true, SH>, hipFuncAttributeMaxDynamicSharedMemorySize, lds2)); \
  STEREO_HIP_CHECK(hipFuncSetAttribute((const void *)NAME<KER, true, false, true, SH>, hipFuncAttributeMaxDynamicSharedMemorySize, lds2));  \
  STEREO_HIP_CHECK(hipFuncSetAttribute((const void *)NAME<KER, false, true, true, SH>, hipFuncAttributeMaxDynamicSharedMemorySize, lds2));  \
  STEREO_HIP_CHECK(hipFuncSetAttribute((const void *)NAME<KER, false, true, false, SH>, hipFuncAttributeMaxDynamicSharedMemorySize, lds2))
  SET_LDS2(trws_pipe2_kernel, 1, true); SET_LDS2(trws_pipe2_kernel, 1, false);
  SET_LDS2(trws_pipe2_group_kernel, 1, true); SET_LDS2(trws_pipe2_group_kernel, 1, false);
  SET_LDS2(trws_pipe2_kernel, 2, true); SET_LDS2(trws_pipe2_kernel, 2, false);
  SET_LDS2(trws_pipe2_group_kernel, 2, true); SET_LDS2(trws_pipe2_group_kernel, 2, false);
#undef SET_LDS2
}

#define PIPE2_SWITCH(NAME, ARG)                                                                   \
  const size_t lds2 = pipe2_lds_bytes();                                                          \
  const dim3 grid2(blocks), block2(kPipeThreads);                                                 \
  if (kernel == 1) { PIPE2_4(NAME, 1, ARG) } else { PIPE2_4(NAME, 2, ARG) }                       \
  STEREO_HIP_CHECK(hipGetLastError());
#define PIPE2_4(NAME, KER, ARG)                                                                   \
  switch (what) {                                                                                 \
    case 0: PIPE2(NAME, KER, false, false, true, ARG); break;                                     \
    case 1: PIPE2(NAME, KER, true, false, true, ARG); break;                                      \
    case 2: PIPE2(NAME, KER, false, true, true, ARG); break;                                      \
    default: PIPE2(NAME, KER, false, true, false, ARG); break;                                    \
  }
#define PIPE2(NAME, KER, BW, PR, UP, ARG)                                                         \
  do {                                                                                            \
    if (shared) hipLaunchKernelGGL((NAME<KER, BW, PR, UP, true>), grid2, block2, lds2, s, ARG, epoch);  \
    else hipLaunchKernelGGL((NAME<KER, BW, PR, UP, false>), grid2, block2, lds2, s, ARG, epoch);        \
  } while (0)

void launch_pipe2(int kernel, bool shared, int what, int blocks, hipStream_t s, const DevParams &p, int epoch) {
  PIPE2_SWITCH(trws_pipe2_kernel, p)
}
void launch_pipe2_group(int kernel, bool shared, int what, int blocks, hipStream_t s, const GroupArgs &ga, int epoch) {
  PIPE2_SWITCH(trws_pipe2_group_kernel, ga)
}
#undef PIPE2_4
#undef PIPE2
#undef PIPE2_SWITCH

}  // namespace stereo
