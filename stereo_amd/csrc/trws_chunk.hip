// TRW-S pipelined sweep kernel for K <= 256 on shared strictly ascending positions, linear kernel,
// STEREO_TRWS_MESSAGES_MINPLUS: the message is the (windowed) min-plus, nothing else.  Part of
// libstereo_hip.so; overview in trws_plan.hip.
//
// Same workgroup skeleton as trws_wide_kernel (two loader waves, storer, primal, the 3-deep LDS
// ring of new messages, one hardware barrier per visit) but the compute waves are organised by
// LABEL CHUNK, not by message: wave c owns labels 64 c .. 64 c + 63 of EVERY outgoing message of
// the node.  All four SIMDs work on every visit whatever the node's degree (interior nodes have two
// outgoing messages: the per-message layout leaves half the compute waves idle), `Di` is formed
// once, and the few quantities that span chunks -- min Di (backward), min H_j, min of the new
// message -- are exchanged through LDS slots with a counter barrier among the compute waves
// (two per visit, three in the backward sweep).  H_j goes into a padded LDS table per message and
// the window loop reads it at lane + d, so neighbouring chunks need no special case.
// Bit for bit the oracle's brute-force min-plus messages (min is order independent):
// tests/test_trws_wide_gpu.py.
#include <hip/hip_runtime.h>

#include <cstdint>

#include "../../include/stereo_hip.h"
#include "common.h"
#include "trws_dev.h"
#include "trws_launch.h"

namespace stereo {
namespace {

constexpr int kChunkCompute = 4;
constexpr int kChunkWaves = kChunkCompute + 4;  // + loader (own data), loader (foreign data), storer, primal
constexpr int kChunkThreads = kChunkWaves * kWave;
constexpr int kWS = 260;    // LDS row stride in doubles
constexpr int kWPad = 16;   // the H tables are padded by this many +inf entries on both sides
constexpr int kCTab = 256 + 2 * kWPad;
constexpr int kWStG = kWS + 8 * kWS + 8;            // gamma = 1 / max(n_out, n_in) of the node (the loader's division)
constexpr int kWStI = kWS + 8 * kWS + 10;           // int area of a stage (in doubles)
constexpr int kWStage = kWStI + 40;  // ints: desc[64] px[8] row[8]
// stage: D[kWS] m[8][kWS] a[8] | ints desc[64] px[8] row[8] (row: where Di's k-th message row lives in LDS, in doubles)

struct ChunkPtrs {
  double *stage0, *hand, *tab, *ptab, *pos, *scal, *redn, *redh;
  int *dring, *ctl;
};
__device__ __forceinline__ ChunkPtrs chunk_carve(double *lds) {
  ChunkPtrs w;
  w.stage0 = lds;                            // 2 * kWStage
  w.hand = w.stage0 + 2 * kWStage;           // 3 * 8 * kWS : new messages of the last three visits
  w.tab = w.hand + 3 * 8 * kWS;              // 4 * kCTab : H_j of the (up to four) messages in flight
  w.ptab = w.tab + 4 * kCTab;                // kCTab : positions with the tables' padding (0 there)
  w.pos = w.ptab + kCTab;                    // kWS
  w.scal = w.pos + kWS;                      // 2 * kScalDoubles
  w.redn = w.scal + 2 * kScalDoubles;        // [4] min Di per chunk (backward sweep)
  w.redh = w.redn + 4;                       // [4][4] min H_j per chunk
  w.dring = (int *)(w.redh + 16);            // 3 * 64 descriptor words (for the storer)
  w.ctl = w.dring + 3 * 64;                  // [0] run, [1] abort, [3] arrivals at the compute waves' barrier (counts up)
  return w;
}
constexpr int kChunkLdsDoubles = 2 * kWStage + 3 * 8 * kWS + 5 * kCTab + kWS + 2 * kScalDoubles + 20 + 96 + 4;

template <int KERNEL, bool BACKWARD, bool PRIMAL, bool UPDATE>
__device__ __forceinline__ void chunk_body(DevParams p, int epoch) {
  extern __shared__ __attribute__((aligned(16))) double lds[];
  const ChunkPtrs L = chunk_carve(lds);
  const int K = p.K;
  const int C = (K + kWave - 1) / kWave;  // 64-label chunks
  const double inf = __builtin_huge_val();
  const int tid = threadIdx.x, lane = tid & (kWave - 1);
  const int wave = __builtin_amdgcn_readfirstlane(tid / kWave);
  constexpr int D = BACKWARD ? 1 : 0;
  constexpr int DW = TrwsGraph::kDescWords;
  const int32_t *desc = p.desc[D];
  for (int k = tid; k < kWS; k += kChunkThreads) L.pos[k] = k < K ? p.pos[k] : inf;
  // positions on an exact arithmetic progression (checked on the host: pos[k+d] - pos[k] == d * step
  // bit for bit): the min-plus source table then holds h only and alpha |d step| is formed once per d
  const double ustep = p.uniform_step;
  const bool uniform = ustep != 0;
  for (int k = tid; k < 5 * kCTab; k += kChunkThreads) {
    // table padding (never overwritten afterwards): +inf costs, position 0
    const int t = k / kCTab, i = k - t * kCTab;
    if (t < 4) { if (i < kWPad || i >= kWPad + K) L.tab[k] = inf; }
    else L.ptab[i] = (i >= kWPad && i < kWPad + K) ? p.pos[i - kWPad] : 0.0;
  }
  if (tid == 0) { L.ctl[1] = 0; L.ctl[3] = 0; }
  int arrivals = 0;  // what the compute waves' barrier counter must reach next
  double posr[4];  // this lane's four label positions
#pragma unroll
  for (int c = 0; c < 4; ++c) posr[c] = c * kWave + lane < K ? p.pos[c * kWave + lane] : inf;
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
#define CHUNK_VISITS_BEGIN     for (int pos = p0 - 1; pos <= p1; ++pos) { \
      double *st = L.stage0 + (pos & 1) * kWStage; \
      double *stn = L.stage0 + ((pos + 1) & 1) * kWStage; \
      const int hb = ((pos % 3) + 3) % 3, hb1 = (((pos - 1) % 3) + 3) % 3, hb2 = (((pos - 2) % 3) + 3) % 3; \
      double *hcur = L.hand + hb * 8 * kWS, *hprev = L.hand + hb1 * 8 * kWS, *hprev2 = L.hand + hb2 * 8 * kWS; \
      double *sc = L.scal + (pos & 1) * kScalDoubles; \
      const bool have_node = pos >= p0 && pos < p1; \
      long long tmark = p.prof ? (long long)__builtin_readcyclecounter() : 0; \
      const long long tvisit = tmark; \
      (void)st; (void)stn; (void)hcur; (void)hprev; (void)hprev2; (void)sc; (void)have_node; (void)tvisit;
#define CHUNK_VISITS_END_(BARRIER)       if (p.prof) { \
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
#define CHUNK_VISITS_END CHUNK_VISITS_END_(__syncthreads())
    if (wave < kChunkCompute) {
      const int off = wave * kWave + lane;   // this lane's label
      const bool mine = wave < C;            // the chunk exists
      const bool valid = off < K;
      const double posv = valid ? p.pos[off] : 0.0;
      const double dabs = fabs((double)(lane - kWPad) * ustep);  // |d step| for d = lane - kWPad
      // barrier among the C compute waves: LDS writes before it are visible after it
      auto csync = [&]() {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        arrivals += C;
        if (lane == 0) __hip_atomic_fetch_add(L.ctl + 3, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        int spins = 0;
        while (__hip_atomic_load(L.ctl + 3, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) - arrivals < 0) {
          __builtin_amdgcn_s_sleep(1);
          if (++spins > kSpinLimit) { if (lane == 0) L.ctl[1] = 1; break; }  // bounded
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
      };
      auto across = [&](const double *slot) {  // min over the chunks' slots
        double v = slot[0];
        if (C > 1) v = min_raw(v, slot[1]);
        if (C > 2) v = min_raw(v, slot[2]);
        if (C > 3) v = min_raw(v, slot[3]);
        return v;
      };
      CHUNK_VISITS_BEGIN
        // ======================================================== compute waves: one label chunk each
        if (UPDATE && have_node && mine) {
          const int *sti = (const int *)(st + kWStI);
          const int f = __builtin_amdgcn_readfirstlane(sti[2]);
          const int nout = f & 15, nin = (f >> 4) & 15, ntot = nout + nin;
          const int myrow = sti[72 + (lane & 7)];  // LDS offsets of the message rows (written by loader A)
          // Di = D + messages in list order (from the ring where the neighbour was one of the last two
          // visits of this run)
          // (every row is requested before the first addition -- a dependent LDS round trip costs
          //  a few hundred cycles here; a row beyond the node's degree reads row 0 and is not added)
          double di = st[off];
          double mrow[8];
#pragma unroll
          for (int jj = 0; jj < 8; ++jj) {
            const double *src = lds + __builtin_amdgcn_readlane(myrow, jj < ntot ? jj : 0);
            mrow[jj] = src[off];
          }
#pragma unroll
          for (int jj = 0; jj < 8; ++jj) {
            const double sum = di + mrow[jj];
            di = jj < ntot ? sum : di;
          }
          di = valid ? di : inf;
          if (BACKWARD) {
            const double lm = wave_min_dpp(di);
            if (lane == 0) L.redn[wave] = lm;
            csync();
            const double node_vmin = across(L.redn);
            if (wave == 0 && lane == 0) sc[8] = node_vmin;
            di -= node_vmin;
          }
          WSTAMP(0);
          const double gamma = st[kWStG];  // (double)1 / (double)max(n_out, n_in)
          const int w = p.window;
          for (int j0 = 0; j0 < nout; j0 += 4) {
            // ---- H_j = gamma Di - m_j of up to four messages into their tables, and the chunks' minima
            double alpha[4], mold[4];
#pragma unroll
            for (int jj = 0; jj < 4; ++jj) {  // (reads first)
              const int j = j0 + jj < nout ? j0 + jj : j0;
              alpha[jj] = st[kWS + 8 * kWS + j];
              mold[jj] = st[kWS + j * kWS + off];
            }
#pragma unroll
            for (int jj = 0; jj < 4; ++jj) {
              if (j0 + jj < nout) {
                double h = gamma * di - mold[jj];
                h = valid ? h : inf;
                if (valid) L.tab[jj * kCTab + kWPad + off] = h;
                const double lm = wave_min_dpp(h);
                if (lane == 0) L.redh[jj * 4 + wave] = lm;
              }
            }
            csync();
            WSTAMP(1);
            // ---- windowed min-plus (a source farther than lambda costs >= vTrunc exactly), normalised.
            // The smallest entry of the new message is min H_j itself: the d = 0 term of destination t
            // is h_t + alpha 0 = h_t, every other term is some h_s plus a non-negative cost, and
            // vTrunc = min H_j + alpha lambda is no smaller -- so no second reduction across the chunks.
#pragma unroll
            for (int jj = 0; jj < 4; ++jj) {
              if (j0 + jj < nout) {
                const int j = j0 + jj;
                const double a = alpha[jj];
                const double hmin = across(L.redh + jj * 4);
                const double vtrunc = hmin + a * p.lambda;
                const double *tb = L.tab + jj * kCTab + kWPad + off;
                double m1 = inf;
                if (uniform && w <= kWPad) {
                  // lane kWPad + d of `adm` holds alpha |d step| (== alpha |t - q| exactly); the loop
                  // takes it from there (two v_readlane) instead of recomputing it in every lane
                  const double adm = a * dabs;
                  if (w == 8) {
                    double hs[17];
#pragma unroll
                    for (int u = 0; u < 17; ++u) hs[u] = tb[u - 8];
#pragma unroll
                    for (int u = 0; u < 17; ++u) m1 = min_raw(m1, readlane_f64(adm, kWPad - 8 + u) + hs[u]);
                  } else {
                    for (int d = -w; d <= w; ++d) m1 = min_raw(m1, readlane_f64(adm, kWPad + d) + tb[d]);
                  }
                } else if (w <= kWPad) {
                  const double *pb = L.ptab + kWPad + off;
                  for (int d = -w; d <= w; ++d) m1 = min_raw(m1, pair_cost<1>(a, posv - pb[d], tb[d]));
                } else {
                  for (int d = -w; d <= w; ++d) {
                    const int i = off + d, ic = i < 0 ? 0 : i > K - 1 ? K - 1 : i;
                    const double cst = pair_cost<1>(a, posv - L.ptab[kWPad + ic], L.tab[jj * kCTab + kWPad + ic]);
                    m1 = min_raw(m1, (i >= 0 && i < K) ? cst : inf);
                  }
                }
                const double o = m1 < vtrunc ? m1 : vtrunc;
                if (valid) hcur[j * kWS + off] = o - hmin;
                if (BACKWARD && wave == 0 && lane == 0) sc[j] = hmin;
              }
            }
            WSTAMP(2);
            // (a second round reuses the tables and slots: every wave must be done reading them)
            if (j0 + 4 < nout) csync();
            WSTAMP(3);
          }
        }
      CHUNK_VISITS_END
    } else if (wave == kChunkCompute) {
      int wnext = desc[(size_t)p0 * DW + lane];
      CHUNK_VISITS_BEGIN
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
#pragma unroll
              for (int j = 0; j < 8; ++j)
                if (j < nx.nout) stn[kWS + j * kWS + c * kWave + lane] = mv[j][c];
            }
          }
          if (lane < 8) stn[kWS + 8 * kWS + lane] = av;
          if (lane == 0) stn[kWStG] = (double)1 / (double)(nx.nout > nx.nin ? nx.nout : nx.nin);
        }
      CHUNK_VISITS_END
    } else if (wave == kChunkCompute + 1) {
      int wnext = desc[(size_t)p0 * DW + lane];
      CHUNK_VISITS_BEGIN
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
      CHUNK_VISITS_END
    } else if (wave == kChunkCompute + 2) {
      CHUNK_VISITS_BEGIN
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
            st_sc1(p.done + pd.rank, epoch);
            if (pd.remote & (1 << 16)) st_sc1(p.peer_done0 + pd.pn[0], epoch);
            if (pd.remote & (1 << 17)) st_sc1(p.peer_done1 + pd.pn[1], epoch);
          }
        }
      CHUNK_VISITS_END
    } else {
      int xprev = 0, xprev2 = 0;
      CHUNK_VISITS_BEGIN
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
            CHUNK_VISITS_END
    }
#undef CHUNK_VISITS_BEGIN
#undef CHUNK_VISITS_END
#undef CHUNK_VISITS_END_
    if (p.timeline && tid == 0) p.timeline[((size_t)D * p.nruns[0] + run) * 2 + 1] = wall_clock64();
  }
#undef WSTAMP
  if (p.prof && lane == 0) {
    // slots [4 w .. 4 w + 3]: the four phases of compute wave w
    if (wave < kChunkCompute) for (int i = 0; i < 4; ++i) atomicAdd(p.prof + 4 * wave + i, pacc[i]);
    if (wave == 0) {
      atomicAdd(p.prof + 21, pwait);
      atomicAdd(p.prof + 22, pvis);
    }
    if (wave >= kChunkCompute) atomicAdd(p.prof + 16 + (wave - kChunkCompute), pbusy);
  }
}
template <int KERNEL, bool BACKWARD, bool PRIMAL, bool UPDATE>
__global__ __launch_bounds__(kChunkThreads) void trws_chunk_kernel(DevParams p, int epoch) {
  chunk_body<KERNEL, BACKWARD, PRIMAL, UPDATE>(p, epoch);
}
template <int KERNEL, bool BACKWARD, bool PRIMAL, bool UPDATE>
__global__ __launch_bounds__(kChunkThreads) void trws_chunk_group_kernel(GroupArgs ga, int epoch) {
  chunk_body<KERNEL, BACKWARD, PRIMAL, UPDATE>(ga.pp[group_strip(ga)], epoch);
}

}  // namespace

size_t chunk_lds_bytes() { return sizeof(double) * kChunkLdsDoubles; }

void chunk_set_attributes() {
  const int wlds = (int)chunk_lds_bytes();
#define SET_W(NAME)                                                                                                             \
  STEREO_HIP_CHECK(hipFuncSetAttribute((const void *)NAME<1, false, false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, wlds)); \
  STEREO_HIP_CHECK(hipFuncSetAttribute((const void *)NAME<1, true, false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, wlds));  \
  STEREO_HIP_CHECK(hipFuncSetAttribute((const void *)NAME<1, false, true, true>, hipFuncAttributeMaxDynamicSharedMemorySize, wlds));  \
  STEREO_HIP_CHECK(hipFuncSetAttribute((const void *)NAME<1, false, true, false>, hipFuncAttributeMaxDynamicSharedMemorySize, wlds))
  SET_W(trws_chunk_kernel); SET_W(trws_chunk_group_kernel);
#undef SET_W
}

#define CHUNK_SWITCH(NAME, ARG)                                                                                    \
  const size_t wlds = chunk_lds_bytes();                                                                           \
  const dim3 wgrid(blocks), wblock(kChunkThreads);                                                                 \
  switch (what) {                                                                                                  \
    case 0: hipLaunchKernelGGL((NAME<1, false, false, true>), wgrid, wblock, wlds, s, ARG, epoch); break;          \
    case 1: hipLaunchKernelGGL((NAME<1, true, false, true>), wgrid, wblock, wlds, s, ARG, epoch); break;           \
    case 2: hipLaunchKernelGGL((NAME<1, false, true, true>), wgrid, wblock, wlds, s, ARG, epoch); break;           \
    default: hipLaunchKernelGGL((NAME<1, false, true, false>), wgrid, wblock, wlds, s, ARG, epoch); break;         \
  }                                                                                                                \
  STEREO_HIP_CHECK(hipGetLastError());

void launch_chunk(int what, int blocks, hipStream_t s, const DevParams &p, int epoch) { CHUNK_SWITCH(trws_chunk_kernel, p) }
void launch_chunk_group(int what, int blocks, hipStream_t s, const GroupArgs &ga, int epoch) { CHUNK_SWITCH(trws_chunk_group_kernel, ga) }
#undef CHUNK_SWITCH

}  // namespace stereo
