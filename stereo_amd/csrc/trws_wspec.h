// The speculative schedule of the long serial run (trws_graph.h: Sweep::Spec) for trws_wide_kernel:
// 64 < K <= 256 labels on shared uniformly spaced positions, linear kernel, even K.  Same protocol as trws_spec.h has for
// trws_pipe_kernel -- a RUNNER walks the cut run computing nothing but the recurrence that makes it serial, the segments
// recompute every visit with the certified routine side by side and commit in order after comparing what they started
// from with what the segment in front really handed over: a wrong row of the runner's costs a second walk, never a bit --
// with the runner's roles laid out for four chunks of 64 labels:
//   waves 0-3   messages: wave w holds labels 64 w .. 64 w + 63, one per lane.  Di from the staged prefix sum and the
//               row(s) node i - 1 handed over (in the wave's registers), H = gamma Di - m into a table in LDS, the four
//               waves meet once per message (arrival words in LDS; twice in the backward sweep, where Di loses its
//               minimum first), then windowed min-plus over +-8 table entries: ~110 instructions per wave and visit
//               where a visit of the certified routine is ~700 on its longest wave;
//   wave 4      labels of the primal pass, four per lane (the primal wave's operations);
//   waves 5-10  loaders: wave l stages the nodes i = l (mod 6) into ring slot l: descriptor, foreign flags, unary and
//               message rows, prefix sum in list order;
//   wave 11     publisher: at every cut the rows / label of the node in front go to p.spec_rows / p.spec_x, drained, then
//               the segment's flag done[N + s].
// The runner's LDS OVERLAYS the visit loops' (it is called at kernel entry, before they set anything up).
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>

#include "trws_dev.h"

#define WRLI(v, i) __builtin_amdgcn_readlane((v), (i))

namespace stereo {
namespace {

constexpr int kWrSlots = 6, kWrLoaders = 6, kWrMsg = 4;
constexpr int kWrK = 256;   // row stride (labels)
// a staged node (doubles): prefix sum P | up to three staged rows of the tail | old rows of the (up to two) messages to
// compute | unary | the node's own (outgoing) rows | words
constexpr int kWrRowP = 0, kWrRowS = kWrK, kWrRowM = 4 * kWrK, kWrRowTH = 6 * kWrK, kWrRowOUT = 7 * kWrK, kWrSc = 11 * kWrK;
constexpr int kWrSlotDoubles = kWrSc + 16;
constexpr int kWrTab = kWrSlots * kWrSlotDoubles;   // 2 tables of 16 + 256 + 16 doubles: H with +inf on both sides
constexpr int kWrTabDoubles = kWrK + 32;
constexpr int kWrPm = kWrTab + 2 * kWrTabDoubles;   // partial minima of the four waves: [2 tables][4] for min H, [2][4] for min Di
constexpr int kWrPub = kWrPm + 16;                  // 2 x (2 rows): what the publisher stores
constexpr int kWrWords = kWrPub + 4 * kWrK;         // 64 ints
constexpr int kWrPos = kWrWords + 32 + 32;          // (behind the words and the 64 words where lanes that have nothing to say store) the positions
constexpr int kWrDoubles = kWrPos + kWrK;
// words: labels consumed 13 | label published x 2: 16 | slots freed 18 | label x 2: 20 | node x 2: 22 | row kinds x 2: 24 |
// rows published, per wave, x 2: 32-39 | arrival of the four message waves: 40-43 | nodes consumed by message wave w: 48-51
constexpr int kWwConsP = 13, kWwPubP = 16, kWwFree = 18, kWwLabel = 20, kWwNode = 22, kWwKinds = 24, kWwPubM = 32, kWwArrive = 40, kWwConsM = 48;
// words of a staged node: as in trws_spec.h (tail length | tail kinds | messages to compute | cut | kinds of the next
// segment's first rows | n_out | incoming rows | their label x 4 | direction bits | node | TAG = position + 1, written last);
// doubles 8-15: alpha x 2, gamma, alpha of the incoming rows x 4
constexpr int kWsNt = 0, kWsKinds = 1, kWsNmsg = 2, kWsCut = 5, kWsPubKinds = 6, kWsNout = 7, kWsNin = 8, kWsSrc = 9, kWsMd = 13, kWsNode = 14, kWsTag = 15;

// (words of the runner's LDS, addressed AS LDS: through a generic pointer these become flat accesses, whose completion is
//  counted together with the global loads -- a loader that polls a word would wait for the rows it has just requested
//  for its next node; the fences are LDS-only for the same reason)
typedef __attribute__((address_space(3))) int wr_lds_int;
__device__ __forceinline__ int wr_load(const int *w) { return __hip_atomic_load((const wr_lds_int *)w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
__device__ __forceinline__ void wr_store(int *w, int v) { __hip_atomic_store((wr_lds_int *)w, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
#define WR_ACQUIRE() __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local")
#define WR_RELEASE() __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local")

template <class T>
__device__ __forceinline__ T wr_uniform(T v) {
  static_assert(sizeof(T) == 4 || sizeof(T) == 8, "wr_uniform");
  if (sizeof(T) == 4) {
    int w;
    __builtin_memcpy(&w, &v, 4);
    w = __builtin_amdgcn_readfirstlane(w);
    __builtin_memcpy(&v, &w, 4);
  } else {
    int w[2];
    __builtin_memcpy(w, &v, 8);
    w[0] = __builtin_amdgcn_readfirstlane(w[0]); w[1] = __builtin_amdgcn_readfirstlane(w[1]);
    __builtin_memcpy(&v, w, 8);
  }
  return v;
}
__device__ __forceinline__ DevParams wr_params(const DevParams *pp) {
  DevParams q;
#define U(f) q.f = wr_uniform(pp->f)
  U(K); U(lambda); U(unary); U(msg); U(pos); U(alpha); U(x); U(done); U(abort_flag); U(spin_ticks); U(n_own); U(N);
  U(desc[0]); U(desc[1]); U(window); U(uniform_step); U(spec_c0[0]); U(spec_c0[1]); U(spec_c1[0]); U(spec_c1[1]);
  U(spec_len); U(spec_nseg); U(spec_max_len); U(spec_rows); U(spec_x); U(spec_undo); U(spec_stat); U(timeline); U(tl_stride); U(debug);
#undef U
  return q;
}

// Waits until *word >= want (words of the runner's own LDS).  What a role of the runner waits for ends, in the last
// instance, with a loader's wait for another workgroup, which is bounded by the wall clock and raises the abort word.
__device__ __attribute__((noinline)) bool wr_wait(const int *word, int want, int *abort_word, int32_t *abort_flag, long long spin_ticks) {
  int spins = 0;
  long long t0 = 0;
  while (wr_load(word) < want) {
    __builtin_amdgcn_s_sleep(1);
    spins = (spins + 1) & 1023;
    if (spins != 0) continue;
    if (wr_load(abort_word) || ld_sc1(abort_flag)) return false;
    const long long now = (long long)wall_clock64();
    if (t0 == 0) { t0 = now | 1; continue; }
    if (now - t0 > 4 * spin_ticks) { wr_store(abort_word, 1); return false; }
  }
  WR_ACQUIRE();
  return true;
}
// (the same, inlined: a call inside a loader's turn makes it spill the rows it holds in registers around the call)
__device__ __forceinline__ bool wr_wait_i(const int *word, int want, int *abort_word, int32_t *abort_flag, long long spin_ticks) {
  int spins = 0;
  long long t0 = 0;
  while (wr_load(word) < want) {
    spins = (spins + 1) & 4095;
    if (spins != 0) continue;
    if (wr_load(abort_word) || ld_sc1(abort_flag)) return false;
    const long long now = (long long)wall_clock64();
    if (t0 == 0) { t0 = now | 1; continue; }
    if (now - t0 > 4 * spin_ticks) { wr_store(abort_word, 1); return false; }
  }
  WR_ACQUIRE();
  return true;
}
// ... until all four of words[0 .. 3] >= want (16-byte aligned: the four words come with ONE LDS read per look)
__device__ __forceinline__ bool wr_wait4(const int *words, int want, int *abort_word, int32_t *abort_flag, long long spin_ticks) {
  typedef int wr_v4i __attribute__((ext_vector_type(4)));
  typedef __attribute__((address_space(3))) const volatile wr_v4i wr_lds_v4i;
  int spins = 0;
  long long t0 = 0;
  for (;;) {
    const wr_v4i v = *(wr_lds_v4i *)words;
    const int lo = min(min(v.x, v.y), min(v.z, v.w));
    if (lo >= want) break;
    spins = (spins + 1) & 4095;
    if (spins != 0) continue;
    if (wr_load(abort_word) || ld_sc1(abort_flag)) return false;
    const long long now = (long long)wall_clock64();
    if (t0 == 0) { t0 = now | 1; continue; }
    if (now - t0 > 4 * spin_ticks) { wr_store(abort_word, 1); return false; }
  }
  WR_ACQUIRE();
  return true;
}

// ---- waves 0-3: the message recurrence, 64 labels each -----------------------------------------------------------------
struct WrNodeM {
  int sw;
  double sd, P, S0, S1, S2, M0, M1;
};
__device__ __forceinline__ void wr_request_m(const double *sl, int lane, int k, WrNodeM &n) {
  n.sw = ((const int *)(sl + kWrSc))[lane & 15];
  asm volatile("" ::: "memory");   // (LDS serves a wave's requests in order: rows behind a tag that was found are the node's)
  n.sd = sl[kWrSc + 8 + (lane & 7)];
  n.P = sl[kWrRowP + k]; n.S0 = sl[kWrRowS + k]; n.S1 = sl[kWrRowS + kWrK + k]; n.S2 = sl[kWrRowS + 2 * kWrK + k];
  n.M0 = sl[kWrRowM + k]; n.M1 = sl[kWrRowM + kWrK + k];
}
struct WrArgsM {
  int K, c0, c1;
  double lambda, step;
  int32_t *abort_flag;
  long long spin_ticks;
  unsigned long long *stat;
};

// The four waves meet: everybody's table entries and partial minimum are written, then read.  `seq` counts the meetings.
__device__ __forceinline__ bool wr_meet(int *rw, int wv, int lane, int seq, int *abort_word, const WrArgsM &a, int dir) {
  const long long t0_ = a.stat ? (long long)wall_clock64() : 0;
  WR_RELEASE();
  if (lane == 0) wr_store(rw + kWwArrive + wv, seq);
  const bool ok = wr_wait4(rw + kWwArrive, seq, abort_word, a.abort_flag, a.spin_ticks);
  if (a.stat && wv == 0 && lane == 0) atomicAdd(a.stat + 16 + dir, (unsigned long long)((long long)wall_clock64() - t0_));
  return ok;
}

template <bool BACKWARD, int G>
__device__ __forceinline__ void wr_messages_g(const WrArgsM &a, double *rb, int lane, int wv, int *abort_word) {
  const double inf = __builtin_huge_val();
  const int k = wv * kWave + lane;   // this lane's label
  const bool act = k < a.K;
  int *rw = (int *)(rb + kWrWords);
  int *cons_word = lane == 0 ? rw + kWwConsM + wv : (int *)(rb + kWrWords + 32) + lane;   // (dummy words behind)
  double ad[4 * G];   // alpha |d step| of the window's index distances, for the weight seen last
#pragma unroll
  for (int d = 0; d < 4 * G; ++d) ad[d] = 0;
  double alpha_have = 0;
  double A0 = 0, A1 = 0;
  WrNodeM cur, nxt;
  wr_request_m(rb, lane, k, cur);
  nxt = cur;
  int slot_off = 0, seq = 0, tsel = 0, nsel = 0;
  for (int i = a.c0; i < a.c1; ++i) {
    if (__builtin_amdgcn_readlane(cur.sw, kWsTag) != i + 1) {   // (not there yet when it was asked for)
      const long long t0_ = (long long)wall_clock64();
      if (!__builtin_amdgcn_readfirstlane((int)wr_wait((const int *)(rb + slot_off + kWrSc) + kWsTag, i + 1, abort_word, a.abort_flag, a.spin_ticks))) return;
      wr_request_m(rb + slot_off, lane, k, cur);
      if (a.stat && wv == 0 && lane == 0) { atomicAdd(a.stat + 3 + 2 * (BACKWARD ? 1 : 0), 1ull); atomicAdd(a.stat + 4 + 2 * (BACKWARD ? 1 : 0), (unsigned long long)((long long)wall_clock64() - t0_)); }
    }
    const int sw = cur.sw;
    const int key = __builtin_amdgcn_readlane(sw, kWsKinds), nmsg = __builtin_amdgcn_readlane(sw, kWsNmsg), cut = __builtin_amdgcn_readlane(sw, kWsCut);
    double Di = cur.P;
    // the tail of the node's list from the first handed-over row on, in list order (the order of the reference's additions)
    if (key == 0x20098) { Di += A0; Di += A1; }
    else if (key == 0x30908) { Di += A0; Di += cur.S0; Di += A1; }
    else if (key == 0x30098) { Di += A0; Di += A1; Di += cur.S0; }
    else if (key == 0x41908) { Di += A0; Di += cur.S0; Di += A1; Di += cur.S1; }
    else {
      const int nt = key >> 16;
      for (int t = 0; t < nt; ++t) {
        const int kd = (key >> (4 * t)) & 15;
        if (kd == 8) Di += A0; else if (kd == 9) Di += A1; else if (kd == 0) Di += cur.S0; else if (kd == 1) Di += cur.S1; else Di += cur.S2;
      }
    }
    const double sd = cur.sd, mold0 = cur.M0, mold1 = cur.M1;
    // this node's words and rows are in registers: its place in the ring is free for this wave, the next node's are asked for
    wr_store(cons_word, i + 1 - a.c0);
    slot_off += kWrSlotDoubles;
    if (slot_off == kWrSlots * kWrSlotDoubles) slot_off = 0;
    if (i + 1 < a.c1) wr_request_m(rb + slot_off, lane, k, nxt);
    double *pm = rb + kWrPm;
    if (BACKWARD) {   // minimize.cpp:79-83: the node's own lower-bound term leaves Di
      const double part = wave_min_dpp(act ? Di : inf);
      if (lane == 0) pm[8 + 4 * nsel + wv] = part;
      if (!wr_meet(rw, wv, lane, ++seq, abort_word, a, BACKWARD ? 1 : 0)) return;
      Di -= min_raw(min_raw(pm[8 + 4 * nsel], pm[8 + 4 * nsel + 1]), min_raw(pm[8 + 4 * nsel + 2], pm[8 + 4 * nsel + 3]));
      nsel ^= 1;   // (two sets of partial minima in turn, like the tables: nobody writes what a wave one meeting behind still reads)
    }
    const double gamma = readlane_f64(sd, 2);
    double R0 = 0, R1 = 0;
    for (int m = 0; m < nmsg; ++m) {
      const double alpha = readlane_f64(sd, m);
      const double h = gamma * Di - (m == 0 ? mold0 : mold1);   // (labels beyond K: the loader staged -inf as their old message: h = +inf)
      double out = 0;   // (alpha == 0: typeStereoLinear.h:390-396, a constant row, normalised)
      if (alpha != 0) {
        double *tabl = rb + kWrTab + tsel * kWrTabDoubles + 16 + k;
        *tabl = h;
        const double part = wave_min_dpp(h);
        if (lane == 0) pm[4 * tsel + wv] = part;
        if (__builtin_expect(alpha != alpha_have, 0)) {
#pragma unroll
          for (int d = 0; d < 4 * G; ++d) ad[d] = alpha * ((double)(d + 1) * a.step);
          alpha_have = alpha;
        }
        if (!wr_meet(rw, wv, lane, ++seq, abort_word, a, BACKWARD ? 1 : 0)) return;
        const double hmin = min_raw(min_raw(pm[4 * tsel], pm[4 * tsel + 1]), min_raw(pm[4 * tsel + 2], pm[4 * tsel + 3]));
        const double vtrunc = hmin + alpha * a.lambda;
        double lo[4 * G], hi[4 * G];
#pragma unroll
        for (int d = 0; d < 4 * G; ++d) { lo[d] = tabl[-(d + 1)]; hi[d] = tabl[d + 1]; }
        double ma = h, mb = vtrunc;   // (two chains)
#pragma unroll
        for (int d = 0; d < 4 * G; ++d) { ma = min_raw(ma, lo[d] + ad[d]); mb = min_raw(mb, hi[d] + ad[d]); }
        out = min_raw(ma, mb) - hmin;
        tsel ^= 1;   // (the other table next: a wave two meetings ahead of a reader cannot exist)
      }
      if (m == 0) R0 = out; else R1 = out;
    }
    A0 = R0; A1 = nmsg == 2 ? R1 : R0;
    if (cut) {
      // the rows the segment behind this node starts from: to the publisher
      const int ps = cut & 1;
      if (__builtin_amdgcn_readfirstlane(wr_load(rw + kWwFree)) < cut - 2 &&
          !__builtin_amdgcn_readfirstlane((int)wr_wait(rw + kWwFree, cut - 2, abort_word, a.abort_flag, a.spin_ticks))) return;
      double *pb = rb + kWrPub + ps * 2 * kWrK;
      pb[k] = A0; pb[kWrK + k] = A1;
      if (wv == 0 && lane == 0) wr_store(rw + kWwKinds + ps, __builtin_amdgcn_readlane(sw, kWsPubKinds));
      WR_RELEASE();
      if (lane == 0) wr_store(rw + kWwPubM + 4 * ps + wv, cut);
    }
    cur = nxt;
  }
}

template <bool BACKWARD>
__device__ __attribute__((noinline)) void wr_messages(const DevParams *pp, int wv_, int abort_off_) {
  extern __shared__ __attribute__((aligned(16))) double wr_lds[];
  constexpr int D = BACKWARD ? 1 : 0;
  WrArgsM a;
  a.K = wr_uniform(pp->K); a.c0 = wr_uniform(pp->spec_c0[D]); a.c1 = wr_uniform(pp->spec_c1[D]);
  a.lambda = wr_uniform(pp->lambda); a.step = wr_uniform(pp->uniform_step);
  a.abort_flag = wr_uniform(pp->abort_flag); a.spin_ticks = wr_uniform(pp->spin_ticks); a.stat = wr_uniform(pp->timeline) ? wr_uniform(pp->spec_stat) : nullptr;   // (development counters: with STEREO_HIP_TRWS_TIMELINE only)
  const int window = wr_uniform(pp->window);
  int *abort_word = (int *)(wr_lds + __builtin_amdgcn_readfirstlane(abort_off_));
  const int lane = threadIdx.x & (kWave - 1), wv = __builtin_amdgcn_readfirstlane(wv_);
  if (window <= 4) wr_messages_g<BACKWARD, 1>(a, wr_lds, lane, wv, abort_word);
  else wr_messages_g<BACKWARD, 2>(a, wr_lds, lane, wv, abort_word);
}

// ---- wave 4: the labels of the primal pass (minimize.cpp:223-264, as the primal wave of a visit computes them) ------
template <bool BACKWARD>
__device__ __attribute__((noinline)) void wr_labels(const DevParams *pp_, int abort_off_) {
  extern __shared__ __attribute__((aligned(16))) double wr_lds[];
  const DevParams p = wr_params(pp_);
  double *rb = wr_lds;
  int *abort_word = (int *)(wr_lds + __builtin_amdgcn_readfirstlane(abort_off_));
  const int lane = threadIdx.x & (kWave - 1);
  const int c0 = p.spec_c0[BACKWARD ? 1 : 0], c1 = p.spec_c1[BACKWARD ? 1 : 0];
  const double inf = __builtin_huge_val();
  const int K = p.K;
  int *rw = (int *)(rb + kWrWords);
  double posk[4];
  bool act[4];
#pragma unroll
  for (int c = 0; c < 4; ++c) { act[c] = c * kWave + lane < K; posk[c] = act[c] ? p.pos[c * kWave + lane] : 0.0; }
  const double lambda = p.lambda;
  int xprev = 0;
  for (int i = c0; i < c1; ++i) {
    const double *sl = rb + ((i - c0) % kWrSlots) * kWrSlotDoubles;
    {
      const long long t0_ = (p.timeline && p.spec_stat) ? (long long)wall_clock64() : 0;
      if (!wr_wait_i((const int *)(sl + kWrSc) + kWsTag, i + 1, abort_word, p.abort_flag, p.spin_ticks)) return;
      if (p.timeline && p.spec_stat && lane == 0) atomicAdd(p.spec_stat + 18, (unsigned long long)((long long)wall_clock64() - t0_));
    }
    const int sw = ((const int *)(sl + kWrSc))[lane & 15];
    const double sd = sl[kWrSc + 8 + (lane & 7)];
    const int nout = __builtin_amdgcn_readlane(sw, kWsNout), nin = __builtin_amdgcn_readlane(sw, kWsNin), md = __builtin_amdgcn_readlane(sw, kWsMd),
              cut = __builtin_amdgcn_readlane(sw, kWsCut);
    double db[4], o[4][4];
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      db[c] = act[c] ? sl[kWrRowTH + c * kWave + lane] : 0.0;
#pragma unroll
      for (int j = 0; j < 4; ++j) o[j][c] = sl[kWrRowOUT + j * kWrK + c * kWave + lane];
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    if (lane == 0) wr_store(rw + kWwConsP, i + 1 - c0);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      if (j < nin) {
        const int src = __builtin_amdgcn_readlane(sw, kWsSrc + j);
        const int ks = src < 0 ? xprev : src;
        const double pks = rb[kWrPos + ks];
        const double aj = readlane_f64(sd, 3 + j);
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          const double d = ((md >> j) & 1) == 0 ? pks - posk[c] : posk[c] - pks;
          db[c] += aj * min_raw(fabs(d), lambda);
        }
      }
    }
    double di[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      di[c] = db[c];
      if (nout > 0) di[c] += o[0][c];
      if (nout > 1) di[c] += o[1][c];
      if (nout > 2) di[c] += o[2][c];
      if (nout > 3) di[c] += o[3][c];
      di[c] = act[c] ? di[c] : inf;
    }
    const double vbest = wave_min_dpp(min_raw(min_raw(di[0], di[1]), min_raw(di[2], di[3])));
    int bi = 0;
    bool found = false;
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      const unsigned long long hit = __builtin_amdgcn_ballot_w64(di[c] == vbest);
      if (!found && hit) { bi = c * kWave + __builtin_ctzll(hit); found = true; }
    }
    xprev = bi;
    if (cut) {
      const int ps = cut & 1;
      if (!wr_wait(rw + kWwFree, cut - 2, abort_word, p.abort_flag, p.spin_ticks)) return;
      if (lane == 0) { rw[kWwLabel + ps] = xprev; rw[kWwNode + ps] = __builtin_amdgcn_readlane(sw, kWsNode); }
      WR_RELEASE();
      if (lane == 0) wr_store(rw + kWwPubP + ps, cut);
    }
  }
}

// ---- waves 5-10: staging -------------------------------------------------------------------------------------------------
template <bool BACKWARD, bool PRIMAL, bool UPDATE>
__device__ __attribute__((noinline)) void wr_loader(const DevParams *pp_, int epoch_, int abort_off_, int lw_) {
  extern __shared__ __attribute__((aligned(16))) double wr_lds[];
  const DevParams p = wr_params(pp_);
  double *rb = wr_lds;
  int *abort_word = (int *)(wr_lds + __builtin_amdgcn_readfirstlane(abort_off_));
  const int lane = threadIdx.x & (kWave - 1);
  const int epoch = __builtin_amdgcn_readfirstlane(epoch_), lw = __builtin_amdgcn_readfirstlane(lw_);
  constexpr int D = BACKWARD ? 1 : 0;
  const int c0 = p.spec_c0[D], c1 = p.spec_c1[D];
  constexpr int DW = TrwsGraph::kDescWords;
  const int32_t *desc = p.desc[D];
  const int K = p.K;
  int *rw = (int *)(rb + kWrWords);
  const int L = p.spec_len, nseg = p.spec_nseg;
  double *sl = rb + lw * kWrSlotDoubles;   // (six loaders, six slots: node i lives in slot i mod 6 = this wave's)
  // The node's own data -- unary row and its (up to four) outgoing rows, which nobody writes before the segment that holds
  // the node walks it -- is requested ONE TURN AHEAD (this wave's next node, six positions on) and waits in registers
  // while the current node is staged: a loader's turn used to be one memory round trip longer than the six visits it has.
  // (no load below is conditional on the lane: a label index beyond K reads label K - 1 again -- rows of such labels are
  //  never looked at --, so that the requests of a node go out back to back instead of one exec-mask region each)
  bool ok[4];
  int lk[4];
#pragma unroll
  for (int c = 0; c < 4; ++c) { ok[c] = c * kWave + lane < K; lk[c] = ok[c] ? c * kWave + lane : K - 1; }
  double theta_n[4] = {0, 0, 0, 0}, own_n[4][4];
#pragma unroll
  for (int k = 0; k < 4; ++k)
#pragma unroll
    for (int c = 0; c < 4; ++c) own_n[k][c] = 0;
#define WR_REQUEST_OWN(W)                                                                                                        \
  do {                                                                                                                         \
    const int nout_ = WRLI((W), 2) & 15;                                                                                       \
    const double *ua_ = p.unary + (size_t)((unsigned long long)(unsigned)WRLI((W), 0) * (unsigned long long)(unsigned)K);      \
    _Pragma("unroll") for (int c = 0; c < 4; ++c) theta_n[c] = ua_[lk[c]];                                                     \
    _Pragma("unroll") for (int k = 0; k < 4; ++k) {                                                                            \
      if (k < nout_) {                                                                                                         \
        const double *mb_ = p.msg + (size_t)((unsigned long long)(unsigned)WRLI((W), 4 + k) * (unsigned long long)(unsigned)K); \
        _Pragma("unroll") for (int c = 0; c < 4; ++c) own_n[k][c] = mb_[lk[c]];                                                \
      }                                                                                                                        \
    }                                                                                                                          \
  } while (0)
  int w2 = 0, wn2 = 0;
  if (c0 + lw < c1) { w2 = desc[(size_t)(c0 + lw) * DW + lane]; wn2 = c0 + lw + 1 < c1 ? desc[(size_t)(c0 + lw + 1) * DW + lane] : 0; WR_REQUEST_OWN(w2); }
  for (int i = c0 + lw; i < c1; i += kWrLoaders) {
    const long long tl0_ = (p.timeline && p.spec_stat && lw == 0) ? (long long)wall_clock64() : 0;
    const int w = w2, wn = wn2;
    double theta[4], r[8][4];
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      theta[c] = theta_n[c];
#pragma unroll
      for (int k = 0; k < 4; ++k) r[k][c] = own_n[k][c];
#pragma unroll
      for (int k = 4; k < 8; ++k) r[k][c] = 0;
    }
    if (i + kWrLoaders < c1) {
      w2 = desc[(size_t)(i + kWrLoaders) * DW + lane];
      wn2 = i + kWrLoaders + 1 < c1 ? desc[(size_t)(i + kWrLoaders + 1) * DW + lane] : 0;
      WR_REQUEST_OWN(w2);
    }
    const int f = WRLI(w, 2), fn = WRLI(wn, 2);
    const int nout = f & 15, nin = (f >> 4) & 15, ndep = (f >> 8) & 15, md = (f >> 16) & 255, ntot = nout + nin;
    const int noutn = fn & 15, ntotn = noutn + ((fn >> 4) & 15);
    const int j8 = lane & 7;
    const int slw = __shfl(w, 12 + j8, kWave), sln = __shfl(wn, 12 + j8, kWave);
    // rows the node in front hands over (this node's and, for the messages to compute here, the next node's)
    const int fr = i > c0 ? (int)(__builtin_amdgcn_ballot_w64(lane < 8 && lane >= nout && lane < ntot && slw >= 0) & 255ull) : 0;
    const int frn = i + 1 < c1 ? (int)(__builtin_amdgcn_ballot_w64(lane < 8 && lane >= noutn && lane < ntotn && sln >= 0) & 255ull) : 0;
    const int kfirst = fr ? __builtin_ctz(fr) : ntot;
    int s0 = -1, s1 = -1;   // slots, in this node's outgoing list, of the (up to two distinct) messages the next node takes from it
    {
      int rest = frn;
      while (rest) {
        const int k = __builtin_ctz(rest);
        rest &= rest - 1;
        const int s = __builtin_amdgcn_readlane(wn, 12 + k);
        if (s0 < 0 || s0 == s) s0 = s; else s1 = s;
      }
    }
    int p0s = -1;
    if (fr) p0s = __builtin_amdgcn_readlane(w, 12 + kfirst);
    int kinds = 0, nt = 0, nstaged = 0, stage_of[4] = {-1, -1, -1, -1};
    for (int k = kfirst; k < ntot; ++k) {
      int kd;
      if ((fr >> k) & 1) kd = __builtin_amdgcn_readlane(w, 12 + k) == p0s ? 8 : 9;
      else { kd = nstaged; if (nstaged < 4) stage_of[nstaged] = k; ++nstaged; }
      kinds |= kd << (4 * nt);
      ++nt;
    }
    int cut = 0, pubkinds = 0;
    if (i + 1 < c1 && (i + 1 - c0) % L == 0 && (i + 1 - c0) / L < nseg) {
      cut = (i + 1 - c0) / L;
      for (int k = noutn; k < ntotn; ++k)
        if ((frn >> k) & 1) pubkinds |= (__builtin_amdgcn_readlane(wn, 12 + k) == s0 ? 8 : 9) << (4 * k);
    }
    // the node's own data (nobody writes it before the segment that holds the node walks it)
    const int fm = WRLI(w, kDescFetch) & 255;
    const double av = p.alpha[__shfl(w, 4 + j8, kWave)];
#pragma unroll
    for (int k = 4; k < 8; ++k) {   // (own rows five to eight: no node of a cut run has them; fetched here for completeness)
      if (k < nout) {
        const double *mb = p.msg + (size_t)((unsigned long long)(unsigned)WRLI(w, 4 + k) * (unsigned long long)(unsigned)K);
#pragma unroll
        for (int c = 0; c < 4; ++c) r[k][c] = mb[lk[c]];
      }
    }
    // foreign dependencies (everything but the node in front), then their rows and labels
    if (ndep > 0) wait_for_dependencies_w(p, ndep, __shfl(w, 20 + (lane & 3), kWave), WRLI(w, 1), epoch, lane, abort_word);
    if (wr_load(abort_word)) return;
    int src = -1;   // lane k < nin: the label the k-th incoming row's pairwise term takes (-1: the node in front)
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      if (k >= nout && k < ntot && ((fm >> k) & 1) && !((fr >> k) & 1)) {
        if (UPDATE) {
          const double *mb = p.msg + (size_t)((unsigned long long)(unsigned)WRLI(w, 4 + k) * (unsigned long long)(unsigned)K);
#pragma unroll
          for (int c = 0; c < 4; ++c) r[k][c] = ld_sc1(mb + lk[c]);
        }
        if (PRIMAL) { const int xv = ld_sc1(p.x + WRLI(w, 32 + k)); if (lane == k - nout) src = xv; }
      }
    }
    // the ring slot: all recurrences have taken the node that had it into their registers
    const int need = i - c0 - kWrSlots + 1;
    const long long tl1_ = (p.timeline && p.spec_stat && lw == 0) ? (long long)wall_clock64() : 0;
    if (UPDATE && need > 0 && !wr_wait4(rw + kWwConsM, need, abort_word, p.abort_flag, p.spin_ticks)) return;
    if (PRIMAL && need > 0 && !wr_wait_i(rw + kWwConsP, need, abort_word, p.abort_flag, p.spin_ticks)) return;
    const long long tl2_ = (p.timeline && p.spec_stat && lw == 0) ? (long long)wall_clock64() : 0;
    // (a row picked by a uniform index: one scalar branch tree per row instead of a select per candidate and chunk)
#define WR_GET(KSEL, DST)                                                                         \
    do {                                                                                          \
      switch (KSEL) {                                                                             \
        case 0: _Pragma("unroll") for (int c = 0; c < 4; ++c) DST[c] = r[0][c]; break;            \
        case 1: _Pragma("unroll") for (int c = 0; c < 4; ++c) DST[c] = r[1][c]; break;            \
        case 2: _Pragma("unroll") for (int c = 0; c < 4; ++c) DST[c] = r[2][c]; break;            \
        case 3: _Pragma("unroll") for (int c = 0; c < 4; ++c) DST[c] = r[3][c]; break;            \
        case 4: _Pragma("unroll") for (int c = 0; c < 4; ++c) DST[c] = r[4][c]; break;            \
        case 5: _Pragma("unroll") for (int c = 0; c < 4; ++c) DST[c] = r[5][c]; break;            \
        case 6: _Pragma("unroll") for (int c = 0; c < 4; ++c) DST[c] = r[6][c]; break;            \
        default: _Pragma("unroll") for (int c = 0; c < 4; ++c) DST[c] = r[7][c]; break;           \
      }                                                                                           \
    } while (0)
    double m0[4] = {0, 0, 0, 0}, m1[4] = {0, 0, 0, 0};
    if (UPDATE) {
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        double P = theta[c];
#pragma unroll
        for (int k = 0; k < 8; ++k)
          if (k < kfirst) P += r[k][c];
        sl[kWrRowP + c * kWave + lane] = P;
      }
      for (int t = 0; t < 3; ++t) {   // (staged rows of the tail: none on an ordinary node of the run)
        if (t < nstaged) {
          double v[4];
          WR_GET(stage_of[t], v);
#pragma unroll
          for (int c = 0; c < 4; ++c) sl[kWrRowS + t * kWrK + c * kWave + lane] = v[c];
        }
      }
      if (s0 >= 0) {   // (labels beyond K: -inf, which makes the recurrence's H = gamma Di - m = +inf there)
        WR_GET(s0, m0);
#pragma unroll
        for (int c = 0; c < 4; ++c) sl[kWrRowM + c * kWave + lane] = ok[c] ? m0[c] : -__builtin_huge_val();
      }
      if (s1 >= 0) {
        WR_GET(s1, m1);
#pragma unroll
        for (int c = 0; c < 4; ++c) sl[kWrRowM + kWrK + c * kWave + lane] = ok[c] ? m1[c] : -__builtin_huge_val();
      }
    }
    if (PRIMAL) {
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        const int kk = c * kWave + lane;
        sl[kWrRowTH + kk] = theta[c];
#pragma unroll
        for (int k = 0; k < 4; ++k)
          if (k < nout) sl[kWrRowOUT + k * kWrK + kk] = r[k][c];
      }
    }
    // the two messages to the next node are ONE message if weights and old rows agree (positions are shared)
    int nmsg = s0 < 0 ? 0 : 1;
    if (UPDATE && s1 >= 0) {
      bool differ = false;
#pragma unroll
      for (int c = 0; c < 4; ++c) differ = differ | (ok[c] & (__double_as_longlong(m0[c]) != __double_as_longlong(m1[c])));
      const bool same = WRLI(__double2hiint(av), s0) == WRLI(__double2hiint(av), s1) && WRLI(__double2loint(av), s0) == WRLI(__double2loint(av), s1) && !UNI(differ);
      nmsg = same ? 1 : 2;
    }
#undef WR_GET
    {
      int word = 0;
      word = lane == kWsNt ? nt : lane == kWsKinds ? (kinds | (nt << 16)) : lane == kWsNmsg ? nmsg : lane == kWsCut ? cut
           : lane == kWsPubKinds ? pubkinds : lane == kWsNout ? nout : lane == kWsNin ? nin : lane == kWsMd ? (md >> nout) : lane == kWsNode ? WRLI(w, 0) : 0;
      const int srck = __shfl(src, lane - kWsSrc, kWave);
      if (lane >= kWsSrc && lane < kWsSrc + 4) word = srck;
      if (lane < kWsTag) ((int *)(sl + kWrSc))[lane] = word;
      const double a0 = readlane_f64(av, s0 < 0 ? 0 : s0), a1 = readlane_f64(av, s1 < 0 ? (s0 < 0 ? 0 : s0) : s1);
      const double ain = __shfl(av, nout + (lane - 3 < 0 ? 0 : lane - 3), kWave);
      const double g = (double)1 / (double)(nout > nin ? nout : nin > 0 ? nin : 1);
      if (lane < 8) sl[kWrSc + 8 + lane] = lane == 0 ? a0 : lane == 1 ? a1 : lane == 2 ? g : ain;
    }
    WR_RELEASE();
    if (lane == 0) wr_store((int *)(sl + kWrSc) + kWsTag, i + 1);
    if (p.timeline && p.spec_stat && lw == 0 && lane == 0) {
      const long long tl3_ = (long long)wall_clock64();
      atomicAdd(p.spec_stat + 21 + D, (unsigned long long)(tl1_ - tl0_)); atomicAdd(p.spec_stat + 19 + D, (unsigned long long)(tl2_ - tl1_));
      atomicAdd(p.spec_stat + 23 + D, (unsigned long long)(tl3_ - tl2_));
    }
  }
#undef WR_REQUEST_OWN
}

// ---- wave 11: what a segment starts from, to global memory ------------------------------------------------------------
template <bool PRIMAL, bool UPDATE>
__device__ __attribute__((noinline)) void wr_publisher(const DevParams *pp_, int epoch_, int abort_off_) {
  extern __shared__ __attribute__((aligned(16))) double wr_lds[];
  const DevParams p = wr_params(pp_);
  double *rb = wr_lds;
  int *abort_word = (int *)(wr_lds + __builtin_amdgcn_readfirstlane(abort_off_));
  const int lane = threadIdx.x & (kWave - 1);
  const int epoch = __builtin_amdgcn_readfirstlane(epoch_);
  const int K = p.K;
  int *rw = (int *)(rb + kWrWords);
  for (int s = 1; s < p.spec_nseg; ++s) {
    const int ps = s & 1;
    if (UPDATE) {
      if (!wr_wait4(rw + kWwPubM + 4 * ps, s, abort_word, p.abort_flag, p.spin_ticks)) return;
      const double *pb = rb + kWrPub + ps * 2 * kWrK;
      const int kinds = wr_load(rw + kWwKinds + ps);
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        const int kk = c * kWave + lane;
        if (kk < K) {
          double a0 = pb[kk], a1 = pb[kWrK + kk];
          if ((p.debug & 16384) && s % 3 == 1 && kk == 0) a0 = __longlong_as_double(__double_as_longlong(a0) ^ 1ll);   // (development: a wrong row)
          if ((p.debug & 65536) && s % 3 == 1 && (kk & 1)) a0 += 0.375;                                                // (development: a VERY wrong row)
#pragma unroll
          for (int k = 0; k < 8; ++k) {
            const int kd = (kinds >> (4 * k)) & 15;
            if (kd) st_sc1(p.spec_rows + ((size_t)s * 8 + k) * K + kk, kd == 8 ? a0 : a1);
          }
        }
      }
    }
    if (PRIMAL) {
      if (!wr_wait(rw + kWwPubP + ps, s, abort_word, p.abort_flag, p.spin_ticks)) return;
      int label = wr_load(rw + kWwLabel + ps);
      if ((p.debug & 32768) && s % 5 == 2) label = label > 0 ? label - 1 : (K > 1 ? 1 : 0);                          // (development: a wrong label)
      if (lane == 0) st_sc1(p.spec_x + s, label);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (lane == 0) { st_sc1(p.done + p.N + s, epoch); wr_store(rw + kWwFree, s); }
  }
}

// (noinline, parameters through a pointer to their copy in global memory: nothing of this routine leaks into the kernel
//  it is called from; called at kernel entry, where only the kernel arguments are live)
template <bool BACKWARD, bool PRIMAL, bool UPDATE>
__device__ __attribute__((noinline)) void wide_chain_runner(const DevParams *pp_, int epoch_) {
  extern __shared__ __attribute__((aligned(16))) double wr_lds[];
  const DevParams &p = *pp_;
  const int epoch = epoch_;
  double *rb = wr_lds;
  constexpr int abort_off = kWrWords + 31;   // (the last double of the words: [0] abort)
  int *abort_word = (int *)(wr_lds + abort_off);
  constexpr int D = BACKWARD ? 1 : 0;
  const int tid = threadIdx.x;
  const int wave = __builtin_amdgcn_readfirstlane(tid / kWave);
  const int c0 = p.spec_c0[D], c1 = p.spec_c1[D];
  int *rw = (int *)(rb + kWrWords);
  if (tid < 64) rw[tid] = 0;
  if (tid < kWrSlots) ((int *)(rb + tid * kWrSlotDoubles + kWrSc))[kWsTag] = 0;   // (no node staged)
  if (tid < 64) {   // +inf on both sides of both tables
    const int t = tid >> 5, e = tid & 31;
    rb[kWrTab + t * kWrTabDoubles + (e < 16 ? e : kWrK + e)] = __builtin_huge_val();
  }
  for (int k = tid; k < kWrK; k += (int)blockDim.x) rb[kWrPos + k] = k < p.K ? p.pos[k] : 0.0;
  if (p.timeline && tid == 0) p.timeline[((size_t)2 * p.tl_stride + D) * 2] = wall_clock64();
  __syncthreads();
  const unsigned long long trole0 = wall_clock64();
  if (wave < kWrMsg) { if (UPDATE) { __builtin_amdgcn_s_setprio(3); wr_messages<BACKWARD>(pp_, wave, abort_off); __builtin_amdgcn_s_setprio(0); } }
  else if (wave == kWrMsg) { if (PRIMAL) { __builtin_amdgcn_s_setprio(3); wr_labels<BACKWARD>(pp_, abort_off); __builtin_amdgcn_s_setprio(0); } }
  else if (wave < kWrMsg + 1 + kWrLoaders) wr_loader<BACKWARD, PRIMAL, UPDATE>(pp_, epoch, abort_off, wave - kWrMsg - 1);
  else wr_publisher<PRIMAL, UPDATE>(pp_, epoch, abort_off);
  // (development: when each role was done, 100 MHz ticks since the roles started -- messages, labels, last loader, publisher)
  if (p.timeline && p.spec_stat && (tid & (kWave - 1)) == 0 && (wave == 0 || wave == kWrMsg || wave == kWrMsg + kWrLoaders || wave == kWrMsg + 1 + kWrLoaders))
    p.spec_stat[8 + 4 * D + (wave == 0 ? 0 : wave == kWrMsg ? 1 : wave == kWrMsg + kWrLoaders ? 2 : 3)] = wall_clock64() - trole0;
  __syncthreads();
  if (wr_load(abort_word) && tid == 0) st_sc1(p.abort_flag, 1);
  if (p.timeline && tid == 0) p.timeline[((size_t)2 * p.tl_stride + D) * 2 + 1] = wall_clock64();
  if (p.spec_stat && tid == 0) atomicAdd(p.spec_stat + 2, (unsigned long long)(c1 - c0));
  __syncthreads();
}

}  // namespace
}  // namespace stereo
