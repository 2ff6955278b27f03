// The RUNNER of the speculative schedule (trws_graph.h: Sweep::Spec), for trws_pipe_kernel with shared
// uniformly spaced positions and the linear kernel.  One workgroup draws the runner's ticket and walks the whole cut run
// computing nothing but the recurrence that makes the run serial:
//   wave 0      messages: Di of node i from the staged prefix sum and the row(s) node i - 1 handed over, H = gamma Di - m,
//               plain windowed min-plus (what message_regs returns whenever its certificate holds), the row for node i + 1
//               stays in registers -- no certificate, no tangency keys, no barrier, ~100 instructions per visit;
//   wave 1      labels of the primal pass: x_i from x_(i-1), exact (the same operations as the primal wave of a visit);
//   waves 2-9   loaders: descriptor of node i, its foreign flags, unary + message rows, prefix sum in list order, into a
//               ring of kRunSlots staged nodes in LDS (node i belongs to loader i mod 8);
//   wave 10     publisher: at every cut, the rows / label of the node in front go to p.spec_rows / p.spec_x, drained, then
//               the segment's flag done[N + s].
// Nothing here decides a result: the segments recompute every visit with the certified routine and compare what they
// started from with what the segment in front produced (pipe_body's commit).  A wrong row costs a second walk of one or
// more segments, never a wrong bit.
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>

#include "trws_dev.h"

namespace stereo {
namespace {

constexpr int kRunSlots = 10;
constexpr int kRunLoaders = 8;
// a staged node (doubles): prefix sum P | up to three staged rows of the tail | old rows of the (up to two) messages to
// compute | unary | the node's own (outgoing) rows | words
constexpr int kRunRowP = 0, kRunRowS = 64, kRunRowM = 256, kRunRowTH = 384, kRunRowOUT = 448, kRunSc = 704;
constexpr int kRunSlotDoubles = kRunSc + 16;
constexpr int kRunTab = kRunSlots * kRunSlotDoubles;   // 16 + 64 + 16: H with +inf on both sides
constexpr int kRunPub = kRunTab + 96;                  // 2 x (2 rows): what the publisher stores
constexpr int kRunWords = kRunPub + 256;               // 64 ints (below)
constexpr int kRunDoubles = kRunWords + 32;
constexpr int kRunDummy = 32;                           // 64 more words: where the lanes that have nothing to say store (run_visit_m)
// words: consumed by messages, by labels | publisher flags (rows x 2, labels x 2) | slots freed | label x 2 | node x 2 |
//        row kinds x 2
constexpr int kRwConsM = 12, kRwConsP = 13, kRwPubM = 14, kRwPubP = 16, kRwFree = 18, kRwLabel = 20, kRwNode = 22, kRwKinds = 24;
// words of a staged node: tail length | tail kinds (a nibble each: 0-2 staged row, 8 / 9 first / second handed-over row;
// the length once more in bits 16-)
// | messages to compute (0, 1, 2) | cut: segment that starts behind this node (0: none) | kinds of that segment's first
// node's rows (a nibble per row) | n_out | incoming rows | their label (-1: the node in front) x 4 | their direction
// bits | node id | word 15: the TAG, schedule position + 1, written last (behind a release fence): a reader that
// finds it finds the node.  doubles 8-15: alpha x 2, gamma, alpha of the incoming rows x 4
constexpr int kRsNt = 0, kRsKinds = 1, kRsNmsg = 2, kRsCut = 5, kRsPubKinds = 6, kRsNout = 7, kRsNin = 8, kRsSrc = 9, kRsMd = 13,
              kRsNode = 14, kRsTag = 15;

__device__ __forceinline__ int lds_load(const int *w) { return __hip_atomic_load(w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
__device__ __forceinline__ void lds_store(int *w, int v) { __hip_atomic_store(w, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }

// Waits until *word >= want.  Everything the runner's waves wait for inside the workgroup ends with a loader's wait for
// another workgroup, which is bounded by the wall clock and raises the abort word when it gives up; this wait looks at
// that word (and, as a last resort against a hang, at the clock itself).
__device__ __forceinline__ bool run_wait(const DevParams &p, const int *word, int want, int *abort_word) {
  int spins = 0;
  long long t0 = 0;
  while (lds_load(word) < want) {
    __builtin_amdgcn_s_sleep(1);
    spins = (spins + 1) & 1023;
    if (spins != 0) continue;
    if (lds_load(abort_word) || ld_sc1(p.abort_flag)) return false;
    const long long now = (long long)wall_clock64();
    if (t0 == 0) { t0 = now | 1; continue; }
    if (now - t0 > 4 * p.spin_ticks) { lds_store(abort_word, 1); return false; }
  }
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
  return true;
}

#define RLI(v, i) __builtin_amdgcn_readlane((v), (i))

// Arguments of a real (non-inlined) device function arrive in vector registers and the compiler must take them, and
// everything loaded through them, for lane-varying: every branch on such a value becomes an exec-mask region, every loop
// bound a mask loop.  They ARE uniform: the fields the runner and the commit use are read once and declared so.
template <class T>
__device__ __forceinline__ T uniform_value(T v) {
  static_assert(sizeof(T) == 4 || sizeof(T) == 8, "uniform_value");
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
__device__ __forceinline__ DevParams uniform_params(const DevParams *pp) {
  DevParams q;
#define U(f) q.f = uniform_value(pp->f)
  U(K); U(lambda); U(unary); U(msg); U(pos); U(alpha); U(x); U(done); U(abort_flag); U(spin_ticks); U(n_own); U(N);
  U(desc[0]); U(desc[1]); U(window); U(uniform_step); U(spec_c0[0]); U(spec_c0[1]); U(spec_c1[0]); U(spec_c1[1]);
  U(spec_len); U(spec_nseg); U(spec_max_len); U(spec_rows); U(spec_x); U(spec_undo); U(spec_stat); U(timeline); U(tl_stride); U(debug);
#undef U
  return q;
}

// ---- wave 0: the message recurrence ------------------------------------------------------------------------------
// What a visit reads from its staged node, requested in one go and WITHOUT waiting for the node to be there: the words
// first (tag included: one instruction, one snapshot), the rows behind a compiler barrier -- LDS serves a wave's
// requests in order, so rows that follow a tag that was found are the node's.  The next node's requests go out while
// this node's window is computed; a tag that was not there yet is waited for at the top of the next visit.
struct RunNodeM {
  int sw;
  double sd, P, S0, S1, S2, M0, M1;
};
__device__ __forceinline__ void run_request_m(const double *sl, int lane, RunNodeM &n) {
  n.sw = ((const int *)(sl + kRunSc))[lane & 15];
  asm volatile("" ::: "memory");
  n.sd = sl[kRunSc + 8 + (lane & 7)];
  n.P = sl[kRunRowP + lane]; n.S0 = sl[kRunRowS + lane]; n.S1 = sl[kRunRowS + 64 + lane]; n.S2 = sl[kRunRowS + 128 + lane];
  n.M0 = sl[kRunRowM + lane]; n.M1 = sl[kRunRowM + 64 + lane];
}
// What the routine needs of the launch parameters, as uniform values (read once by chain_runner).
struct RunArgsM {
  int K, window, c0, c1;
  double lambda, step;
  int32_t *abort_flag;
  long long spin_ticks;
  unsigned long long *stat;
};

// Waits until *word >= want (the slow path of a visit that found its node not staged yet).
__device__ __attribute__((noinline)) bool run_wait_m(const int *word, int want, int *abort_word, int32_t *abort_flag, long long spin_ticks) {
  int spins = 0;
  long long t0 = 0;
  while (lds_load(word) < want) {
    __builtin_amdgcn_s_sleep(1);
    spins = (spins + 1) & 1023;
    if (spins != 0) continue;
    if (lds_load(abort_word) || ld_sc1(abort_flag)) return false;
    const long long now = (long long)wall_clock64();
    if (t0 == 0) { t0 = now | 1; continue; }
    if (now - t0 > 4 * spin_ticks) { lds_store(abort_word, 1); return false; }
  }
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
  return true;
}

// One visit of the recurrence.  `cur`: the node's words and rows (asked for during the previous visit), `nxt`: where the
// next node's go.  Everything that steers the visit is a scalar; the lanes only ever do the arithmetic.
template <bool BACKWARD, int G>
__device__ __forceinline__ bool run_visit_m(const RunArgsM &a, double *rb, int *rw, double *tabl, int *cons_word, int lane, bool act, int i, int &slot_off,
                                            RunNodeM &cur, RunNodeM &nxt, double &A0, double &A1, double (&ad)[4 * G], double &alpha_have,
                                            int *abort_word) {
  const double inf = __builtin_huge_val();
  if (__builtin_amdgcn_readlane(cur.sw, kRsTag) != i + 1) {   // (not there yet when it was asked for)
    const long long t0_ = (long long)wall_clock64();
    // (a real call's result arrives in a vector register: say that it is uniform, or the loop gets exec-mask exits)
    if (!__builtin_amdgcn_readfirstlane((int)run_wait_m((const int *)(rb + slot_off + kRunSc) + kRsTag, i + 1, abort_word, a.abort_flag, a.spin_ticks))) return false;
    run_request_m(rb + slot_off, lane, cur);
    if (a.stat && lane == 0) { atomicAdd(a.stat + 3 + 2 * (BACKWARD ? 1 : 0), 1ull); atomicAdd(a.stat + 4 + 2 * (BACKWARD ? 1 : 0), (unsigned long long)((long long)wall_clock64() - t0_)); }
  }
  const int sw = cur.sw;
  const int key = __builtin_amdgcn_readlane(sw, kRsKinds), nmsg = __builtin_amdgcn_readlane(sw, kRsNmsg), cut = __builtin_amdgcn_readlane(sw, kRsCut);
  double Di = cur.P;
  // the tail of the node's list from the first handed-over row on, in list order (the order of the reference's additions)
  if (key == 0x20098) { Di += A0; Di += A1; }                                   // handed over, handed over (forward chain)
  else if (key == 0x30908) { Di += A0; Di += cur.S0; Di += A1; }                // ... with a row of another run in between
  else if (key == 0x30098) { Di += A0; Di += A1; Di += cur.S0; }
  else if (key == 0x41908) { Di += A0; Di += cur.S0; Di += A1; Di += cur.S1; }
  else {
    const int nt = key >> 16;
    for (int t = 0; t < nt; ++t) {
      const int kd = (key >> (4 * t)) & 15;
      if (kd == 8) Di += A0; else if (kd == 9) Di += A1; else if (kd == 0) Di += cur.S0; else if (kd == 1) Di += cur.S1; else Di += cur.S2;
    }
  }
  if (BACKWARD) Di -= wave_min_dpp(act ? Di : inf);   // minimize.cpp:79-83 (the node's own lower-bound term)
  const double sd = cur.sd, mold0 = cur.M0, mold1 = cur.M1;
  const double gamma = readlane_f64(sd, 2);
  // this node's words and rows are in registers: its place in the ring is free (one store instruction without an
  // exec-mask region: lane 0 hits the word, the other lanes words of their own nobody reads), and the next node's are
  // asked for
  lds_store(cons_word, i + 1 - a.c0);
  slot_off += kRunSlotDoubles;
  if (slot_off == kRunSlots * kRunSlotDoubles) slot_off = 0;
  if (i + 1 < a.c1) run_request_m(rb + slot_off, lane, nxt);
  double R0 = 0, R1 = 0;
  for (int m = 0; m < nmsg; ++m) {
    const double alpha = readlane_f64(sd, m);
    const double h = gamma * Di - (m == 0 ? mold0 : mold1);   // (lanes beyond K: the loader staged -inf as their old message, so h = +inf)
    double out = 0;   // (alpha == 0: typeStereoLinear.h:390-396, a constant row, normalised)
    if (alpha != 0) {
      *tabl = h;
      if (__builtin_expect(alpha != alpha_have, 0)) {
        // (a real branch: as a select the refresh costs every message 4 G multiplies and 8 G conditional moves)
        asm volatile("" ::: "memory");
#pragma unroll
        for (int d = 0; d < 4 * G; ++d) ad[d] = alpha * ((double)(d + 1) * a.step);
        alpha_have = alpha;
      }
      const double hmin = wave_min_dpp(h);
      const double vtrunc = hmin + alpha * a.lambda;
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      double lo[4 * G], hi[4 * G];
#pragma unroll
      for (int d = 0; d < 4 * G; ++d) { lo[d] = tabl[-(d + 1)]; hi[d] = tabl[d + 1]; }
      double ma = h, mb = vtrunc;   // (two chains)
#pragma unroll
      for (int d = 0; d < 4 * G; ++d) { ma = min_raw(ma, lo[d] + ad[d]); mb = min_raw(mb, hi[d] + ad[d]); }
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      out = min_raw(ma, mb) - hmin;
    }
    if (m == 0) R0 = out; else R1 = out;
  }
  A0 = R0; A1 = nmsg == 2 ? R1 : R0;
  if (cut) {
    // the rows the segment behind this node starts from: to the publisher
    const int ps = cut & 1;
    if (__builtin_amdgcn_readfirstlane(lds_load(rw + kRwFree)) < cut - 2 &&
        !__builtin_amdgcn_readfirstlane((int)run_wait_m(rw + kRwFree, cut - 2, abort_word, a.abort_flag, a.spin_ticks))) return false;
    double *pb = rb + kRunPub + ps * 128;
    pb[lane] = A0; pb[64 + lane] = A1;
    if (lane == 0) lds_store(rw + kRwKinds + ps, __builtin_amdgcn_readlane(sw, kRsPubKinds));
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    if (lane == 0) lds_store(rw + kRwPubM + ps, cut);
  }
  return true;
}

template <bool BACKWARD, int G>
__device__ __forceinline__ void run_messages_g(const RunArgsM &a, double *rb, int lane, int *abort_word) {
  const bool act = lane < a.K;
  int *rw = (int *)(rb + kRunWords);
  double *tabl = rb + kRunTab + 16 + lane;   // this lane's entry of the H table
  int *cons_word = lane == 0 ? rw + kRwConsM : (int *)(rb + kRunDoubles) + lane;   // (kRunDummy words behind the runner's LDS)
  double ad[4 * G];   // alpha |d step| of the window's index distances, for the weight seen last
#pragma unroll
  for (int d = 0; d < 4 * G; ++d) ad[d] = 0;
  double alpha_have = 0;   // (never a weight that takes the window path)
  double A0 = 0, A1 = 0;
  RunNodeM na, nb;
  run_request_m(rb, lane, na);
  nb = na;
  int slot_off = 0;
  // (two visits per trip: the nodes' registers swap roles instead of being copied)
  for (int i = a.c0; i < a.c1; i += 2) {
    if (!run_visit_m<BACKWARD, G>(a, rb, rw, tabl, cons_word, lane, act, i, slot_off, na, nb, A0, A1, ad, alpha_have, abort_word)) return;
    if (i + 1 < a.c1 && !run_visit_m<BACKWARD, G>(a, rb, rw, tabl, cons_word, lane, act, i + 1, slot_off, nb, na, A0, A1, ad, alpha_have, abort_word)) return;
  }
}

template <bool BACKWARD>
__device__ __attribute__((noinline)) void run_messages(const DevParams *pp, int rb_off_, int abort_off_) {
  extern __shared__ __attribute__((aligned(16))) double run_lds[];
  constexpr int D = BACKWARD ? 1 : 0;
  RunArgsM a;
  a.K = uniform_value(pp->K); a.window = uniform_value(pp->window); a.c0 = uniform_value(pp->spec_c0[D]); a.c1 = uniform_value(pp->spec_c1[D]);
  a.lambda = uniform_value(pp->lambda); a.step = uniform_value(pp->uniform_step);
  a.abort_flag = uniform_value(pp->abort_flag); a.spin_ticks = uniform_value(pp->spin_ticks); a.stat = uniform_value(pp->spec_stat);
  double *rb = run_lds + __builtin_amdgcn_readfirstlane(rb_off_);
  int *abort_word = (int *)(run_lds + __builtin_amdgcn_readfirstlane(abort_off_)) + 1;
  const int lane = threadIdx.x & (kWave - 1);
  // the window in groups of four entries (entries beyond it cost >= vTrunc bit for bit, finish_inputs checks the spacing);
  // windows of more than eight entries keep the plain schedule (spec_active, trws_plan.hip)
  if (a.window <= 4) run_messages_g<BACKWARD, 1>(a, rb, lane, abort_word);
  else run_messages_g<BACKWARD, 2>(a, rb, lane, abort_word);
}

// ---- wave 1: the labels of the primal pass (minimize.cpp:223-264, as the primal wave of a visit computes them) ------
struct RunNodeP {
  int sw;
  double sd, TH, O0, O1, O2, O3;
};
__device__ __forceinline__ void run_request_p(const double *sl, int lane, RunNodeP &n) {
  n.sw = ((const int *)(sl + kRunSc))[lane & 15];
  asm volatile("" ::: "memory");
  n.sd = sl[kRunSc + 8 + (lane & 7)];
  n.TH = sl[kRunRowTH + lane];
  n.O0 = sl[kRunRowOUT + lane]; n.O1 = sl[kRunRowOUT + 64 + lane]; n.O2 = sl[kRunRowOUT + 128 + lane]; n.O3 = sl[kRunRowOUT + 192 + lane];
}
template <bool BACKWARD>
__device__ __attribute__((noinline)) void run_labels(const DevParams *pp_, int rb_off_, int abort_off_) {
  extern __shared__ __attribute__((aligned(16))) double run_lds[];
  const DevParams p = uniform_params(pp_);
  double *rb = run_lds + __builtin_amdgcn_readfirstlane(rb_off_);
  int *abort_word = (int *)(run_lds + __builtin_amdgcn_readfirstlane(abort_off_)) + 1;
  const int lane = threadIdx.x & (kWave - 1);
  const int c0 = p.spec_c0[BACKWARD ? 1 : 0], c1 = p.spec_c1[BACKWARD ? 1 : 0];
  const double inf = __builtin_huge_val();
  const int K = p.K;
  const bool act = lane < K;
  int *rw = (int *)(rb + kRunWords);
  const double posk = act ? p.pos[lane] : 0.0;
  const double lambda = p.lambda;
  int xprev = 0;
  RunNodeP cur, nxt;
  run_request_p(rb, lane, cur);
  nxt = cur;
  for (int i = c0; i < c1; ++i) {
    const double *sl = rb + ((i - c0) % kRunSlots) * kRunSlotDoubles;
    if (RLI(cur.sw, kRsTag) != i + 1) {
      if (!run_wait(p, (const int *)(sl + kRunSc) + kRsTag, i + 1, abort_word)) return;
      run_request_p(sl, lane, cur);
    }
    const int sw = cur.sw;
    const double sd = cur.sd;
    const int nout = RLI(sw, kRsNout), nin = RLI(sw, kRsNin), md = RLI(sw, kRsMd), cut = RLI(sw, kRsCut);
    double db = act ? cur.TH : 0.0;
    const double o0 = cur.O0, o1 = cur.O1, o2 = cur.O2, o3 = cur.O3;
    if (lane == 0) lds_store(rw + kRwConsP, i + 1 - c0);
    if (i + 1 < c1) run_request_p(rb + ((i + 1 - c0) % kRunSlots) * kRunSlotDoubles, lane, nxt);
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      if (k < nin) {
        const int src = RLI(sw, kRsSrc + k);
        const int ks = src < 0 ? xprev : src;
        const double pks = readlane_f64(posk, ks);
        const double d = ((md >> k) & 1) == 0 ? pks - posk : posk - pks;
        db += readlane_f64(sd, 3 + k) * min_raw(fabs(d), lambda);
      }
    }
    double di = db;
    if (nout > 0) di += o0;
    if (nout > 1) di += o1;
    if (nout > 2) di += o2;
    if (nout > 3) di += o3;
    const double dim = act ? di : inf;
    const double vbest = wave_min_dpp(dim);
    xprev = __builtin_ctzll(__builtin_amdgcn_ballot_w64(dim == vbest));
    if (cut) {
      const int ps = cut & 1;
      if (!run_wait(p, rw + kRwFree, cut - 2, abort_word)) return;
      if (lane == 0) { rw[kRwLabel + ps] = xprev; rw[kRwNode + ps] = RLI(sw, kRsNode); }
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
      if (lane == 0) lds_store(rw + kRwPubP + ps, cut);
    }
    cur = nxt;
  }
}

// ---- waves 2-9: staging ---------------------------------------------------------------------------------------------
template <bool BACKWARD, bool PRIMAL, bool UPDATE>
__device__ __attribute__((noinline)) void run_loader(const DevParams *pp_, int epoch_, int rb_off_, int abort_off_, int lw_) {
  extern __shared__ __attribute__((aligned(16))) double run_lds[];
  const DevParams p = uniform_params(pp_);
  double *rb = run_lds + __builtin_amdgcn_readfirstlane(rb_off_);
  int *abort_word = (int *)(run_lds + __builtin_amdgcn_readfirstlane(abort_off_)) + 1;
  const int lane = threadIdx.x & (kWave - 1);
  const int epoch = __builtin_amdgcn_readfirstlane(epoch_), lw = __builtin_amdgcn_readfirstlane(lw_);
  constexpr int D = BACKWARD ? 1 : 0;
  const int c0 = p.spec_c0[D], c1 = p.spec_c1[D];
  constexpr int DW = TrwsGraph::kDescWords;
  const int32_t *desc = p.desc[D];
  const int K = p.K;
  const int lk = lane < K ? lane : K - 1;
  int *rw = (int *)(rb + kRunWords);
  const int L = p.spec_len, nseg = p.spec_nseg;
  // (the descriptors of this wave's NEXT node are asked for before the current one is worked on)
  int w2 = 0, wn2 = 0;
  if (c0 + lw < c1) { w2 = desc[(size_t)(c0 + lw) * DW + lane]; wn2 = c0 + lw + 1 < c1 ? desc[(size_t)(c0 + lw + 1) * DW + lane] : 0; }
  for (int i = c0 + lw; i < c1; i += kRunLoaders) {
    const int w = w2, wn = wn2;
    if (i + kRunLoaders < c1) {
      w2 = desc[(size_t)(i + kRunLoaders) * DW + lane];
      wn2 = i + kRunLoaders + 1 < c1 ? desc[(size_t)(i + kRunLoaders + 1) * DW + lane] : 0;
    }
    const int f = RLI(w, 2), fn = RLI(wn, 2);
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
    // the slots of the rows THIS node receives, in the numbering of the node in front (first distinct one: row "A0")
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
    // cut behind this node?
    int cut = 0, pubkinds = 0;
    if (i + 1 < c1 && (i + 1 - c0) % L == 0 && (i + 1 - c0) / L < nseg) {
      cut = (i + 1 - c0) / L;
      for (int k = noutn; k < ntotn; ++k)
        if ((frn >> k) & 1) pubkinds |= (__builtin_amdgcn_readlane(wn, 12 + k) == s0 ? 8 : 9) << (4 * k);
    }
    // the node's own data (nobody writes it before the segment that holds the node walks it)
    const int fm = RLI(w, kDescFetch) & 255;
    const double *ua = p.unary + (size_t)((unsigned long long)(unsigned)RLI(w, 0) * (unsigned long long)(unsigned)K) + lk;
    const double theta = *ua;
    const double av = p.alpha[__shfl(w, 4 + j8, kWave)];
    double r[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      r[k] = 0;
      if (k < nout) r[k] = *(p.msg + (size_t)((unsigned long long)(unsigned)RLI(w, 4 + k) * (unsigned long long)(unsigned)K) + lk);
    }
    // foreign dependencies (everything but the node in front), then their rows and labels
    if (ndep > 0) wait_for_dependencies_w(p, ndep, __shfl(w, 20 + (lane & 3), kWave), RLI(w, 1), epoch, lane, abort_word);
    if (lds_load(abort_word)) return;
    int src = -1;   // lane k < nin: the label the k-th incoming row's pairwise term takes (-1: the node in front)
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      if (k >= nout && k < ntot && ((fm >> k) & 1) && !((fr >> k) & 1)) {
        if (UPDATE) r[k] = ld_sc1(p.msg + (size_t)((unsigned long long)(unsigned)RLI(w, 4 + k) * (unsigned long long)(unsigned)K) + lk);
        if (PRIMAL) { const int xv = ld_sc1(p.x + RLI(w, 32 + k)); if (lane == k - nout) src = xv; }
      }
    }
    // the ring slot: both recurrences have taken the node that had it into their registers
    double *sl = rb + ((i - c0) % kRunSlots) * kRunSlotDoubles;
    if (UPDATE && !run_wait(p, rw + kRwConsM, i - c0 - kRunSlots + 1, abort_word)) return;
    if (PRIMAL && !run_wait(p, rw + kRwConsP, i - c0 - kRunSlots + 1, abort_word)) return;
    if (UPDATE) {
      double P = theta;
#pragma unroll
      for (int k = 0; k < 8; ++k)
        if (k < kfirst) P += r[k];
      sl[kRunRowP + lane] = P;
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        if (stage_of[0] == k) sl[kRunRowS + lane] = r[k];
        if (stage_of[1] == k) sl[kRunRowS + 64 + lane] = r[k];
        if (stage_of[2] == k) sl[kRunRowS + 128 + lane] = r[k];
      }
#pragma unroll
      for (int k = 0; k < 4; ++k) {   // (lanes beyond K: -inf, which makes the recurrence's H = gamma Di - m = +inf there without a select)
        if (s0 == k) sl[kRunRowM + lane] = lane < K ? r[k] : -__builtin_huge_val();
        if (s1 == k) sl[kRunRowM + 64 + lane] = lane < K ? r[k] : -__builtin_huge_val();
      }
    }
    if (PRIMAL) {
      sl[kRunRowTH + lane] = theta;
#pragma unroll
      for (int k = 0; k < 4; ++k)
        if (k < nout) sl[kRunRowOUT + k * 64 + lane] = r[k];
    }
    // the two messages to the next node are ONE message if weights and old rows agree (positions are shared)
    int nmsg = s0 < 0 ? 0 : 1;
    if (UPDATE && s1 >= 0) {
      bool same = RLI(__double2hiint(av), s0) == RLI(__double2hiint(av), s1) && RLI(__double2loint(av), s0) == RLI(__double2loint(av), s1);
#pragma unroll
      for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b)
          if (a != b && s0 == a && s1 == b) same = same && !UNI(lane < K && __double_as_longlong(r[a]) != __double_as_longlong(r[b]));
      nmsg = same ? 1 : 2;
    }
    {
      int word = 0;
      word = lane == kRsNt ? nt : lane == kRsKinds ? (kinds | (nt << 16)) : lane == kRsNmsg ? nmsg : lane == kRsCut ? cut
           : lane == kRsPubKinds ? pubkinds : lane == kRsNout ? nout : lane == kRsNin ? nin : lane == kRsMd ? (md >> nout) : lane == kRsNode ? RLI(w, 0) : 0;
      const int srck = __shfl(src, lane - kRsSrc, kWave);
      if (lane >= kRsSrc && lane < kRsSrc + 4) word = srck;
      if (lane < kRsTag) ((int *)(sl + kRunSc))[lane] = word;
      // doubles 8-15: alpha of the two messages, gamma (MRFEnergy.cpp:207-228), alpha of the incoming rows
      const double a0 = readlane_f64(av, s0 < 0 ? 0 : s0), a1 = readlane_f64(av, s1 < 0 ? (s0 < 0 ? 0 : s0) : s1);
      const double ain = __shfl(av, nout + (lane - 3 < 0 ? 0 : lane - 3), kWave);
      const double g = (double)1 / (double)(nout > nin ? nout : nin > 0 ? nin : 1);
      if (lane < 8) sl[kRunSc + 8 + lane] = lane == 0 ? a0 : lane == 1 ? a1 : lane == 2 ? g : ain;
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    if (lane == 0) lds_store((int *)(sl + kRunSc) + kRsTag, i + 1);
  }
}

// ---- wave 10: what a segment starts from, to global memory ------------------------------------------------------------
template <bool PRIMAL, bool UPDATE>
__device__ __attribute__((noinline)) void run_publisher(const DevParams *pp_, int epoch_, int rb_off_, int abort_off_) {
  extern __shared__ __attribute__((aligned(16))) double run_lds[];
  const DevParams p = uniform_params(pp_);
  double *rb = run_lds + __builtin_amdgcn_readfirstlane(rb_off_);
  int *abort_word = (int *)(run_lds + __builtin_amdgcn_readfirstlane(abort_off_)) + 1;
  const int lane = threadIdx.x & (kWave - 1);
  const int epoch = __builtin_amdgcn_readfirstlane(epoch_);
  const int K = p.K;
  const int lk = lane < K ? lane : K - 1;
  int *rw = (int *)(rb + kRunWords);
  for (int s = 1; s < p.spec_nseg; ++s) {
    const int ps = s & 1;
    if (UPDATE) {
      if (!run_wait(p, rw + kRwPubM + ps, s, abort_word)) return;
      const double *pb = rb + kRunPub + ps * 128;
      double a0 = pb[lk], a1 = pb[64 + lk];
      const int kinds = lds_load(rw + kRwKinds + ps);
      if ((p.debug & 16384) && s % 3 == 1 && lane == 0) a0 = __longlong_as_double(__double_as_longlong(a0) ^ 1ll);   // (development: a wrong row)
      if ((p.debug & 65536) && s % 3 == 1 && (lane & 1)) a0 += 0.375;                                                // (development: a VERY wrong row)
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        const int kd = (kinds >> (4 * k)) & 15;
        if (kd) st_sc1(p.spec_rows + ((size_t)s * 8 + k) * K + lk, kd == 8 ? a0 : a1);
      }
    }
    if (PRIMAL) {
      if (!run_wait(p, rw + kRwPubP + ps, s, abort_word)) return;
      int label = lds_load(rw + kRwLabel + ps);
      const int node = lds_load(rw + kRwNode + ps);
      if ((p.debug & 32768) && s % 5 == 2) label = label > 0 ? label - 1 : (K > 1 ? 1 : 0);                          // (development: a wrong label)
      (void)node;
      if (lane == 0) st_sc1(p.spec_x + s, label);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (lane == 0) { st_sc1(p.done + p.N + s, epoch); lds_store(rw + kRwFree, s); }
  }
}

// (noinline, parameters through a pointer to their copy in global memory, LDS through the dynamic allocation's own
//  symbol: nothing of this routine -- registers, scalars, the parameter block's address -- leaks into the kernel it is
//  called from, whose visit loops sit at the register limits)
template <bool BACKWARD, bool PRIMAL, bool UPDATE>
__device__ __attribute__((noinline)) void chain_runner(const DevParams *pp_, int epoch_, int rb_off_, int abort_off_) {
  extern __shared__ __attribute__((aligned(16))) double run_lds[];
  const DevParams &p = *pp_;
  const int epoch = epoch_, rb_off = rb_off_, abort_off = abort_off_;
  double *rb = run_lds + rb_off;
  int *abort_word = (int *)(run_lds + abort_off) + 1;
  constexpr int D = BACKWARD ? 1 : 0;
  const int tid = threadIdx.x, lane = tid & (kWave - 1);
  const int wave = __builtin_amdgcn_readfirstlane(tid / kWave);
  const int c0 = p.spec_c0[D], c1 = p.spec_c1[D];
  int *rw = (int *)(rb + kRunWords);
  if (tid < 64) rw[tid] = 0;
  if (tid < kRunSlots) ((int *)(rb + tid * kRunSlotDoubles + kRunSc))[kRsTag] = 0;   // (no node staged)
  if (tid < 16) { rb[kRunTab + tid] = __builtin_huge_val(); rb[kRunTab + 80 + tid] = __builtin_huge_val(); }
  if (p.timeline && tid == 0) p.timeline[((size_t)2 * p.tl_stride + D) * 2] = wall_clock64();
  __syncthreads();
  const unsigned long long trole0 = wall_clock64();
  if (wave == 0) { if (UPDATE) { __builtin_amdgcn_s_setprio(3); run_messages<BACKWARD>(pp_, rb_off, abort_off_); __builtin_amdgcn_s_setprio(0); } }
  else if (wave == 1) { if (PRIMAL) { __builtin_amdgcn_s_setprio(3); run_labels<BACKWARD>(pp_, rb_off, abort_off_); __builtin_amdgcn_s_setprio(0); } }
  else if (wave < 2 + kRunLoaders) run_loader<BACKWARD, PRIMAL, UPDATE>(pp_, epoch, rb_off, abort_off_, wave - 2);
  else if (wave == 2 + kRunLoaders) run_publisher<PRIMAL, UPDATE>(pp_, epoch, rb_off, abort_off_);
  // (development: when each role was done, 100 MHz ticks since the roles started -- messages, labels, last loader, publisher)
  if (p.timeline && p.spec_stat && lane == 0 && (wave <= 1 || wave == 1 + kRunLoaders || wave == 2 + kRunLoaders))
    p.spec_stat[8 + 4 * D + (wave <= 1 ? wave : wave - kRunLoaders + 1)] = wall_clock64() - trole0;
  __syncthreads();
  if (p.timeline && tid == 0) p.timeline[((size_t)2 * p.tl_stride + D) * 2 + 1] = wall_clock64();
  if (p.spec_stat && tid == 0) atomicAdd(p.spec_stat + 2, (unsigned long long)(c1 - c0));
}

// ---- commit of a speculative segment (called by all waves of the workgroup behind the barrier that ended the segment's
// last visit: the storer has drained the last node's rows).  The segment in front commits first (its flag); then what
// this segment started from is compared, bit for bit, with what that segment's last node really handed over: equal ->
// every visit here saw the sequential sweep's inputs and the nodes' flags go up; different (the runner's plain min-plus
// row was not the reference's envelope, or a row of the runner's was stale) -> the overwritten rows are put back and the
// caller walks the visits again from the real rows.  Returns 0 committed, 1 walk again (ctl[3] set), 2 gave up.
template <bool BACKWARD, bool PRIMAL, bool UPDATE>
__device__ __attribute__((noinline)) int spec_commit(const DevParams *pp_, int epoch_, int p0_, int p1_, int seg_, int compare_) {
  extern __shared__ __attribute__((aligned(16))) double run_lds[];
  const DevParams &p = *pp_;
  const int epoch = epoch_, p0 = p0_, p1 = p1_, seg = seg_, compare = compare_;
  constexpr int D = BACKWARD ? 1 : 0;
  constexpr int DW = TrwsGraph::kDescWords;
  const int32_t *desc = p.desc[D];
  int *ctl = (int *)(run_lds + kPipeCtlOff);
  const int tid = threadIdx.x, lane = tid & (kWave - 1);
  const int wave = __builtin_amdgcn_readfirstlane(tid / kWave);
  const int K = p.K;
  const int lkc = lane < K ? lane : K - 1;
  if (compare) {
    if (wave == 0) {
      int differ = 0;
      if (!wait_flag(p, p.N + p.spec_nseg + seg, epoch, -2)) { if (lane == 0) ctl[1] = 1; }
      else {
        const int w = desc[(size_t)p0 * DW + lane];
        const int f = RLI(w, 2);
        const int nout = f & 15, ntot = nout + ((f >> 4) & 15);
        for (int j = nout; j < ntot; ++j) {
          if (__builtin_amdgcn_readlane(w, 12 + j) < 0) continue;
          if (UPDATE) {
            const double a = ld_sc1(p.spec_rows + ((size_t)seg * 8 + j) * K + lkc);
            const double b = ld_sc1(p.msg + (size_t)__builtin_amdgcn_readlane(w, 4 + j) * K + lkc);
            differ |= UNI(__double_as_longlong(a) != __double_as_longlong(b)) ? 1 : 0;
          }
          if (PRIMAL) differ |= ld_sc1(p.spec_x + seg) != ld_sc1(p.x + __builtin_amdgcn_readlane(w, 32 + j)) ? 1 : 0;
        }
      }
      if (lane == 0) ctl[2] = differ;
    }
    __syncthreads();
    const int differ = __builtin_amdgcn_readfirstlane(ctl[2]);
    const int gave_up = __builtin_amdgcn_readfirstlane(ctl[1]);
    __syncthreads();
    if (gave_up) return 2;
    if (differ) {
      if (UPDATE) {
        for (int pos = p0 + wave; pos < p1; pos += kPipeWaves) {
          const int w = desc[(size_t)pos * DW + lane];
          const int nout = RLI(w, 2) & 15;
          for (int j = 0; j < nout && j < 4; ++j)
            st_sc1(p.msg + (size_t)__builtin_amdgcn_readlane(w, 4 + j) * K + lkc,
                   ld_sc1(p.spec_undo + ((size_t)(seg * p.spec_max_len + (pos - p0)) * 4 + j) * K + lkc));
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

#undef RLI

}  // namespace
}  // namespace stereo
