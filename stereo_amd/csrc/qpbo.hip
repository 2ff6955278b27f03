// QPBO roof-duality binary fusion on MI355X (gfx950): kernels, host driver, C ABI.
//
// Replaces the reference's rd_mex gateway + QPBO v1.3 library for this path
// (cpp/rd_mex.cpp:14-100; cpp/QPBO-v1.3.src/QPBO.cpp:408-507 AddPairwiseTerm,
// QPBO.h:760-807 ComputeWeights, QPBO.cpp:786-816 + QPBO_extra.cpp:137-239
// MergeParallelEdges, QPBO.cpp:818-845 Solve, QPBO_maxflow.cpp:477-617 maxflow,
// QPBO_postprocessing.cpp:10-120 ComputeWeakPersistencies, QPBO_extra.cpp:1151-1233
// Improve, QPBO.cpp:847-917 energy / lower bound).
//
// Construction: every neighbour pair's directed 2x2 tables are
// summed (the reversed ones transposed), normal-formed once, and laid out as the
// doubled graph: node v < N is x_v, node v + N its mate.  A submodular pair (i,j)
// gives arcs i->j, j->i and the mirror j'->i', i'->j'; a supermodular pair gives
// i->j', j'->i and j->i', i'->j.  Arc a and its reverse are a, a^1.
//
// Max-flow: deterministic synchronous push-relabel.  Every iteration is
//   K1 push      (thread per node, old heights; a pair of arcs is only ever
//                 modified by the one endpoint whose height is larger)
//   K2 gather    (each node adds the flow pushed into it in its own arc order,
//                 then relabels from the post-push residual graph)
// with periodic global relabelling (frontier BFS from the sink).  Nothing depends
// on scheduling or atomics' arrival order, so results are bitwise reproducible.
// Strong persistency only needs T = {v : v reaches the sink in the residual
// graph}: x_i = 1 iff i in T, x_i = 0 iff mate(i) in T (QPBO.cpp:840-844); T is
// the same for every maximum flow/preflow.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cmath>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <mutex>
#include <numeric>
#include <string>
#include <vector>

#include "../../include/stereo_hip.h"
#include "common.h"

namespace stereo {
namespace {

constexpr int kQB = 256;

struct QpboDev {
  int n;             // doubled node count
  int m;             // arcs
  const int32_t *aptr, *head, *rev;  // arcs grouped by tail; rev[a] = reverse arc
  double *r, *delta, *ex, *snk;
  int32_t *h, *h2;
  // Improve launch only (else nullptr): the exact heights at the moment the current node was fixed
  int32_t *keep;
  int32_t *counters;  // [0] active nodes, [1] frontier size (next), [2] changed flag
  // tiling for the block-local relabelling: tile T owns positions [T * 1024, (T + 1) * 1024);
  // perm[pos] = node or -1, pos_of[node] = pos.  A tile holds nodes that are close in the graph
  // (a 16 x 32 pixel patch and its mates when the grid shape is known).
  const int32_t *perm, *pos_of;
  int ntiles;
  // graphs with at most four arcs per node: what a tile needs to know about its arcs, by tile position, so that
  // a tile is loaded in two dependent round trips (this table | the state it points to) instead of four
  // (node -> arc range -> heads -> their positions).  13 words per position, one array per word:
  // [0] first arc | degree << 28, [1 + k] head of arc k, [5 + k] slot of the head in this tile | index of the
  // reverse arc among the head's arcs << 10, or -1 - (the head's tile), [9 + k] reverse arc.  Filled at the
  // start of a launch (the heads depend on the move), kept for the Improve launch of the same move.
  int32_t *tab;
  unsigned long long *hx;   // tiled rounds: hx_pack word per node (nullptr: no tiled rounds)
  // tiled rounds: dirty[parity][tile] == number of the round = the tile holds excess that can still move, or
  // flow was pushed into it across its border in the round before; other tiles are skipped.  (Round numbers
  // instead of flags that are cleared: every workgroup reads ALL marks of a round to find its share of the
  // marked tiles, see `collect`, so nothing may change them while the round runs.)
  int32_t *dirty;
  // the same idea for the label-correcting relabelling: rdirty[parity][tile] = a height next to
  // the tile (or inside it) went down in the last step
  int32_t *rdirty;
};

// ---- the whole max-flow in ONE launch -------------------------------------------------------
// A loop of launches is launch bound (round 1 measured ~800 launches of a few microseconds of
// work each for a Teddy move).  This kernel runs synchronous rounds -- push | barrier |
// gather + relabel | barrier -- and the level-synchronous global relabelling between grid-wide
// barriers: one workgroup of 1024 threads per CU, all resident (cooperative launch), nodes
// in a grid-stride loop.  Everything one workgroup writes and another reads inside the launch
// (heights, residuals, pushed amounts) goes through agent-scope (sc1) accesses -- the XCDs' L2s are
// not coherent for plain accesses inside a launch -- so the barrier itself needs no cache
// write-back / invalidate: __syncthreads drains the stores, one atomic counter does the rest.
// Same arithmetic in the same per-node order as the kernels above, so results are reproducible.
constexpr int kMB = 1024;
// ctl words: [0] barrier counter, [1] abort, [2..4] "changed" slots, [5..7] active-count slots,
// [8] rounds done, [9] global relabels done
struct QpboCtl { enum { kBar = 0, kAbort = 1, kChanged = 2, kActive = 5, kRounds = 8, kRelabels = 9, kNext = 16, kWords = 20 }; };   // [16..18]: result words of the Improve search, used in turn
constexpr int kGridSpinLimit = 1 << 24;
constexpr int kImproveBlocks = 64;    // workgroups of an Improve launch (see QpboSolver::maxflow)
// -DSTEREO_HIP_QPBO_PROFILE: workgroup 0 adds up where its time goes (10 ns ticks and event counts in
// counters[1200 ..], printed with STEREO_HIP_QPBO_VERBOSE): relabelling = init pass | tile set-up | tile
// relaxation | grid barriers; tiled rounds = tile load | local rounds | store | grid barriers
#ifdef STEREO_HIP_QPBO_PROFILE
#define QPROF_T(var) const long long var = wall_clock64()
#define QPROF_ADD(slot, val) do { if (blockIdx.x == 0 && threadIdx.x == 0 && g.counters) g.counters[1200 + (slot)] += (int)(val); } while (0)
#else
#define QPROF_T(var) do {} while (0)
#define QPROF_ADD(slot, val) do {} while (0)
#endif
__device__ __forceinline__ int ldc(const int32_t *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void stc(int32_t *p, int v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ unsigned long long ldc(const unsigned long long *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void stc(unsigned long long *p, unsigned long long v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
// One word per node for the tiled rounds: height | height at the last grid barrier << 24 | round of the store << 48.
// A tile reads its outside neighbours' heights AS OF THE BARRIER (only then does exactly the higher endpoint of an
// arc pair push), but tiles are stored while others are still being loaded -- a workgroup's second tile of a round
// starts when its neighbours' first tiles are done.  The owner stores all three in one word, the reader takes the
// old height if the word was written in the running round: the same result whatever the timing.
__device__ __forceinline__ unsigned long long hx_pack(int cur, int prev, int round) {
  return (unsigned long long)(unsigned)cur | ((unsigned long long)(unsigned)prev << 24) | ((unsigned long long)(round & 0xffff) << 48);
}
__device__ __forceinline__ int hx_at_barrier(unsigned long long w, int round) {
  const int cur = (int)(w & 0xffffffu), prev = (int)((w >> 24) & 0xffffffu);
  return (int)(w >> 48) == (round & 0xffff) ? prev : cur;
}
__device__ __forceinline__ double ldc(const double *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void stc(double *p, double v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

// "Did any thread of the workgroup say yes" on the barrier the caller needs anyway: a wave with a yes stores a
// flag, one s_barrier, everybody reads it.  Three flags in turn -- the next one is cleared while this one is in
// use (its last readers passed the previous call's barrier) -- instead of __syncthreads_or's workgroup reduction,
// which costs several times the relaxation step it guards (0.93 us per step measured, 16 waves).
// Called by all threads; s_any[0..2] zero before the first call.
__device__ __forceinline__ bool wg_any(bool pred, int *s_any, int &slot) {
  if (__builtin_amdgcn_ballot_w64(pred) != 0 && (threadIdx.x & 63) == 0) s_any[slot] = 1;
  const int nxt = slot == 2 ? 0 : slot + 1;
  if (threadIdx.x == 0) s_any[nxt] = 0;
  __syncthreads();
  const bool r = s_any[slot] != 0;
  slot = nxt;
  return r;
}

__device__ __forceinline__ bool grid_sync(int32_t *ctl, unsigned &gen) {
  __syncthreads();
  if (threadIdx.x == 0) {
    ++gen;
    __hip_atomic_fetch_add(ctl + QpboCtl::kBar, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const int target = (int)(gen * gridDim.x);
    int spins = 0;
    while (__hip_atomic_load(ctl + QpboCtl::kBar, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
      __builtin_amdgcn_s_sleep(1);
      if (++spins > kGridSpinLimit) {  // bounded: never hang the device
        __hip_atomic_store(ctl + QpboCtl::kAbort, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        break;
      }
    }
  }
  __syncthreads();
  return __hip_atomic_load(ctl + QpboCtl::kAbort, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0;
}

// improve_perm != nullptr: the launch is the whole loop of QPBO::Improve (QPBO_extra.cpp:1190-1200) --
// find the next node of the permutation that is not strongly labelled, fix it to 0 with the
// reference's INFTY term, maximise the flow again from the current heights, until the permutation
// is exhausted -- instead of one launch and three host round trips per fixed node (round 2: 100-250
// re-solves of ~0.3 ms each on the hard globalstereo moves).  improve_N = number of variables.
__global__ __launch_bounds__(kMB) void qpbo_maxflow_kernel(QpboDev g, int32_t *ctl, int relabel_every, int max_rounds,
                                                            int tiled, int switch_at, int incremental, int first_interval,
                                                            int adaptive, const int32_t *improve_perm, int improve_N) {
  extern __shared__ __attribute__((aligned(16))) double dyn_lds[];  // tile state of the tiled rounds
  __shared__ int s_red;
  __shared__ int s_h[kMB + 1];   // [kMB]: n, the height an arc without residual capacity leads to
  __shared__ int s_h2[kMB];      // tiled rounds: the heights being written while s_h is read, and the other way round
  __shared__ int s_any[3];
  int any_slot = 0;
  if (threadIdx.x < 3) s_any[threadIdx.x] = 0;
  if (threadIdx.x == 0) s_h[kMB] = g.n;
  __syncthreads();
  // The marked tiles of a round / relabelling step, dealt out evenly: every workgroup reads all marks, ranks the
  // marked tiles (ballot + wave counts, the same list everywhere, no atomics) and takes ranks b, b + #workgroups, ...
  // With the fixed tile -> workgroup map a step lasted as long as the workgroup that happened to own two or three
  // marked tiles (some 190 of 658 are marked in a round of the hard globalstereo moves: 73 us per round against
  // ~35 for one tile); results do not depend on who processes a tile.
  constexpr int kMaxMine = 64;
  __shared__ int s_mine[kMaxMine];
  __shared__ int s_wcnt[kMB / 64];
  const bool dealt = g.ntiles <= kMaxMine * (int)gridDim.x;
  auto collect = [&](const int32_t *marks, int tag) -> int {
    if (!dealt) return (g.ntiles - (int)blockIdx.x + (int)gridDim.x - 1) / (int)gridDim.x;   // candidates of the fixed map
    int total = 0;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int base = 0; base < g.ntiles; base += kMB) {
      const int T = base + threadIdx.x;
      const bool f = T < g.ntiles && ldc(marks + T) == tag;
      const unsigned long long b = __builtin_amdgcn_ballot_w64(f);
      if (lane == 0) s_wcnt[wave] = __builtin_popcountll(b);
      __syncthreads();
      int before = total, chunk = 0;
#pragma unroll
      for (int w = 0; w < kMB / 64; ++w) {
        const int c = s_wcnt[w];
        before += w < wave ? c : 0;
        chunk += c;
      }
      if (f) {
        const int rank = before + __builtin_popcountll(b & ((1ull << lane) - 1ull));
        if (rank % (int)gridDim.x == (int)blockIdx.x) s_mine[rank / (int)gridDim.x] = T;
      }
      total += chunk;
      __syncthreads();
    }
    return total > (int)blockIdx.x ? (total - (int)blockIdx.x + (int)gridDim.x - 1) / (int)gridDim.x : 0;
  };
  int RG = 0;   // relabelling steps so far in this launch: the marks of step RG are rdirty[RG & 1][tile] == RG
  unsigned gen = 0;
  // (`adaptive`: bit 0 = the slow-tail test below; bits 8.. = progress, in thousandths of the active nodes per round,
  //  below which a round counts as stagnant)
  const int adaptive_in = adaptive;
  const int stall_permille = (adaptive >> 8) & 0x3ff;
  adaptive &= 1;
  const int n = g.n;
  const int first = blockIdx.x * kMB + threadIdx.x, stride = gridDim.x * kMB;
  int32_t *h = g.h, *h2 = g.h2;
  // rotating flag slots, one ring per flag family: written in phase k of that family, read after its
  // barrier, cleared one phase (of the same family) later
  int slotA = 0, slotC = 0;
  auto ld = [&](int w) { return __hip_atomic_load(ctl + w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); };
  auto clear_next = [&](int base, int slot) {
    if (blockIdx.x == 0 && threadIdx.x == 0)
      __hip_atomic_store(ctl + base + (slot + 1) % 3, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  };
  int G = 0;   // tiled rounds of this launch so far (marks, pushed-amount buffers and hx words are numbered by it)
  bool hx_refresh_all = false;
  const bool tabbed = g.tab != nullptr && tiled > 0;
  const size_t tabn = (size_t)g.ntiles * kMB;
  if (tabbed && improve_perm == nullptr) {
    for (int T = blockIdx.x; T < g.ntiles; T += gridDim.x) {
      const size_t p = (size_t)T * kMB + threadIdx.x;
      const int v = g.perm[p];
      int a0 = 0, deg = 0;
      if (v >= 0) { a0 = g.aptr[v]; deg = g.aptr[v + 1] - a0; }
      int w4[4], rv4[4], pw4[4], aw4[4];
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const int a = k < deg ? a0 + k : 0;
        const int w = g.head[a], rv = g.rev[a];
        w4[k] = k < deg ? w : 0; rv4[k] = k < deg ? rv : 0;
      }
#pragma unroll
      for (int k = 0; k < 4; ++k) { pw4[k] = g.pos_of[w4[k]]; aw4[k] = g.aptr[w4[k]]; }
      g.tab[p] = a0 | (deg << 28);
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        g.tab[(1 + k) * tabn + p] = w4[k];
        g.tab[(5 + k) * tabn + p] = pw4[k] / kMB == T ? ((pw4[k] % kMB) | ((rv4[k] - aw4[k]) << 10)) : -1 - pw4[k] / kMB;
        g.tab[(9 + k) * tabn + p] = rv4[k];
      }
    }
    // (plain stores, read by whichever workgroup is dealt the tile: written back before the barrier, and the
    // relabelling every solve starts with invalidates before anybody reads)
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
    if (!grid_sync(ctl, gen)) return;
  }
  // Exact distances to the sink in the residual graph, then #active nodes.  Label correcting
  // instead of one grid barrier per BFS level: every workgroup relaxes h[v] = min(h[v], h[w] + 1)
  // over the residual arcs of its tile (heights in LDS, arc status in registers) until nothing
  // changes inside the tile, then the tiles exchange their boundary heights at a grid barrier;
  // done when no tile changed.  The fixpoint is the BFS distance whatever the schedule, so the
  // result is deterministic, and a front crosses a whole tile per barrier instead of one level.
  int relabels_done = 0;
  [[maybe_unused]] int improve_steps_dbg = 0;   // (STEREO_HIP_QPBO_CHECK_CONFINED)
  int fixed_node = 0;        // the node the current Improve step fixed
  bool keep_valid = false;   // g.keep holds the exact heights of the flow the current Improve step started from
  auto global_relabel = [&](int &active) -> bool {
    constexpr int kArcRegs = 8;
    const long long t_in = wall_clock64();
    const unsigned gen_in = gen;
    // (`warm`: the heights are the exact distances of a flow that was maximal before a unary term
    // changed -- Improve.  Kept as the starting point of the label correction they stay a valid
    // labelling (every residual arc still sees at most one level down once the relaxation has
    // converged, sink arcs are re-seeded), which is all the push phase needs; far fewer passes.)
    const bool warm = incremental != 0 && relabels_done == 0;
    // (`confined`: a later relabelling of the same Improve step.  Fixing node i hands excess to i alone,
    // and i had no path to the sink (keep_valid is only set then): the excess moves only through nodes that had none either, so no
    // residual arc changes on the old shortest path of a node that had one -- its distance can only have
    // gone down (a shorter way through i's mate).  The snapshot of the exact heights at the moment of
    // the fix is therefore an upper bound of the distances for those nodes, n for all others, and the
    // label correction started from ANY upper bound ends in the same fixpoint, the BFS distances: the
    // same heights as the search from scratch, but the front only crosses the region the fix touched.)
    const bool confined = !warm && keep_valid;
    // (`local`: inside such a step only the tiles the step has touched are looked at.  The step's first, warm
    // relabelling lowers the nodes that can reach the new sink arc at i's mate -- the tiles where that happened are
    // recorded in `touched` --, and from then on distances only grow (pushes along admissible arcs never shorten a
    // path), for nodes of that set alone: every other node keeps the height it has.  So the first pass relaxes the
    // tiles whose input changed instead of all of them; the dirty marks carry the rest as before.)
    const bool local = keep_valid && (warm || confined);
    int32_t *touched = g.keep ? g.keep + n : nullptr;
    int32_t *rd_first = g.rdirty + (size_t)((RG + 1) & 1) * g.ntiles;   // marks of the first step
    ++relabels_done;
    QPROF_T(qp0);
    auto start_height = [&](int v, double snk_v, int cur_h, int keep_h) {
      const int old_h = warm ? cur_h : (confined ? keep_h : n);
      const int new_h = snk_v > 0 ? 1 : (old_h < n ? old_h : n);
      stc(h + v, new_h);
#ifdef STEREO_HIP_QPBO_CHECK_CONFINED
      // development check: outside the touched tiles a confined relabelling starts from the heights that are there
      if (local && confined && g.counters && new_h != cur_h && !ldc(touched + g.pos_of[v] / kMB)) atomicAdd(g.counters + 1107, 1);
#endif
      if (local && warm && new_h < cur_h) {   // a new sink arc: this node's tile and the tiles of its neighbours
        const int T = g.pos_of[v] / kMB;
        stc(touched + T, 1); stc(rd_first + T, RG + 1);
        for (int a = g.aptr[v]; a < g.aptr[v + 1]; ++a) stc(rd_first + g.pos_of[g.head[a]] / kMB, RG + 1);
      }
    };
    constexpr bool kLocalPasses = true;
    if (local && kLocalPasses) {
      // Only what the step can have changed: its first relabelling starts from heights that are exact but for the
      // new sink arc at the fixed node's mate (and the fixed node itself); a later one resets the touched tiles,
      // every other node has the height it started the step with (the check build verifies exactly that).
      if (warm) {
        if (blockIdx.x == 0 && threadIdx.x < 2) {
          const int v = fixed_node + (threadIdx.x ? improve_N : 0);
          start_height(v, ldc(g.snk + v), ldc(h + v), n);
        }
      } else {
        for (int T = blockIdx.x; T < g.ntiles; T += gridDim.x) {
          if (!ldc(touched + T)) continue;
          const int v = g.perm[T * kMB + threadIdx.x];
          if (v >= 0) start_height(v, ldc(g.snk + v), ldc(h + v), ldc(g.keep + v));
        }
      }
    } else {
      // (four nodes per thread requested together: the loads are agent-scope and would otherwise go one by one)
      for (int v0 = first; v0 < n; v0 += 4 * stride) {
        double sk4[4];
        int ch4[4], kh4[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const int v = v0 + u * stride < n ? v0 + u * stride : v0;
          sk4[u] = ldc(g.snk + v);
          ch4[u] = (warm || local) ? ldc(h + v) : n;
          kh4[u] = confined ? ldc(g.keep + v) : n;
        }
#pragma unroll
        for (int u = 0; u < 4; ++u)
          if (v0 + u * stride < n) start_height(v0 + u * stride, sk4[u], ch4[u], kh4[u]);
      }
    }
#ifdef STEREO_HIP_QPBO_CHECK_CONFINED
    // development check: the local pass left every node where the pass over all nodes would have put it
    if (local && kLocalPasses && g.counters) {
      __syncthreads();
      for (int v = first; v < n; v += stride) {
        const int cur_h = ldc(h + v);
        const int old_h = warm ? cur_h : ldc(g.keep + v);
        const int want = ldc(g.snk + v) > 0 ? 1 : (old_h < n ? old_h : n);
        const bool mine_to_write = warm ? (v == fixed_node || v == fixed_node + improve_N) : ldc(touched + g.pos_of[v] / kMB) != 0;
        if (!mine_to_write && want != cur_h && atomicAdd(g.counters + 1108, 1) == 0) {
          g.counters[1110] = v; g.counters[1111] = cur_h; g.counters[1112] = want; g.counters[1113] = warm ? 1 : 0;
        }
      }
    }
#endif
    // A tile is relaxed again only if a height next to it went down in the last step (every tile in
    // the first): once a tile has reached its fixpoint it stays there until an input changes.  The
    // search front crosses the image, the warm search of an Improve step touches a small region.
    if (!local) {
      for (int T = first; T < g.ntiles; T += stride) stc(rd_first + T, RG + 1);
    } else if (confined) {
      for (int T = first; T < g.ntiles; T += stride)
        if (ldc(touched + T)) stc(rd_first + T, RG + 1);
    }
    // (everything another workgroup may have written inside the launch -- heights, residuals, excess, sink
    // capacities -- is read with agent-scope loads; an acquire fence here, i.e. an invalidate of the whole L2 per
    // relabelling, cost more than those loads: STEREO_HIP_QPBO_PROFILE, `init`)
    QPROF_T(qp1);
    QPROF_ADD(0, qp1 - qp0);
    if (!grid_sync(ctl, gen)) return false;
    QPROF_ADD(3, wall_clock64() - qp1);
    for (;;) {
      slotC = (slotC + 1) % 3;
      clear_next(QpboCtl::kChanged, slotC);
      bool any_changed = false;
      ++RG;
      int32_t *rd_in = g.rdirty + (size_t)(RG & 1) * g.ntiles, *rd_out = g.rdirty + (size_t)((RG + 1) & 1) * g.ntiles;
      const int mine = collect(rd_in, RG);
      for (int j = 0; j < mine; ++j) {
        int T;
        if (dealt) T = s_mine[j];
        else {
          T = blockIdx.x + j * gridDim.x;
          if (ldc(rd_in + T) != RG) continue;
        }
        QPROF_T(qp2);
        const int v = g.perm[T * kMB + threadIdx.x];
        int my = n, a0 = 0, a1 = 0;
        int lidx[kArcRegs];  // residual arc to a node of this tile: its slot in s_h; every other arc: the slot that holds n
        int extT[kArcRegs];  // tile of the arc's head if it is another tile (whatever the residual), else -1
        int ext_best = n;    // what the residual arcs that leave the tile allow (their heads do not move during this step)
        if (tabbed) {
          // the arc table of the tile position, then the state it points to: two dependent round trips
          const size_t p = (size_t)T * kMB + threadIdx.x;
          const int ta = g.tab[p];
          int tw[4], tl[4], hwk[4];
          double rk[4];
#pragma unroll
          for (int k = 0; k < 4; ++k) { tw[k] = g.tab[(1 + k) * tabn + p]; tl[k] = g.tab[(5 + k) * tabn + p]; }
          a0 = ta & 0x0fffffff; a1 = a0 + (int)((unsigned)ta >> 28);
          my = v >= 0 ? ldc(h + v) : n;
#pragma unroll
          for (int k = 0; k < 4; ++k) {
            hwk[k] = ldc(h + tw[k]);
            rk[k] = ldc(g.r + (a0 + k < a1 ? a0 + k : 0));
          }
#pragma unroll
          for (int k = 0; k < kArcRegs; ++k) { lidx[k] = kMB; extT[k] = -1; }
#pragma unroll
          for (int k = 0; k < 4; ++k) {
            if (a0 + k < a1) {
              const bool inside = tl[k] >= 0;
              if (!inside) extT[k] = -1 - tl[k];
              if (rk[k] > 0) {
                if (inside) lidx[k] = tl[k] & (kMB - 1);
                else ext_best = hwk[k] + 1 < ext_best ? hwk[k] + 1 : ext_best;
              }
            }
          }
          s_h[threadIdx.x] = my;
        } else {
          if (v >= 0) {
            my = ldc(h + v);
            a0 = g.aptr[v]; a1 = g.aptr[v + 1];
          }
          s_h[threadIdx.x] = my;
          // heads, then everything that hangs on them, requested together: three dependent round trips per tile
          int wk[kArcRegs], pwk[kArcRegs], hwk[kArcRegs];
          double rk[kArcRegs];
          const int nk = tiled > 0 ? 4 : kArcRegs;   // (tiled rounds: at most four arcs per node)
#pragma unroll
          for (int k = 0; k < kArcRegs; ++k) wk[k] = (k < nk && a0 + k < a1) ? g.head[a0 + k] : 0;
#pragma unroll
          for (int k = 0; k < kArcRegs; ++k) {
            pwk[k] = 0; hwk[k] = n; rk[k] = 0.0;
            if (k < nk) {
              pwk[k] = g.pos_of[wk[k]];
              hwk[k] = ldc(h + wk[k]);
              rk[k] = a0 + k < a1 ? ldc(g.r + a0 + k) : 0.0;
            }
          }
#pragma unroll
          for (int k = 0; k < kArcRegs; ++k) {
            lidx[k] = kMB; extT[k] = -1;
            if (a0 + k < a1) {
              const bool inside = pwk[k] / kMB == T;
              if (!inside) extT[k] = pwk[k] / kMB;
              if (rk[k] > 0) {
                if (inside) lidx[k] = pwk[k] % kMB;
                else ext_best = hwk[k] + 1 < ext_best ? hwk[k] + 1 : ext_best;
              }
            }
          }
        }
        __syncthreads();
        QPROF_T(qp3);
        QPROF_ADD(1, qp3 - qp2); QPROF_ADD(4, 1);
        const int start = my;
        bool ch;
        if (tiled > 0) {
          // (tiled rounds are only switched on for graphs with at most four arcs per node: the 4-connected grid)
          do {
            QPROF_ADD(5, 1);
            int best = ext_best < my ? ext_best : my;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
              const int hw = s_h[lidx[k]] + 1;
              best = hw < best ? hw : best;
            }
            ch = best < my;
            if (ch) { my = best; s_h[threadIdx.x] = my; }  // heights only decrease: racing readers are harmless
          } while (wg_any(ch, s_any, any_slot));
        } else {
          do {
            QPROF_ADD(5, 1);
            int best = ext_best < my ? ext_best : my;
#pragma unroll
            for (int k = 0; k < kArcRegs; ++k) {
              const int hw = s_h[lidx[k]] + 1;
              best = hw < best ? hw : best;
            }
            for (int a = a0 + kArcRegs; a < a1; ++a) {  // nodes of higher degree: the rest from memory
              if (!(ldc(g.r + a) > 0)) continue;
              const int w = g.head[a], pw = g.pos_of[w];
              const int hw = pw / kMB == T ? s_h[pw % kMB] : ldc(h + w);
              best = hw + 1 < best ? hw + 1 : best;
            }
            ch = best < my;
            if (ch) { my = best; s_h[threadIdx.x] = my; }
          } while (wg_any(ch, s_any, any_slot));
        }
        if (my < start) {
          stc(h + v, my); any_changed = true;
          if (local) stc(touched + T, 1);
          // whoever has a residual arc INTO v may come down now: the tiles of v's neighbours
#pragma unroll
          for (int k = 0; k < kArcRegs; ++k)
            if (extT[k] >= 0) stc(rd_out + extT[k], RG + 1);
          for (int a = a0 + kArcRegs; a < a1; ++a) {
            const int tw = g.pos_of[g.head[a]] / kMB;
            if (tw != T) stc(rd_out + tw, RG + 1);
          }
        }
        __syncthreads();
        QPROF_ADD(2, wall_clock64() - qp3);
      }
      if (wg_any(any_changed, s_any, any_slot) && threadIdx.x == 0)
        __hip_atomic_store(ctl + QpboCtl::kChanged + slotC, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      QPROF_T(qp4);
      if (!grid_sync(ctl, gen)) return false;
      QPROF_ADD(3, wall_clock64() - qp4); QPROF_ADD(6, 1);
      if (!ld(QpboCtl::kChanged + slotC)) break;
    }
#ifdef STEREO_HIP_QPBO_CHECK_CONFINED
    // development check: exact distances are tight (a node below n without a sink arc has a residual arc one
    // level down); a start that was NOT an upper bound leaves nodes that hang in the air
    if ((confined || local) && g.counters) {
      for (int v = first; v < n; v += stride) {
        const int hv = ldc(h + v);
        if (hv >= n || ldc(g.snk + v) > 0) continue;
        bool sup = false;
        for (int a = g.aptr[v]; a < g.aptr[v + 1]; ++a) sup = sup || (ldc(g.r + a) > 0 && ldc(h + g.head[a]) == hv - 1);
        if (!sup && atomicAdd(g.counters + 1100, 1) == 0) {
          g.counters[1101] = v; g.counters[1102] = hv; g.counters[1103] = ldc(g.keep + v);
          g.counters[1104] = ldc(g.ex + v) > 0; g.counters[1105] = relabels_done; g.counters[1106] = improve_steps_dbg;
        }
      }
    }
#endif
    slotA = (slotA + 1) % 3;
    clear_next(QpboCtl::kActive, slotA);
    int cnt = 0;
    if (local && kLocalPasses) {
      // (a node that can move excess inside such a step was lowered by the step: it lies in a touched tile)
      for (int T = blockIdx.x; T < g.ntiles; T += gridDim.x) {
        if (!ldc(touched + T)) continue;
        const int v = g.perm[T * kMB + threadIdx.x];
        if (v >= 0) {
          const int hh = ldc(h + v);
          cnt += (ldc(g.ex + v) > 0 && hh < n) ? 1 : 0;
          if (g.hx) stc(g.hx + v, hx_pack(hh, hh, 0));   // (the heights the next tiled round reads)
        }
      }
      if (hx_refresh_all && g.hx)
        for (int v = first; v < n; v += stride) { const int hh = ldc(h + v); stc(g.hx + v, hx_pack(hh, hh, 0)); }
    } else {
      for (int v0 = first; v0 < n; v0 += 4 * stride) {
        double e4[4];
        int h4[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const int v = v0 + u * stride < n ? v0 + u * stride : v0;
          e4[u] = ldc(g.ex + v); h4[u] = ldc(h + v);
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) cnt += (v0 + u * stride < n && e4[u] > 0 && h4[u] < n) ? 1 : 0;
        if (g.hx) {   // (the heights the next tiled round reads)
#pragma unroll
          for (int u = 0; u < 4; ++u)
            if (v0 + u * stride < n) stc(g.hx + v0 + u * stride, hx_pack(h4[u], h4[u], 0));
        }
#ifdef STEREO_HIP_QPBO_CHECK_CONFINED
#pragma unroll
        for (int u = 0; u < 4; ++u)   // development check: inside a local step excess only moves in touched tiles
          if (local && g.counters && v0 + u * stride < n && e4[u] > 0 && h4[u] < n && !ldc(touched + g.pos_of[v0 + u * stride] / kMB))
            atomicAdd(g.counters + 1107, 1);
#endif
      }
    }
#ifdef STEREO_HIP_QPBO_CHECK_CONFINED
    if (local && kLocalPasses && g.counters)   // development check: nothing that can move excess outside the touched tiles
      for (int v = first; v < n; v += stride)
        if (ldc(g.ex + v) > 0 && ldc(h + v) < n && !ldc(touched + g.pos_of[v] / kMB)) atomicAdd(g.counters + 1109, 1);
#endif
    hx_refresh_all = false;
    cnt = wg_any(cnt > 0, s_any, any_slot) ? cnt : 0;
    if (threadIdx.x == 0) s_red = 0;
    __syncthreads();
    if (cnt) atomicAdd(&s_red, cnt);
    __syncthreads();
    if (threadIdx.x == 0 && s_red)
      __hip_atomic_fetch_add(ctl + QpboCtl::kActive + slotA, s_red, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (!grid_sync(ctl, gen)) return false;
    active = ld(QpboCtl::kActive + slotA);
    if (blockIdx.x == 0 && threadIdx.x == 0) {  // statistics: #relabels, their barriers and time (10 ns ticks)
      ctl[QpboCtl::kRelabels] += 1;
      ctl[11] += (int)(wall_clock64() - t_in);
      ctl[12] += (int)(gen - gen_in);
    }
    return true;
  };

  int improve_from = 0, improve_steps = 0;
  const bool no_skip_relabel = (adaptive_in >> 30) & 1;   // (development switch STEREO_HIP_QPBO_FINAL_RELABEL=1: always run the final relabelling)
  bool heights_copied = false;   // the solve that just ended wrote g.h after its last grid barrier
  bool solve = improve_perm == nullptr;   // the Improve launch starts from a maximal flow: first find a node to fix
  for (;;) {
  if (solve) {
  int active = 0;
  relabels_done = 0;
  if (!global_relabel(active)) return;
  int rounds = 0, since_relabel = 0, interval = relabel_every < first_interval ? relabel_every : first_interval, stagnant = 0, last_active = active;

  // When to leave the plain rounds for the tiled ones: at round `switch_at` at the latest, and earlier if
  // the excess left over by an exact relabelling drains slowly -- two rounds after the relabelling more
  // than 70 % of it is still there (real NCC fusion moves: 1748 -> 1463, a long tail the tiled rounds
  // cut short; noisy synthetic terms: 322 -> 65, done within a dozen plain rounds).
  int after_relabel = -1;
  bool slow_tail = false;
  while (active > 0 && rounds < max_rounds && (tiled <= 0 || (rounds < switch_at && !slow_tail))) {
    // ---- push (old heights; a pair of arcs is only modified by the endpoint that is higher)
    for (int v = first; v < n; v += stride) {
      double e = ldc(g.ex + v);
      const int hv = ldc(h + v);
      if (!(e > 0) || hv >= n) continue;
      if (hv == 1) {
        const double sk = ldc(g.snk + v);
        if (sk > 0) {
          const double d = e < sk ? e : sk;
          stc(g.snk + v, sk - d);   // (written through: the local passes of an Improve step read it from another workgroup)
          e -= d;
        }
      }
      const int a0 = g.aptr[v], a1 = g.aptr[v + 1];
      // the first eight arcs: residuals, heads, reverse arcs and head heights are requested together
      // (three dependent round trips instead of three per arc); decisions stay in arc order
      constexpr int kB = 8;
      double rb[kB];
      int hb[kB], wb[kB], bb[kB];
#pragma unroll
      for (int k = 0; k < kB; ++k) {
        const int a = a0 + k < a1 ? a0 + k : a0;
        rb[k] = a0 + k < a1 ? ldc(g.r + a) : 0.0;
        wb[k] = g.head[a]; bb[k] = g.rev[a];
      }
#pragma unroll
      for (int k = 0; k < kB; ++k) hb[k] = (a0 + k < a1 && rb[k] > 0) ? ldc(h + wb[k]) : n;
#pragma unroll
      for (int k = 0; k < kB; ++k) {
        if (a0 + k < a1 && e > 0 && rb[k] > 0 && hv == hb[k] + 1) {
          const double d = e < rb[k] ? e : rb[k];
          stc(g.r + a0 + k, rb[k] - d);
          stc(g.r + bb[k], ldc(g.r + bb[k]) + d);
          stc(g.delta + a0 + k, d);
          e -= d;
        }
      }
      for (int a = a0 + kB; a < a1 && e > 0; ++a) {
        const double ra = ldc(g.r + a);
        if (ra > 0 && hv == ldc(h + g.head[a]) + 1) {
          const double d = e < ra ? e : ra;
          const int b = g.rev[a];
          stc(g.r + a, ra - d);
          stc(g.r + b, ldc(g.r + b) + d);
          stc(g.delta + a, d);
          e -= d;
        }
      }
      stc(g.ex + v, e);
    }
    if (!grid_sync(ctl, gen)) return;
    // ---- gather the pushed flow in the node's own arc order, relabel from the post-push residual graph
    slotA = (slotA + 1) % 3;
    clear_next(QpboCtl::kActive, slotA);
    int cnt = 0;
    for (int v = first; v < n; v += stride) {
      double e = ldc(g.ex + v);
      const int a0 = g.aptr[v], a1 = g.aptr[v + 1];
      constexpr int kB = 8;
      int bb[kB];
      double db[kB];
#pragma unroll
      for (int k = 0; k < kB; ++k) bb[k] = g.rev[a0 + k < a1 ? a0 + k : a0];
#pragma unroll
      for (int k = 0; k < kB; ++k) db[k] = a0 + k < a1 ? ldc(g.delta + bb[k]) : 0.0;
#pragma unroll
      for (int k = 0; k < kB; ++k)
        if (db[k] != 0) { e += db[k]; stc(g.delta + bb[k], 0.0); }
      for (int a = a0 + kB; a < a1; ++a) {
        const int b = g.rev[a];
        const double d = ldc(g.delta + b);
        if (d != 0) { e += d; stc(g.delta + b, 0.0); }
      }
      stc(g.ex + v, e);
      int hv = ldc(h + v);
      if (e > 0 && hv < n) {
        int hmin = n;
        if (ldc(g.snk + v) > 0) hmin = 0;
        double rb[kB];
        int hw8[kB];
#pragma unroll
        for (int k = 0; k < kB; ++k) rb[k] = a0 + k < a1 ? ldc(g.r + a0 + k) : 0.0;
#pragma unroll
        for (int k = 0; k < kB; ++k) hw8[k] = rb[k] > 0 ? ldc(h + g.head[a0 + k]) : n;
#pragma unroll
        for (int k = 0; k < kB; ++k) hmin = hw8[k] < hmin ? hw8[k] : hmin;
        for (int a = a0 + kB; a < a1; ++a)
          if (ldc(g.r + a) > 0) {
            const int hw = ldc(h + g.head[a]);
            hmin = hw < hmin ? hw : hmin;
          }
        if (hmin + 1 > hv) hv = hmin + 1 < n ? hmin + 1 : n;
        if (hv < n) ++cnt;
      }
      stc(h2 + v, hv);
    }
    if (threadIdx.x == 0) s_red = 0;
    __syncthreads();
    if (cnt) atomicAdd(&s_red, cnt);
    __syncthreads();
    if (threadIdx.x == 0 && s_red)
      __hip_atomic_fetch_add(ctl + QpboCtl::kActive + slotA, s_red, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (!grid_sync(ctl, gen)) return;
    active = ld(QpboCtl::kActive + slotA);
    if (g.counters && blockIdx.x == 0 && threadIdx.x == 0 && rounds < 1024) g.counters[16 + rounds] = active;
    { int32_t *t = h; h = h2; h2 = t; }
    ++rounds; ++since_relabel;
    // Excess that cannot reach the sink any more only climbs one level per round; an exact
    // relabelling retires it at once.  Early relabels are cheap (few BFS levels) and usually
    // end the solve after a handful of rounds, so the interval starts small and doubles.
    stagnant = (long long)active * 1000 >= (long long)last_active * (1000 - stall_permille) ? stagnant + 1 : 0;  // no progress: what is left is probably cut off
    last_active = active;
    if (adaptive && after_relabel > 0 && since_relabel == 2 && (long long)active * 10 > (long long)after_relabel * 7) slow_tail = true;
    if (active > 0 && (since_relabel >= interval || (stagnant >= 2 && since_relabel >= 4))) {
      if (!global_relabel(active)) return;
      since_relabel = 0; stagnant = 0; last_active = active;
      after_relabel = active;
      interval = interval * 2 < relabel_every ? interval * 2 : relabel_every;
    }
  }
  // ---- tiled rounds (every node has at most four arcs: the 4-connected grid) ------------------
  // One round per grid barrier moves excess one hop; on instances with strong smoothness the flow
  // has to travel hundreds of pixels.  Here a workgroup takes its tile (a 16 x 32 pixel patch and
  // its mates: excess, heights, sink capacities and the residuals of the tile's own arcs) into
  // LDS and runs up to `tiled` synchronous push / gather + relabel rounds there between two grid
  // barriers.  Arcs that leave the tile are pushed on in the first local round only, when both
  // endpoints see the heights of the last barrier (so still only the higher endpoint of an arc pair
  // pushes); the amount goes into delta[parity][arc] and the owner of the reverse arc takes it in
  // at the start of its next round.  Same per-node arithmetic order in every run: deterministic.
  bool exact = false;
  if (tiled > 0 && active > 0 && rounds < max_rounds) {
    const long long t_tiled = wall_clock64();
    const int rounds_in = rounds;
    // hand-over from the plain rounds: heights into g.h, excess and sink capacities written through
    // (from here on the workgroup that owns a node's tile reads and writes them)
    for (int v = first; v < n; v += stride) {
      const int hh = ldc(h + v);
      if (h != g.h) stc(g.h + v, hh);
      if (g.hx) stc(g.hx + v, hx_pack(hh, hh, 0));
      stc(g.ex + v, ldc(g.ex + v)); stc(g.snk + v, ldc(g.snk + v));
    }
    h = g.h; h2 = g.h2;
    // (G, the round number: pushes of round G land in delta buffer G & 1)
    // Few tiles hold excess after the plain rounds (7k active nodes in some dozens of 660 tiles on
    // the long globalstereo moves): a tile without excess that can move and without flow arriving
    // over its border does not change in a round, so it is skipped -- exactly, not heuristically.
    // After an exact relabelling every tile is looked at once (nodes may have become active again).
    auto mark_all_dirty = [&]() {
      // (hx words carry 16 bits of the round number: long before they wrap, start again from 0 and have the
      // relabelling that follows every call rewrite all words as "not stored in any round"; nothing is in transit here)
      // The dirty marks are raw round numbers too: at the wrap both planes are cleared (by the thread that
      // then writes the tile's new mark, so the two stores are ordered), or marks left from the launch's
      // first rounds 2, 3, ... would compare equal to the restarted round numbers.
      // STEREO_HIP_QPBO_WRAP_AT=<n> (1 .. 4095, development): wrap that early, so that a test reaches the wrap at all
      // (tests/test_rd_gpu.py::test_round_numbers_wrap).
      const int wrap_at = ((adaptive_in >> 18) & 0xfff) ? ((adaptive_in >> 18) & 0xfff) : 60000;
      const bool wrap = G > wrap_at;
      if (wrap) { G = 0; hx_refresh_all = true; }
      for (int T = first; T < g.ntiles; T += stride) {
        if (wrap) { stc(g.dirty + T, 0); stc(g.dirty + (size_t)g.ntiles + T, 0); }
        stc(g.dirty + (size_t)((G + 1) & 1) * g.ntiles + T, G + 1);
      }
    };
    mark_all_dirty();
    if (!grid_sync(ctl, gen)) return;
    double *s_d = dyn_lds;   // [thread][arc]: what a neighbour inside the tile pushed over the reverse of that arc
    // L local rounds (L == 0: only take in what was pushed across tile borders); counts the active
    // nodes and one more per workgroup that pushed across a border (that flow is still in transit)
    auto tile_round = [&](int L) -> bool {
      ++G;
      double *dout = g.delta + (size_t)(G & 1) * g.m, *din = g.delta + (size_t)((G + 1) & 1) * g.m;
      int32_t *dirty_in = g.dirty + (size_t)(G & 1) * g.ntiles, *dirty_out = g.dirty + (size_t)((G + 1) & 1) * g.ntiles;
      slotA = (slotA + 1) % 3;
      clear_next(QpboCtl::kActive, slotA);
      int cnt = 0;
      bool crossed = false;
      const int mine = collect(dirty_in, G);
      for (int j = 0; j < mine; ++j) {
        int T;
        if (dealt) T = s_mine[j];
        else {
          T = blockIdx.x + j * gridDim.x;
          if (ldc(dirty_in + T) != G) continue;
        }
        QPROF_T(qt0);
        const int v = g.perm[T * kMB + threadIdx.x];
        const bool valid = v >= 0;
        int loc[4] = {-1, -1, -1, -1}, rvk[4] = {0, 0, 0, 0}, exth[4] = {n, n, n, n}, extT[4] = {0, 0, 0, 0}, a0 = 0, deg = 0;
        double e = 0;
        int hv = n;
        double sk = 0, rk[4] = {0, 0, 0, 0}, e0_in = 0;
        if (tabbed) {
          // the arc table of the tile position, then the state it points to: two dependent round trips
          const size_t p = (size_t)T * kMB + threadIdx.x;
          const int ta = g.tab[p];
          int tw[4], tl[4], rvs[4], hwk[4];
          double din_k[4];
#pragma unroll
          for (int k = 0; k < 4; ++k) {
            tw[k] = g.tab[(1 + k) * tabn + p]; tl[k] = g.tab[(5 + k) * tabn + p]; rvs[k] = g.tab[(9 + k) * tabn + p];
          }
          a0 = ta & 0x0fffffff; deg = (int)((unsigned)ta >> 28);
          const int vv = valid ? v : 0;
          e = ldc(g.ex + vv); hv = ldc(h + vv); sk = ldc(g.snk + vv);
#pragma unroll
          for (int k = 0; k < 4; ++k) {
            rk[k] = ldc(g.r + (k < deg ? a0 + k : 0));
            hwk[k] = hx_at_barrier(ldc(g.hx + tw[k]), G); din_k[k] = ldc(din + rvs[k]);
          }
          if (!valid) { e = 0; hv = n; sk = 0; }
          e0_in = e;
#pragma unroll
          for (int k = 0; k < 4; ++k) {
            if (k < deg) {
              if (tl[k] >= 0) { loc[k] = tl[k] & (kMB - 1); rvk[k] = tl[k] >> 10; din_k[k] = 0; }
              else { exth[k] = hwk[k]; extT[k] = -1 - tl[k]; }
            } else {
              rk[k] = 0; din_k[k] = 0;
            }
          }
#pragma unroll
          for (int k = 0; k < 4; ++k) {
            if (k < deg && din_k[k] != 0) {  // flow that arrived over this arc's reverse during the last round
              e += din_k[k]; rk[k] += din_k[k];
              stc(din + rvs[k], 0.0);
            }
          }
        } else if (valid) {
          // four dependent round trips per tile: node | its terminals and arc range | heads, reverse arcs,
          // residuals | what hangs on the heads -- each level requested as a whole before any of it is used
          e = ldc(g.ex + v); hv = ldc(h + v);
          a0 = g.aptr[v]; deg = g.aptr[v + 1] - a0;
          sk = ldc(g.snk + v);
          e0_in = e;
          double din_k[4];
          int rvs[4], wk[4], pwk[4], awk[4], hwk[4];
#pragma unroll
          for (int k = 0; k < 4; ++k) {
            const int a = k < deg ? a0 + k : 0;
            const int w = g.head[a], rv = g.rev[a];
            wk[k] = k < deg ? w : 0; rvs[k] = k < deg ? rv : 0;
            rk[k] = ldc(g.r + a);
          }
#pragma unroll
          for (int k = 0; k < 4; ++k) {
            pwk[k] = g.pos_of[wk[k]]; awk[k] = g.aptr[wk[k]];
            hwk[k] = hx_at_barrier(ldc(g.hx + wk[k]), G); din_k[k] = ldc(din + rvs[k]);
          }
#pragma unroll
          for (int k = 0; k < 4; ++k) {
            if (k < deg) {
              rvk[k] = rvs[k] - awk[k];
              if (pwk[k] / kMB == T) { loc[k] = pwk[k] % kMB; din_k[k] = 0; }
              else { exth[k] = hwk[k]; extT[k] = pwk[k] / kMB; }
            } else {
              rk[k] = 0; din_k[k] = 0;
            }
          }
#pragma unroll
          for (int k = 0; k < 4; ++k) {
            if (k < deg && din_k[k] != 0) {  // flow that arrived over this arc's reverse during the last round
              e += din_k[k]; rk[k] += din_k[k];
              stc(din + rvs[k], 0.0);
            }
          }
        }
        // A node's excess, sink capacity and residuals are only ever changed by its own thread: they stay in
        // registers for the local rounds.  LDS holds what neighbours read (heights, two buffers in turn: a round
        // writes the new ones where nobody is reading) and what they write (the amounts pushed over a local arc).
#pragma unroll
        for (int k = 0; k < 4; ++k) s_d[threadIdx.x * 4 + k] = 0;
        int *hcur = s_h, *hnext = s_h2;
        hcur[threadIdx.x] = hv;
        const int h0 = hv;  // height at the barrier
        // what this thread has to write back: bit 0 excess, bit 1 sink capacity, bits 2 .. 5 the residuals
        // (most nodes of a tile do not move in a round; a store that is written through costs what a load does)
        int wrote = (e0_in != e) ? 0x3d : 0;   // flow taken in over the border: excess and residuals changed
        __syncthreads();
        QPROF_T(qt1);
        QPROF_ADD(8, qt1 - qt0); QPROF_ADD(12, 1);
        for (int l = 0; l < L; ++l) {
          QPROF_ADD(13, 1);
          // the neighbours' heights of this round, requested together (the buffer being read does not change
          // before the round's last barrier: the push and the relabel below see the same values)
          int hw4[4];
#pragma unroll
          for (int k = 0; k < 4; ++k) hw4[k] = hcur[loc[k] >= 0 ? loc[k] : threadIdx.x];
#pragma unroll
          for (int k = 0; k < 4; ++k) hw4[k] = loc[k] >= 0 ? hw4[k] : exth[k];
          // push (a pair of arcs is only modified by the endpoint that is higher)
          if (valid && e > 0 && hv < n) {
            if (hv == 1 && sk > 0) {
              const double d = e < sk ? e : sk;
              sk -= d;
              e -= d;
              wrote |= 3;
            }
#pragma unroll
            for (int k = 0; k < 4; ++k) {
              if (k < deg && e > 0 && rk[k] > 0) {
                const bool local = loc[k] >= 0;
                const int hw = hw4[k];
                if ((local || l == 0) && hv == hw + 1) {
                  const double d = e < rk[k] ? e : rk[k];
                  rk[k] -= d;
                  wrote |= 1 | (4 << k);
                  if (local) s_d[loc[k] * 4 + rvk[k]] = d;
                  else { stc(dout + a0 + k, d); stc(dirty_out + extT[k], G + 1); crossed = true; }
                  e -= d;
                }
              }
            }
          }
          __syncthreads();
          // gather in the node's own arc order, relabel from the post-push residual graph
          int newh = hv;
          if (valid) {
#pragma unroll
            for (int k = 0; k < 4; ++k) {
              if (k < deg) {
                const double d = s_d[threadIdx.x * 4 + k];
                if (d != 0) { e += d; rk[k] += d; s_d[threadIdx.x * 4 + k] = 0; wrote |= 1 | (4 << k); }
              }
            }
            if (e > 0 && hv < n) {
              int hmin = sk > 0 ? 0 : n;
#pragma unroll
              for (int k = 0; k < 4; ++k) {
                // outside heights are lower bounds.  An outside neighbour that stood exactly one above
                // this node at the barrier may have pushed to it in this round; the residual that gives
                // this node's arc arrives with the next round, so the arc counts as open until then.
                if (k < deg && (rk[k] > 0 || (loc[k] < 0 && exth[k] == h0 + 1)))
                  hmin = hw4[k] < hmin ? hw4[k] : hmin;
              }
              if (hmin + 1 > hv) newh = hmin + 1 < n ? hmin + 1 : n;
            }
          }
          hnext[threadIdx.x] = newh;   // (the other buffer: the old heights are still being read)
          hv = newh;
          { int *t = hcur; hcur = hnext; hnext = t; }
          if (!wg_any(valid && e > 0 && hv < n, s_any, any_slot)) break;
        }
        QPROF_T(qt2);
        QPROF_ADD(9, qt2 - qt1);
        if (valid) {
          if (wrote & 1) stc(g.ex + v, e);   // (written through: read by other workgroups)
          if (wrote & 2) stc(g.snk + v, sk);
          if (hv != h0) {   // (an unchanged height leaves a word whose current-height field is right whatever its round)
            stc(h + v, hv);
            stc(g.hx + v, hx_pack(hv, h0, G));
          }
#pragma unroll
          for (int k = 0; k < 4; ++k)
            if (k < deg && (wrote & (4 << k))) stc(g.r + a0 + k, rk[k]);
          if (e > 0 && hv < n) ++cnt;
        }
        // (also the barrier before the tile buffers are reused)
        const int left = wg_any(valid && e > 0 && hv < n, s_any, any_slot);
        if (left && threadIdx.x == 0) stc(dirty_out + T, G + 1);
        QPROF_ADD(10, wall_clock64() - qt2);
      }
      QPROF_T(qt3);
      if (threadIdx.x == 0) s_red = 0;
      __syncthreads();
      if (cnt) atomicAdd(&s_red, cnt);
      if (crossed) atomicOr(&s_red, 1 << 30);
      __syncthreads();
      if (threadIdx.x == 0 && s_red)
        __hip_atomic_fetch_add(ctl + QpboCtl::kActive + slotA, (s_red & ((1 << 30) - 1)) + (s_red >> 30), __ATOMIC_RELAXED,
                               __HIP_MEMORY_SCOPE_AGENT);
      if (!grid_sync(ctl, gen)) return false;
      QPROF_ADD(11, wall_clock64() - qt3); QPROF_ADD(14, 1);
      active = ld(QpboCtl::kActive + slotA);
      return true;
    };
    do {
      while (active > 0 && rounds < max_rounds) {
        if (!tile_round(tiled)) return;
        if (blockIdx.x == 0 && threadIdx.x == 0 && g.counters && rounds < 1024) g.counters[16 + rounds] = active;
        ++rounds; ++since_relabel;
        stagnant = (long long)active * 1000 >= (long long)last_active * (1000 - stall_permille) ? stagnant + 1 : 0;
        last_active = active;
        if (active > 0 && (since_relabel >= interval || (stagnant >= 2 && since_relabel >= 4))) {
          if (!tile_round(0)) return;  // nothing in transit while the residual graph is searched
          mark_all_dirty();
          if (!global_relabel(active)) return;
          since_relabel = 0; stagnant = 0; last_active = active;
          interval = interval * 2 < relabel_every ? interval * 2 : relabel_every;
        }
      }
      mark_all_dirty();
      if (!global_relabel(active)) return;  // exact heights: anything left over is cut off
    } while (active > 0 && rounds < max_rounds);
    exact = true;
    if (blockIdx.x == 0 && threadIdx.x == 0) { ctl[13] += (int)(wall_clock64() - t_tiled); ctl[14] += rounds - rounds_in; }
  }

  // heights may be stale lower bounds: one exact BFS defines T; leave it in g.h
  // (unless nothing has been pushed since the last relabelling and that one was exact -- from scratch, confined, or
  //  the warm one of an Improve step whose fixed node was cut off from the sink, where the old heights are upper
  //  bounds: most Improve steps find nothing to push after their warm relabelling, and this second one was three of
  //  their seven grid barriers)
  if (!exact && since_relabel == 0 && (relabels_done > 1 || incremental == 0 || keep_valid) && !no_skip_relabel) exact = true;
  if (!exact && !global_relabel(active)) return;
  heights_copied = h != g.h;
  if (h != g.h) {
    for (int v = first; v < n; v += stride) stc(g.h + v, ldc(h + v));
  }
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    ctl[QpboCtl::kRounds] += rounds;
    if (rounds >= max_rounds && active > 0) ctl[QpboCtl::kAbort] = 2;
  }
  h = g.h; h2 = g.h2;
  }  // solve
  if (!improve_perm) break;
  // ---- Improve: the next node of the permutation whose two sides are both / neither connected to the
  // sink (not strongly labelled).  Heights in g.h are exact and written through.
  //
  // A step is TRIVIAL if its node i is cut off from the sink (height n, and so is its mate's) and no residual arc
  // leads into the mate: fixing i hands i excess that cannot move and gives the mate a sink arc that lowers nobody
  // but the mate itself (height 1) -- no node becomes active, the flow stays maximal; the step's warm relabelling
  // would find exactly that in three grid barriers.  Such a step changes the terminal capacities of i and its mate
  // and the mate's height, nothing else: it does not change which other entries of the permutation are ambiguous,
  // nor whether their steps are trivial (heights of other variables, residuals), nor the INFTY of another node
  // (its own terminals and arcs).  So all trivial steps in front of the first non-trivial ambiguous entry commute
  // with each other and are applied TOGETHER, each by the thread that owns the entry, behind the one barrier of the
  // search -- which now looks for the first NON-TRIVIAL ambiguous entry.  (99 % of the Improve steps of the Teddy
  // pair's example_global moves are trivial: 2045 of 2071 in the longest one, 114 ms of grid barriers before.)
  {
    // (excess and sink capacities are written through wherever they are stored: nothing to write back
    // before another workgroup's thread rewrites a node's terminal capacities below)
    const int N = improve_N;
    // (three result words in turn: the one of the NEXT step is cleared before this step's barrier -- its last readers
    //  passed the barrier of the step before)
    int32_t *word = ctl + QpboCtl::kNext + improve_steps % 3, *other = ctl + QpboCtl::kNext + (improve_steps + 1) % 3;
    if (improve_steps == 0 && blockIdx.x == 0 && threadIdx.x == 0) { stc(word, N); stc(other, N); }   // both start at "none"
    // (every workgroup's final heights and terminal capacities are in memory: a solve ends with the grid barriers of
    // its last relabelling, so the barrier is only needed if heights were copied after them -- and in the first step,
    // between the initialisation of the result words and their use)
    if (improve_steps == 0 || heights_copied) {
      if (!grid_sync(ctl, gen)) return;
    }
    heights_copied = false;
    if (improve_steps > 0 && blockIdx.x == 0 && threadIdx.x == 0) stc(other, N);
    if (g.keep) {   // the starting point of this step's later relabellings (global_relabel, `confined`)
      for (int v = first; v < n; v += stride) stc(g.keep + v, ldc(g.h + v));
      for (int T = first; T < g.ntiles; T += stride) stc(g.keep + n + T, 0);   // `touched`, global_relabel
    }
    // AddUnaryTerm(i, 0, INFTY), INFTY = max(-t_i + sum of outgoing residuals, t_i + sum of incoming) + 1
    // evaluated on node i (QPBO_extra.cpp:241-254, :1177-1187), same summation order as the reference
    auto fix_to_zero = [&](int i, bool trivial_step) {
      const int im = i + N;
      const double exi = ldc(g.ex + i), ski = ldc(g.snk + i), exm = ldc(g.ex + im), skm = ldc(g.snk + im);
      const double tcap = exi - ski;
      double c1 = -tcap, c2 = tcap;
      for (int a = g.aptr[i]; a < g.aptr[i + 1]; ++a) { c1 += ldc(g.r + a); c2 += ldc(g.r + g.rev[a]); }
      const double INFTY = (c1 > c2 ? c1 : c2) + 1;
      const double t0 = exi - ski + INFTY, t1 = exm - skm - INFTY;
      stc(g.ex + i, t0 > 0 ? t0 : 0.0); stc(g.snk + i, t0 < 0 ? -t0 : 0.0);
      stc(g.ex + im, t1 > 0 ? t1 : 0.0); stc(g.snk + im, t1 < 0 ? -t1 : 0.0);
      if (trivial_step) {   // what the step's relabelling would leave behind (t1 < 0: the mate has a sink arc now)
        stc(g.h + i, n); stc(g.h + im, 1);
        if (g.hx) { stc(g.hx + i, hx_pack(n, n, 0)); stc(g.hx + im, hx_pack(1, 1, 0)); }
        if (g.keep) { stc(g.keep + i, n); stc(g.keep + im, 1); }
      }
    };
    auto is_trivial = [&](int i) -> bool {   // (i ambiguous) cut off from the sink, no residual arc into the mate
      if (no_skip_relabel || !g.keep || ldc(g.h + i) < n) return false;
      const int im = i + N;
      bool t = true;
      for (int a = g.aptr[im]; a < g.aptr[im + 1]; ++a) t = t && !(ldc(g.r + g.rev[a]) > 0);
      if (!t) return false;
      // ... and the mate really ends up with a sink arc (t1 < 0 in fix_to_zero, which a trivial step's h[mate] = 1 relies
      // on): the doubled graph's symmetry says so -- excess stuck at an ambiguous node's mate is rounding residue --, but
      // that is an argument; this is the check (the same INFTY, the same summation order)
      const double tcap = ldc(g.ex + i) - ldc(g.snk + i);
      double c1 = -tcap, c2 = tcap;
      for (int a = g.aptr[i]; a < g.aptr[i + 1]; ++a) { c1 += ldc(g.r + a); c2 += ldc(g.r + g.rev[a]); }
      const double INFTY = (c1 > c2 ? c1 : c2) + 1;
      return ldc(g.ex + im) - ldc(g.snk + im) - INFTY < 0;
    };
    int mine = N;
    for (int j = improve_from + first; j < N; j += stride) {
      const int i = improve_perm[j];
      if ((ldc(g.h + i) < n) != (ldc(g.h + i + N) < n)) continue;
      if (!is_trivial(i)) { mine = j; break; }   // (ascending j: the first one is this thread's smallest)
    }
    if (threadIdx.x == 0) s_red = N;
    __syncthreads();
    if (mine < N) atomicMin(&s_red, mine);
    __syncthreads();
    if (threadIdx.x == 0 && s_red < N) __hip_atomic_fetch_min(word, s_red, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (!grid_sync(ctl, gen)) return;
    const int next = ldc(word);
    // the trivial steps in front of it, all at once
    {
      int done = 0;
      for (int j = improve_from + first; j < (next < N ? next : N); j += stride) {
        const int i = improve_perm[j];
        if ((ldc(g.h + i) < n) != (ldc(g.h + i + N) < n)) continue;
        fix_to_zero(i, true);
        ++done;
      }
      if (done && g.counters) atomicAdd(g.counters + 1302, done);   // (statistics)
    }
    if (next >= N) break;
    // (the argument for `confined` needs the node that receives the excess to be cut off from the sink: "neither
    // side connected".  The other ambiguous case, both sides connected, sends the new excess through nodes that
    // do have a path and lengthens distances there: those steps relabel from scratch.)
    fixed_node = improve_perm[next];
    keep_valid = g.keep != nullptr && ldc(g.keep + fixed_node) >= n;
    if (blockIdx.x == 0 && threadIdx.x == 0 && g.counters) g.counters[keep_valid ? 1300 : 1301] += 1;   // (statistics: steps of either kind)
    if (blockIdx.x == 0 && threadIdx.x == 0) fix_to_zero(fixed_node, false);
    improve_from = next + 1;
    ++improve_steps;
    improve_steps_dbg = improve_steps;
    incremental = 1;   // the heights stay a valid labelling when a unary term changes: warm search
    solve = true;
    // the new terminal capacities -- of the fixed node and its mate, and of the trivial steps in front of it, whose
    // mates' heights other workgroups wrote as well: everybody meets them behind a grid barrier
    if (!grid_sync(ctl, gen)) return;
  }
  }  // for (;;)
}

// ---- Improve (QPBO_extra.cpp:1151-1233) helpers: the loop over the rand() permutation stays on the
// host, but what it needs per step -- the next node of the permutation that is still not strongly
// labelled, and the large unary term that fixes it -- is computed where the data lives.
__global__ void qpbo_next_ambiguous_kernel(const int32_t *perm, int from, int N, int n, const int32_t *h, int32_t *out) {
  const int j = from + blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= N) return;
  const int i = perm[j];
  if ((h[i] < n) == (h[i + N] < n)) atomicMin(out, j);  // both or neither side reaches the sink
}

// AddUnaryTerm(i, 0, INFTY) with INFTY = max(-t_i + sum of outgoing residuals, t_i + sum of incoming) + 1
// evaluated on node i (QPBO_extra.cpp:241-254, :1177-1187), same summation order as the reference
__global__ void qpbo_fix_to_zero_kernel(QpboDev g, int i, int N) {
  if (blockIdx.x != 0 || threadIdx.x != 0) return;
  const int im = i + N;
  const double tcap = g.ex[i] - g.snk[i];
  double c1 = -tcap, c2 = tcap;
  for (int a = g.aptr[i]; a < g.aptr[i + 1]; ++a) { c1 += g.r[a]; c2 += g.r[g.rev[a]]; }
  const double INFTY = (c1 > c2 ? c1 : c2) + 1;
  const double t0 = g.ex[i] - g.snk[i] + INFTY, t1 = g.ex[im] - g.snk[im] - INFTY;
  g.ex[i] = t0 > 0 ? t0 : 0; g.snk[i] = t0 < 0 ? -t0 : 0;
  g.ex[im] = t1 > 0 ? t1 : 0; g.snk[im] = t1 < 0 ? -t1 : 0;
}

// ---- device-side construction for the plan API ------------------------------------------
// The doubled graph has a fixed slot layout per neighbour pair (one outgoing arc at each of
// i, j, i', j'); only heads, reverse links and capacities depend on whether the summed table
// is submodular, so a fusion move re-builds the whole network with two streaming kernels.
struct RdPlanDev {
  int N, npairs;
  const int32_t *pair_i, *pair_j, *pe_ptr, *pe_edge;  // pe_edge: edge id * 2 + transposed bit
  const int32_t *slots;                                // 4 per pair: out of i, j, i', j'
  const int32_t *aptr;                                 // CSR by doubled node
  const int32_t *slot_pair;                            // per slot: pair * 2 + role (0 = i side, 1 = j side)
  const double *U0, *U1, *E00, *E01, *E10, *E11;
  int32_t *head, *rev;
  double *r, *ci, *cjs, *konst, *ex, *snk, *trv, *u0;
};

__global__ __launch_bounds__(kQB) void rd_pairs_kernel(RdPlanDev d) {
  const int k = blockIdx.x * kQB + threadIdx.x;
  if (k >= d.npairs) return;
  double A = 0, B = 0, C = 0, D = 0;
  for (int q = d.pe_ptr[k]; q < d.pe_ptr[k + 1]; ++q) {
    const int e = d.pe_edge[q] >> 1, tr = d.pe_edge[q] & 1;
    A += d.E00[e]; D += d.E11[e];
    if (tr) { B += d.E10[e]; C += d.E01[e]; } else { B += d.E01[e]; C += d.E10[e]; }
  }
  const bool sub = B + C >= A + D;  // QPBO.cpp:434
  double a = sub ? A : B, b = sub ? B : A, c = sub ? C : D, dd = sub ? D : C;
  double ci = dd - a, cj, cij, cji;   // QPBO.h:760-807
  b -= a; c -= dd;
  if (b < 0) { ci += -b; cj = b; cji = b + c; cij = 0; }
  else if (c < 0) { ci += c; cj = -c; cij = b + c; cji = 0; }
  else { cj = 0; cij = b; cji = c; }
  d.ci[k] = ci;
  d.cjs[k] = sub ? cj : -cj;
  d.konst[k] = sub ? A : B + cj;
  const int i = d.pair_i[k], j = d.pair_j[k], N = d.N;
  const int si = d.slots[4 * k], sj = d.slots[4 * k + 1], sim = d.slots[4 * k + 2], sjm = d.slots[4 * k + 3];
  if (sub) {  // i->j (cij), j->i (cji), j'->i' (cij), i'->j' (cji)
    d.head[si] = j; d.rev[si] = sj; d.r[si] = cij;
    d.head[sj] = i; d.rev[sj] = si; d.r[sj] = cji;
    d.head[sjm] = i + N; d.rev[sjm] = sim; d.r[sjm] = cij;
    d.head[sim] = j + N; d.rev[sim] = sjm; d.r[sim] = cji;
  } else {    // i->j' (cij), j'->i (cji), j->i' (cij), i'->j (cji)
    d.head[si] = j + N; d.rev[si] = sjm; d.r[si] = cij;
    d.head[sjm] = i; d.rev[sjm] = si; d.r[sjm] = cji;
    d.head[sj] = i + N; d.rev[sj] = sim; d.r[sj] = cij;
    d.head[sim] = j; d.rev[sim] = sj; d.r[sim] = cji;
  }
}

__global__ __launch_bounds__(kQB) void rd_nodes_kernel(RdPlanDev d) {
  const int v = blockIdx.x * kQB + threadIdx.x;
  if (v >= d.N) return;
  double t = 0;
  for (int s = d.aptr[v]; s < d.aptr[v + 1]; ++s) {
    const int pr = d.slot_pair[s];
    t += (pr & 1) ? d.cjs[pr >> 1] : d.ci[pr >> 1];
  }
  t += d.U1[v] - d.U0[v];  // QPBO.h:615-624
  d.trv[v] = t;
  d.ex[v] = t > 0 ? t : 0.0; d.snk[v] = t < 0 ? -t : 0.0;
  d.ex[v + d.N] = t < 0 ? -t : 0.0; d.snk[v + d.N] = t > 0 ? t : 0.0;  // mate: -t (QPBO.cpp:689)
}

// strong persistency on the device (QPBO.cpp:840-844): x_i = 1 iff i reaches the sink, 0 iff its
// mate does, -1 if both or neither; counts the -1s
__global__ __launch_bounds__(kQB) void rd_labels_kernel(int64_t N, int n, const int32_t *h, int8_t *label,
                                                       int32_t *unlabelled, int32_t *free_list = nullptr) {
  const int64_t i = (int64_t)blockIdx.x * kQB + threadIdx.x;
  if (i >= N) return;
  const int li = h[i] < n ? 1 : 0, lm = h[i + N] < n ? 1 : 0;
  const int l = li == lm ? -1 : li;
  label[i] = (int8_t)l;
  if (l < 0) {
    const int slot = atomicAdd(unlabelled, 1);
    if (free_list) free_list[slot] = (int32_t)i;   // (in no particular order: the host sorts the few there are)
  }
}

// The residual arcs of the free nodes (both nodes of every unlabelled variable, ids ascending), for the weak
// persistency pass on the host: per node its arcs in list order -- head and two bits, residual capacity on the arc
// and on its reverse.  Only this travels, not the residual network.
__global__ __launch_bounds__(kQB) void rd_free_arcs_kernel(int cnt, int N, const int32_t *ids, const int32_t *aptr,
                                                          const int32_t *head, const int32_t *rev, const double *r,
                                                          int maxdeg, int32_t *out_head, uint8_t *out_flag) {
  const int t = blockIdx.x * kQB + threadIdx.x;
  if (t >= 2 * cnt) return;
  const int v = t < cnt ? ids[t] : ids[t - cnt] + N;
  const int a0 = aptr[v], deg = aptr[v + 1] - a0;
  for (int k = 0; k < maxdeg; ++k) {
    int w = -1;
    uint8_t f = 0;
    if (k < deg) {
      const int a = a0 + k;
      w = head[a];
      f = (uint8_t)((r[a] > 0 ? 1 : 0) | (r[rev[a]] > 0 ? 2 : 0));
    }
    out_head[(size_t)t * maxdeg + k] = w;
    out_flag[(size_t)t * maxdeg + k] = f;
  }
}

__global__ __launch_bounds__(kQB) void rd_set_labels_kernel(int cnt, const int32_t *ids, const int8_t *lab, int8_t *label) {
  const int t = blockIdx.x * kQB + threadIdx.x;
  if (t < cnt) label[ids[t]] = lab[t];
}

// fixed-shape reduction (same tree for every run): partial[b] = sum of a 2048-element chunk
__global__ __launch_bounds__(kQB) void det_sum_kernel(const double *x, int64_t n, double *partial) {
  __shared__ double sh[kQB];
  const int64_t base = (int64_t)blockIdx.x * (kQB * 8);
  double acc = 0;
  for (int k = 0; k < 8; ++k) {
    const int64_t i = base + (int64_t)k * kQB + threadIdx.x;
    if (i < n) acc += x[i];
  }
  sh[threadIdx.x] = acc;
  __syncthreads();
  for (int s = kQB / 2; s > 0; s >>= 1) {
    if ((int)threadIdx.x < s) sh[threadIdx.x] += sh[threadIdx.x + s];
    __syncthreads();
  }
  if (threadIdx.x == 0) partial[blockIdx.x] = sh[0];
}

// flow absorbed by every node's sink arc: initial minus residual capacity
__global__ __launch_bounds__(kQB) void rd_absorbed_kernel(int n, const double *snk0, const double *snk, double *out) {
  const int v = blockIdx.x * kQB + threadIdx.x;
  if (v < n) out[v] = snk0[v] - snk[v];
}

// energy of a labelling from the caller's tables: per-node and per-edge terms
__global__ __launch_bounds__(kQB) void rd_energy_terms_kernel(int64_t N, int64_t E, const uint32_t *conn,
                                                              const int32_t *h, int n, const double *U0,
                                                              const double *U1, const double *E00,
                                                              const double *E01, const double *E10,
                                                              const double *E11, const int8_t *label,
                                                              double *terms) {
  const int64_t t = (int64_t)blockIdx.x * kQB + threadIdx.x;
  if (t < N) {
    terms[t] = label[t] == 1 ? U1[t] : U0[t];
  } else if (t < N + E) {
    const int64_t e = t - N;
    const int xi = label[conn[2 * e]] == 1, xj = label[conn[2 * e + 1]] == 1;
    terms[t] = xi ? (xj ? E11[e] : E10[e]) : (xj ? E01[e] : E00[e]);
  }
}

// ---- host-side construction ----------------------------------------------

struct Pair {
  int32_t i, j;  // i < j
  double A, B, C, D;
};

// QPBO.h:760-807
inline void compute_weights(double A, double B, double C, double D, double &ci, double &cj, double &cij,
                            double &cji) {
  ci = D - A;
  B -= A;
  C -= D;
  if (B < 0) { ci += -B; cj = B; cji = B + C; cij = 0; }
  else if (C < 0) { ci += C; cj = -C; cij = B + C; cji = 0; }
  else { cj = 0; cij = B; cji = C; }
}

struct QpboProblem {
  int64_t N = 0;
  std::vector<int32_t> aptr, head;  // doubled graph, arcs grouped by tail; reverse of a is rev[a]
  std::vector<int32_t> rev;
  std::vector<double> cap, tr;      // initial residuals, terminal capacities (2N)
  std::vector<uint8_t> pair_super;  // per pair
  double const0 = 0;                // constant of the normal form: sum E0 + sum_sub A + sum_super B
};

}  // namespace
}  // namespace stereo

using namespace stereo;

namespace {

// Builds the doubled graph.  Arcs are sorted by tail so that the kernels can use `a` both
// as CSR position and as arc id; the reverse arc is found through `rev`.
bool build_problem(const double *U0, const double *U1, const double *E00, const double *E01,
                   const double *E10, const double *E11, const uint32_t *conn, int64_t N, int64_t E,
                   QpboProblem &P, std::string &err) {
  if (N <= 0) { err = "stereo_rd: no nodes"; return false; }
  if (2 * N >= INT32_MAX / 2 || E >= INT32_MAX / 4) { err = "stereo_rd: problem too large for 32-bit ids"; return false; }
  P.N = N;
  // merge parallel / antiparallel edges: key (lo, hi), tables oriented lo -> hi, summed in input order
  std::vector<int64_t> order(E);
  std::iota(order.begin(), order.end(), 0);
  auto key = [&](int64_t e) {
    const uint32_t a = conn[2 * e], b = conn[2 * e + 1];
    return ((uint64_t)std::min(a, b) << 32) | std::max(a, b);
  };
  for (int64_t e = 0; e < E; ++e) {
    const uint32_t a = conn[2 * e], b = conn[2 * e + 1];
    if (a >= (uint64_t)N || b >= (uint64_t)N) { err = "connectivity index out of range"; return false; }
    if (a == b) { err = "stereo_rd: self loops are not supported"; return false; }
  }
  std::stable_sort(order.begin(), order.end(), [&](int64_t x, int64_t y) { return key(x) < key(y); });
  std::vector<Pair> pairs;
  pairs.reserve(E / 2 + 1);
  double const0 = 0;
  for (int64_t k = 0; k < E; ++k) {
    const int64_t e = order[k];
    const uint32_t a = conn[2 * e], b = conn[2 * e + 1];
    const int32_t lo = (int32_t)std::min(a, b), hi = (int32_t)std::max(a, b);
    if (pairs.empty() || pairs.back().i != lo || pairs.back().j != hi) pairs.push_back(Pair{lo, hi, 0, 0, 0, 0});
    Pair &p = pairs.back();
    // first index of the table = label of conn(0,e) (rd_mex.cpp:59)
    if ((int32_t)a == lo) { p.A += E00[e]; p.B += E01[e]; p.C += E10[e]; p.D += E11[e]; }
    else { p.A += E00[e]; p.B += E10[e]; p.C += E01[e]; p.D += E11[e]; }
  }
  const int64_t n = 2 * N, np = (int64_t)pairs.size();
  P.tr.assign(n, 0.0);
  P.pair_super.assign(np, 0);
  std::vector<int32_t> deg(n + 1, 0);
  struct ArcTmp { int32_t tail, head; double cap; };
  std::vector<ArcTmp> arcs;
  arcs.reserve(4 * np);
  for (int64_t k = 0; k < np; ++k) {
    const Pair &p = pairs[k];
    double ci, cj, cij, cji;
    const int32_t i = p.i, j = p.j, im = i + (int32_t)N, jm = j + (int32_t)N;
    if (p.B + p.C >= p.A + p.D) {  // QPBO.cpp:434
      compute_weights(p.A, p.B, p.C, p.D, ci, cj, cij, cji);
      const0 += p.A;  // constant of the normal form
      P.tr[i] += ci; P.tr[j] += cj;
      arcs.push_back({i, j, cij}); arcs.push_back({j, i, cji});
      arcs.push_back({jm, im, cij}); arcs.push_back({im, jm, cji});
    } else {
      P.pair_super[k] = 1;
      compute_weights(p.B, p.A, p.D, p.C, ci, cj, cij, cji);
      // normal form of phi(x_i, y) = theta(x_i, 1 - y): constant theta(0,1), and the unary
      // cj * y = cj - cj * x_j contributes cj to the constant as well
      const0 += p.B + cj;
      P.tr[i] += ci; P.tr[j] -= cj;
      arcs.push_back({i, jm, cij}); arcs.push_back({jm, i, cji});
      arcs.push_back({j, im, cij}); arcs.push_back({im, j, cji});
    }
  }
  for (int64_t u = 0; u < N; ++u) {
    P.tr[u] += U1[u] - U0[u];  // QPBO.h:615-624
    const0 += U0[u];
  }
  for (int64_t u = 0; u < N; ++u) P.tr[u + N] = -P.tr[u];  // QPBO.cpp:689
  P.const0 = const0;
  // group arcs by tail (counting sort), remember where each arc's reverse lands
  const int64_t m = (int64_t)arcs.size();
  for (const ArcTmp &a : arcs) ++deg[a.tail + 1];
  P.aptr.assign(n + 1, 0);
  for (int64_t v = 0; v < n; ++v) P.aptr[v + 1] = P.aptr[v] + deg[v + 1];
  std::vector<int32_t> fill(P.aptr.begin(), P.aptr.end() - 1), newid(m);
  for (int64_t a = 0; a < m; ++a) newid[a] = fill[arcs[a].tail]++;
  P.head.resize(m); P.cap.resize(m); P.rev.resize(m);
  for (int64_t a = 0; a < m; ++a) {
    P.head[newid[a]] = arcs[a].head;
    P.cap[newid[a]] = arcs[a].cap;
    P.rev[newid[a]] = newid[a ^ 1];
  }
  return true;
}

}  // namespace

// ------------------------------------------------------------------ solver

namespace {

struct QpboSolver {
  QpboProblem P;
  int n = 0, m = 0;
  DevBuf<int32_t> d_aptr, d_head, d_rev, d_h, d_h2, d_cnt, d_flags, d_ctl, d_perm, d_posof, d_dirty, d_keep, d_tab, d_iperm;
  DevBuf<double> d_r, d_delta, d_ex, d_snk;
  DevBuf<unsigned long long> d_hx;
  std::vector<double> snk0;
  QpboDev g{};
  int64_t iterations = 0, relabels = 0;
  int max_degree = 0;  // arcs per doubled node (4 on the 4-connected grid)

  void upload() {
    n = (int)(2 * P.N); m = (int)P.head.size();
    d_aptr.upload(P.aptr.data(), P.aptr.size());
    d_head.upload(P.head.data(), P.head.size());
    d_rev.upload(P.rev.data(), P.rev.size());
    d_r.upload(P.cap.data(), P.cap.size());
    d_delta.alloc((size_t)2 * std::max(m, 1));
    STEREO_HIP_CHECK(hipMemset(d_delta.p, 0, sizeof(double) * 2 * std::max(m, 1)));
    max_degree = 0;
    for (int v = 0; v < n; ++v) max_degree = std::max(max_degree, (int)(P.aptr[v + 1] - P.aptr[v]));
    std::vector<double> ex(n), snk(n);
    for (int v = 0; v < n; ++v) {  // QPBO_maxflow.cpp:135-150: tr_cap > 0 source arc, < 0 sink arc
      ex[v] = P.tr[v] > 0 ? P.tr[v] : 0.0;
      snk[v] = P.tr[v] < 0 ? -P.tr[v] : 0.0;
    }
    snk0 = snk;
    d_ex.upload(ex.data(), n); d_snk.upload(snk.data(), n);
    d_h.alloc(n); d_h2.alloc(n); d_cnt.alloc(2048);
    STEREO_HIP_CHECK(hipMemset(d_cnt.p, 0, sizeof(int32_t) * 2048));
    g.n = n; g.m = m; g.aptr = d_aptr.p; g.head = d_head.p; g.rev = d_rev.p; g.r = d_r.p;
    g.delta = d_delta.p; g.ex = d_ex.p; g.snk = d_snk.p; g.h = d_h.p; g.h2 = d_h2.p; g.counters = d_cnt.p;
    set_tiling(P.N, 0, 0);
    STEREO_HIP_CHECK(hipDeviceSynchronize());
  }

  int grid() const { return (n + kQB - 1) / kQB; }

  // Tiles of the block-local relabelling.  With the grid shape (pixels numbered col * H + row):
  // a 16 column x 32 row patch and its mates; otherwise 512 consecutive nodes and their mates.
  void set_tiling(int64_t Nn, int H, int W) {
    const int half = kMB / 2;
    std::vector<int32_t> perm, posof(2 * Nn, -1);
    if (H > 0 && W > 0 && (int64_t)H * W == Nn) {
      const int tw = 16, th = half / tw, tc = (W + tw - 1) / tw, tr = (H + th - 1) / th;
      perm.assign((size_t)tc * tr * kMB, -1);
      for (int64_t v = 0; v < Nn; ++v) {
        const int col = (int)(v / H), row = (int)(v % H);
        const int64_t T = (int64_t)(col / tw) * tr + row / th;
        const int sl = (col % tw) * th + row % th;
        perm[T * kMB + sl] = (int32_t)v; perm[T * kMB + half + sl] = (int32_t)(v + Nn);
      }
    } else {
      const int64_t nt = (Nn + half - 1) / half;
      perm.assign((size_t)nt * kMB, -1);
      for (int64_t v = 0; v < Nn; ++v) {
        perm[(v / half) * kMB + v % half] = (int32_t)v;
        perm[(v / half) * kMB + half + v % half] = (int32_t)(v + Nn);
      }
    }
    for (size_t q = 0; q < perm.size(); ++q)
      if (perm[q] >= 0) posof[perm[q]] = (int32_t)q;
    d_perm.upload(perm.data(), perm.size()); d_posof.upload(posof.data(), posof.size());
    g.perm = d_perm.p; g.pos_of = d_posof.p; g.ntiles = (int)(perm.size() / kMB);
    d_dirty.alloc((size_t)4 * std::max(g.ntiles, 1));
    STEREO_HIP_CHECK(hipMemset(d_dirty.p, 0, sizeof(int32_t) * 4 * std::max(g.ntiles, 1)));
    g.dirty = d_dirty.p; g.rdirty = d_dirty.p + (size_t)2 * std::max(g.ntiles, 1);
    // the arc table of the tiled rounds (QpboDev::tab): graphs with at most four arcs per node
    g.hx = nullptr;
    if (max_degree <= 4 && 2 * Nn < (1 << 24)) {   // tiled rounds (heights fit the 24-bit fields of an hx word)
      d_hx.alloc((size_t)2 * Nn);
      g.hx = d_hx.p;
    }
    g.tab = nullptr;
    if (max_degree <= 4 && !std::getenv("STEREO_HIP_QPBO_NO_TABLE")) {
      d_tab.alloc((size_t)13 * perm.size());
      g.tab = d_tab.p;
    }
  }

  // AddUnaryTerm(i, 0, INFTY) with INFTY = 1 + max over the two saturation sums of node i
  // (QPBO_extra.cpp:241-254, :1185-1199), applied to the push-relabel state: more source
  // capacity at i, more sink capacity at its mate.
  void fix_to_zero(int i) {
    hipLaunchKernelGGL(qpbo_fix_to_zero_kernel, dim3(1), dim3(64), 0, 0, g, i, (int)P.N);
    STEREO_HIP_CHECK(hipGetLastError());
  }

  // The Improve loop over a permutation of the nodes (QPBO_extra.cpp:1190-1200): every node that is
  // not strongly labelled when its turn comes is fixed to 0 and the flow is maximised again.
  // `h` returns the final heights.
  void improve(const std::vector<int32_t> &perm, std::vector<int32_t> &h) {
    // one cooperative launch walks the whole permutation (qpbo_maxflow_kernel, improve_perm);
    // STEREO_HIP_QPBO_IMPROVE_HOST=1 keeps round 2's host loop (one launch per fixed node) for comparison
    // (kept across calls: hipMalloc / hipFree per Improve call cost more than the upload)
    if (d_iperm.n < perm.size()) d_iperm.alloc(perm.size());
    STEREO_HIP_CHECK(hipMemcpy(d_iperm.p, perm.data(), sizeof(int32_t) * perm.size(), hipMemcpyHostToDevice));
    DevBuf<int32_t> &d_perm = d_iperm;
    if (!std::getenv("STEREO_HIP_QPBO_IMPROVE_HOST")) {
      // the relabellings inside an Improve step start from the heights of the flow the step began with
      // (qpbo_maxflow_kernel, `confined`); STEREO_HIP_QPBO_CONFINED=0: every one of them from scratch (round 3)
      const char *e = std::getenv("STEREO_HIP_QPBO_CONFINED");
      if (!e || std::atoi(e) != 0) {
        if (d_keep.n < (size_t)n + (size_t)std::max(g.ntiles, 1)) d_keep.alloc((size_t)n + (size_t)std::max(g.ntiles, 1));   // heights + one mark per tile
        g.keep = d_keep.p;
      }
      try {
        maxflow(true, d_perm.p);
      } catch (...) {
        g.keep = nullptr;
        throw;
      }
      g.keep = nullptr;
    } else {
      const int N = (int)P.N;
      DevBuf<int32_t> d_next;
      d_next.alloc(1);
      for (int from = 0; from < N;) {
        int32_t next = N;
        STEREO_HIP_CHECK(hipMemcpyAsync(d_next.p, &next, sizeof(next), hipMemcpyHostToDevice, 0));
        hipLaunchKernelGGL(qpbo_next_ambiguous_kernel, dim3((unsigned)((N - from + 255) / 256)), dim3(256), 0, 0,
                           d_perm.p, from, N, n, g.h, d_next.p);
        STEREO_HIP_CHECK(hipMemcpy(&next, d_next.p, sizeof(next), hipMemcpyDeviceToHost));
        if (next >= N) break;
        fix_to_zero(perm[next]);
        maxflow(true);
        from = next + 1;
      }
    }
    h.resize(n);
    STEREO_HIP_CHECK(hipMemcpy(h.data(), g.h, sizeof(int32_t) * n, hipMemcpyDeviceToHost));
  }

  // One cooperative launch runs the whole max-flow (qpbo_maxflow_kernel).
  void maxflow(bool warm = false, const int32_t *improve_perm = nullptr) {
    int relabel_every = 256;
    if (const char *e = std::getenv("STEREO_HIP_QPBO_RELABEL_EVERY")) relabel_every = std::max(1, std::atoi(e));
    if (d_ctl.n < (size_t)QpboCtl::kWords) d_ctl.alloc(QpboCtl::kWords);
    STEREO_HIP_CHECK(hipMemsetAsync(d_ctl.p, 0, sizeof(int32_t) * QpboCtl::kWords, 0));
    // tile marks are round numbers counted from the start of a launch: none left over from the last one
    if (d_dirty.p) STEREO_HIP_CHECK(hipMemsetAsync(d_dirty.p, 0, sizeof(int32_t) * d_dirty.n, 0));
    // (device properties and the occupancy of the kernel are asked once per process and device:
    // an Improve pass calls this function hundreds of times)
    // (thread local: stereo_hip_set_device selects a device per host thread, and two threads
    // driving different devices must not see each other's half-written entries)
    static thread_local int cached_dev = -1, cached_cus = 0, cached_per_cu = 0;
    int dev = 0;
    STEREO_HIP_CHECK(hipGetDevice(&dev));
    if (dev != cached_dev) {
      int c = 0, pc = 0;
      STEREO_HIP_CHECK(hipDeviceGetAttribute(&c, hipDeviceAttributeMultiprocessorCount, dev));
      STEREO_HIP_CHECK(hipFuncSetAttribute((const void *)qpbo_maxflow_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)(sizeof(double) * 4 * kMB)));
      STEREO_HIP_CHECK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&pc, qpbo_maxflow_kernel, kMB, sizeof(double) * 4 * kMB));
      cached_cus = c; cached_per_cu = pc; cached_dev = dev;
    }
    const int cus = cached_cus, per_cu = cached_per_cu;
    if (per_cu < 1) throw HipError{"qpbo_maxflow_kernel does not fit on a CU"};
    int blocks = std::min(cus * std::min(per_cu, 2), std::max(g.ntiles, 1));
    blocks = std::max(blocks, 1);
    // The Improve loop is thousands of tiny steps, each a handful of grid barriers over a few touched tiles: its cost is
    // the barrier, which grows with the number of workgroups that meet there (swept on the Teddy pair's example_global
    // moves, tools/sweep_qpbo.sh).  STEREO_HIP_QPBO_IMPROVE_BLOCKS overrides.
    if (improve_perm) {
      int ib = kImproveBlocks;
      if (const char *e = std::getenv("STEREO_HIP_QPBO_IMPROVE_BLOCKS")) ib = std::max(1, std::atoi(e));
      blocks = std::min(blocks, ib);
    }
    QpboDev gg = g;
    int32_t *ctl = d_ctl.p;
    int max_rounds = 1 << 21;
    // tiled rounds: every node has at most four arcs and the pushed-amount buffer has two halves
    int tiled = (max_degree <= 4 && g.hx != nullptr && d_delta.n >= (size_t)2 * std::max(m, 1)) ? 16 : 0;
    if (const char *e = std::getenv("STEREO_HIP_QPBO_TILED")) tiled = tiled ? std::max(0, std::atoi(e)) : 0;
    const size_t dyn = tiled ? sizeof(double) * 4 * kMB : 0;
    int switch_at = 16;  // plain rounds first: most moves end within a dozen of them
    if (const char *e = std::getenv("STEREO_HIP_QPBO_SWITCH")) switch_at = std::max(0, std::atoi(e));
    int incremental = warm ? 1 : 0;
    if (const char *e = std::getenv("STEREO_HIP_QPBO_WARM")) incremental = incremental && std::atoi(e) != 0;
    int first_interval = 4;  // first relabelling after four rounds: by then the excess that is cut off from the sink just climbs
    if (const char *e = std::getenv("STEREO_HIP_QPBO_FIRST_INTERVAL")) first_interval = std::max(1, std::atoi(e));
    int adaptive = 1;
    if (const char *e = std::getenv("STEREO_HIP_QPBO_ADAPTIVE")) adaptive = std::atoi(e) != 0;
    // a round that retires less than 1 % of the active nodes counts as stagnant (two of them bring the next exact
    // relabelling forward): excess that is cut off climbs one level per round, a trickle of it still reaching the
    // sink is not progress (swept 0 / 2 / 5 / 10 / 20 / 50: 107 -> 112 moves/s on example_global from 10 on, same labels up to 20)
    int stall = 10;
    if (const char *e = std::getenv("STEREO_HIP_QPBO_STALL_PERMILLE")) stall = std::min(999, std::max(0, std::atoi(e)));
    adaptive |= stall << 8;
    if (const char *e = std::getenv("STEREO_HIP_QPBO_FINAL_RELABEL")) if (std::atoi(e) != 0) adaptive |= 1 << 30;
    if (const char *e = std::getenv("STEREO_HIP_QPBO_WRAP_AT")) adaptive |= std::min(4095, std::max(0, std::atoi(e))) << 18;   // (development)
    int improve_N = (int)P.N;
    void *args[] = {&gg, &ctl, &relabel_every, &max_rounds, &tiled, &switch_at, &incremental, &first_interval, &adaptive,
                    &improve_perm, &improve_N};
    STEREO_HIP_CHECK(hipLaunchCooperativeKernel((const void *)qpbo_maxflow_kernel, dim3(blocks), dim3(kMB), args, dyn, 0));
    int32_t host_ctl[QpboCtl::kWords];
    STEREO_HIP_CHECK(hipMemcpy(host_ctl, d_ctl.p, sizeof(host_ctl), hipMemcpyDeviceToHost));
#ifdef STEREO_HIP_QPBO_CHECK_CONFINED
    {
      int32_t dbg[16];
      STEREO_HIP_CHECK(hipMemcpy(dbg, d_cnt.p + 1100, sizeof(dbg), hipMemcpyDeviceToHost));
      if (dbg[8] || dbg[9])
        std::fprintf(stderr, "[stereo_hip qpbo] confined check: local passes: %d nodes left at another height than the full pass (first v=%d has %d wants %d, warm %d), %d active nodes outside the touched tiles\n",
                     dbg[8], dbg[10], dbg[11], dbg[12], dbg[13], dbg[9]);
      if (dbg[0] || dbg[7])
        std::fprintf(stderr, "[stereo_hip qpbo] confined check: %d unsupported nodes, %d heights outside the touched tiles changed; first v=%d (N=%d) h=%d keep=%d excess=%d relabel#%d step %d\n",
                     dbg[0], dbg[7], dbg[1], (int)P.N, dbg[2], dbg[3], dbg[4], dbg[5], dbg[6]);
      STEREO_HIP_CHECK(hipMemset(d_cnt.p + 1100, 0, sizeof(dbg)));
    }
#endif
    if (host_ctl[QpboCtl::kAbort] == 1) throw HipError{"stereo_rd: grid barrier gave up (device-side spin bound)"};
    if (host_ctl[QpboCtl::kAbort] == 2) throw HipError{"stereo_rd: push-relabel did not converge within the round bound"};
    if (std::getenv("STEREO_HIP_QPBO_VERBOSE")) {
      std::fprintf(stderr, "[stereo_hip qpbo] relabels: %.3f ms, %d barriers; tiled section: %.3f ms, %d rounds\n", host_ctl[11] * 1e-5,
                   host_ctl[12], host_ctl[13] * 1e-5, host_ctl[14]);
      if (improve_perm) {
        int32_t st[3] = {0, 0, 0};
        STEREO_HIP_CHECK(hipMemcpy(st, d_cnt.p + 1300, sizeof(st), hipMemcpyDeviceToHost));
        STEREO_HIP_CHECK(hipMemset(d_cnt.p + 1300, 0, sizeof(st)));
        std::fprintf(stderr, "[stereo_hip qpbo] Improve steps: %d trivial (no residual arc into the fixed node's mate: no solve), %d with the fixed node cut off "
                             "from the sink (confined relabellings), %d with both sides connected\n", st[2], st[0], st[1]);
      }
    }
#ifdef STEREO_HIP_QPBO_PROFILE
    if (std::getenv("STEREO_HIP_QPBO_VERBOSE")) {
      int32_t q[16];
      STEREO_HIP_CHECK(hipMemcpy(q, d_cnt.p + 1200, sizeof(q), hipMemcpyDeviceToHost));
      STEREO_HIP_CHECK(hipMemset(d_cnt.p + 1200, 0, sizeof(q)));
      std::fprintf(stderr, "[stereo_hip qpbo] workgroup 0, relabelling: init %.3f ms, tile set-up %.3f ms (%d tiles), relaxation %.3f ms (%d iterations), "
                   "grid barriers %.3f ms (%d steps); tiled rounds: load %.3f ms (%d tiles), local rounds %.3f ms (%d), store %.3f ms, reduction + grid barrier %.3f ms (%d rounds)\n",
                   q[0] * 1e-5, q[1] * 1e-5, q[4], q[2] * 1e-5, q[5], q[3] * 1e-5, q[6], q[8] * 1e-5, q[12], q[9] * 1e-5, q[13], q[10] * 1e-5, q[11] * 1e-5, q[14]);
    }
#endif
    if (std::getenv("STEREO_HIP_QPBO_VERBOSE")) { std::vector<int32_t> tr(1040); (void)hipMemcpy(tr.data(), d_cnt.p, sizeof(int32_t) * 1040, hipMemcpyDeviceToHost); std::fprintf(stderr, "[stereo_hip qpbo] active per round:"); for (int i = 0; i < host_ctl[QpboCtl::kRounds] && i < 1024; i += (i < 32 ? 1 : 8)) std::fprintf(stderr, " %d", tr[16 + i]); std::fprintf(stderr, "\n"); }
    iterations += host_ctl[QpboCtl::kRounds];
    relabels += host_ctl[QpboCtl::kRelabels];
  }
};

// Kosaraju pass of QPBO_postprocessing.cpp:10-120 on the unlabelled nodes and their mates.
// The component numbering among INCOMPARABLE components follows this file's arc order, not
// the reference's linked-list order (those labels are an artefact of DFS order there too).
void weak_persistencies(const QpboProblem &P, const std::vector<double> &r, std::vector<int> &label) {
  const int64_t N = P.N, n = 2 * N;
  std::vector<int32_t> region(n, 0), parent(n, 0), cursor(n, 0);
  std::vector<uint8_t> seen(n, 1);
  bool any = false;
  for (int64_t i = 0; i < N; ++i)
    if (label[i] < 0) { seen[i] = seen[i + N] = 0; region[i] = region[i + N] = -1; any = true; }
  if (!any) return;
  std::vector<int32_t> stack;  // finish order
  for (int64_t s = 0; s < n; ++s) {
    if (seen[s]) continue;
    int32_t i = (int32_t)s;
    seen[i] = 1; parent[i] = i; cursor[i] = P.aptr[i];
    for (;;) {
      if (cursor[i] == P.aptr[i + 1]) {
        stack.push_back(i);
        if (parent[i] == i) break;
        i = parent[i];
        ++cursor[i];
        continue;
      }
      const int32_t a = cursor[i], j = P.head[a];
      if (!(r[a] > 0) || seen[j]) { ++cursor[i]; continue; }
      seen[j] = 1; parent[j] = i; i = j; cursor[i] = P.aptr[i];
    }
  }
  int component = 0;
  for (int64_t k = (int64_t)stack.size() - 1; k >= 0; --k) {
    int32_t i = stack[k];
    if (region[i] > 0) continue;
    region[i] = ++component; parent[i] = i; cursor[i] = P.aptr[i];
    for (;;) {
      if (cursor[i] == P.aptr[i + 1]) {
        if (parent[i] == i) break;
        i = parent[i];
        ++cursor[i];
        continue;
      }
      const int32_t a = cursor[i], j = P.head[a];
      if (!(r[P.rev[a]] > 0) || region[j] >= 0) { ++cursor[i]; continue; }
      parent[j] = i; i = j; cursor[i] = P.aptr[i]; region[i] = component;
    }
  }
  for (int64_t i = 0; i < N; ++i)
    if (label[i] < 0) {
      if (region[i] > region[i + N]) label[i] = 0;
      else if (region[i] < region[i + N]) label[i] = 1;
    }
}

// The same pass on the free subgraph alone (rd_free_arcs_kernel): local node t < cnt is variable ids[t], t >= cnt its
// mate; arcs in the order of the arc lists, a head that is not free ends nowhere.  Visiting order, arc order and
// numbering are those of weak_persistencies, so the labels are the same.  lab[t]: new label of variable ids[t].
void weak_persistencies_compact(int cnt, int N, const std::vector<int32_t> &ids, int maxdeg,
                                const std::vector<int32_t> &heads, const std::vector<uint8_t> &flags,
                                std::vector<int8_t> &lab) {
  const int M = 2 * cnt;
  std::vector<int32_t> loc((size_t)M * maxdeg, -1);   // local index of every arc's head, -1: not free / no arc
  for (int t = 0; t < M; ++t)
    for (int k = 0; k < maxdeg; ++k) {
      const int32_t w = heads[(size_t)t * maxdeg + k];
      if (w < 0) continue;
      const int32_t var = w >= N ? w - N : w;
      const auto it = std::lower_bound(ids.begin(), ids.end(), var);
      if (it != ids.end() && *it == var) loc[(size_t)t * maxdeg + k] = (int32_t)(it - ids.begin()) + (w >= N ? cnt : 0);
    }
  std::vector<int32_t> region(M, -1), parent(M, 0), cursor(M, 0), stack;
  std::vector<uint8_t> seen(M, 0);
  stack.reserve(M);
  for (int s0 = 0; s0 < M; ++s0) {
    if (seen[s0]) continue;
    int i = s0;
    seen[i] = 1; parent[i] = i; cursor[i] = 0;
    for (;;) {
      if (cursor[i] == maxdeg) {
        stack.push_back(i);
        if (parent[i] == i) break;
        i = parent[i];
        ++cursor[i];
        continue;
      }
      const size_t a = (size_t)i * maxdeg + cursor[i];
      const int j = loc[a];
      if (j < 0 || !(flags[a] & 1) || seen[j]) { ++cursor[i]; continue; }
      seen[j] = 1; parent[j] = i; i = j; cursor[i] = 0;
    }
  }
  int component = 0;
  for (int k = (int)stack.size() - 1; k >= 0; --k) {
    int i = stack[k];
    if (region[i] > 0) continue;
    region[i] = ++component; parent[i] = i; cursor[i] = 0;
    for (;;) {
      if (cursor[i] == maxdeg) {
        if (parent[i] == i) break;
        i = parent[i];
        ++cursor[i];
        continue;
      }
      const size_t a = (size_t)i * maxdeg + cursor[i];
      const int j = loc[a];
      if (j < 0 || !(flags[a] & 2) || region[j] >= 0) { ++cursor[i]; continue; }
      parent[j] = i; i = j; cursor[i] = 0; region[i] = component;
    }
  }
  lab.assign(cnt, -1);
  for (int t = 0; t < cnt; ++t) {
    if (region[t] > region[t + cnt]) lab[t] = 0;
    else if (region[t] < region[t + cnt]) lab[t] = 1;
  }
}

// The permutation of QPBO::Improve (QPBO_extra.cpp:13-27), drawn from libc rand() like the
// reference's.  Note for callers that seed rand() to reproduce the reference: the HIP runtime draws
// from the same generator during its one-time initialisation (measured), so initialise it first
// (stereo_hip_warm_up; the Python package does so when it loads the library).
// `count` values of rand(), in order, from the process-wide generator -- the same values and the same generator
// state afterwards as `count` calls of rand(), without glibc's lock per call (2.7 ms of a 3.0 ms permutation at
// 450 x 375; an Improve move pays it on the host).  POSIX semantics only: initstate() parks the generator on a
// scratch array and hands back the array it was running on, position included; random_r() runs on that array
// through a random_data of our own; parking that one writes the advanced position back; setstate() resumes.
// The first use checks itself against rand() on a saved copy of the state (nothing is consumed by the check) and
// falls back to rand() for good if anything differs.  STEREO_HIP_FAST_RAND=0: always rand().
// Threads: the park / resume window swaps the PROCESS-WIDE generator state, so every window of this
// library is serialised by rand_window_mutex() (two host threads running Improve moves would otherwise
// hand each other's park buffers around and leave the generator on a static array for good).  A
// foreign thread that calls rand() inside a window draws from the parked stream (seeded 1) instead of
// the process stream -- the same hazard class as any unsynchronised rand() user, documented in
// INTEGRATION.md.
std::mutex &rand_window_mutex() {
  static std::mutex m;
  return m;
}

// (callers hold rand_window_mutex())
bool draw_rand_fast(int32_t *out, int64_t count) {
  alignas(int32_t) static char park[256];
  alignas(int32_t) static char park_r[256];
  alignas(int32_t) static char scratch[256];
  char *old = initstate(1u, park, sizeof(park));
  if (!old) return false;
  struct random_data rd;
  std::memset(&rd, 0, sizeof(rd));
  bool ok = initstate_r(1u, scratch, sizeof(scratch), &rd) == 0 && setstate_r(old, &rd) == 0;
  if (ok) {
    for (int64_t i = 0; i < count; ++i) { int32_t v; random_r(&rd, &v); out[i] = v; }
    ok = initstate_r(1u, park_r, sizeof(park_r), &rd) == 0;   // (parks rd: the position goes back into `old`)
  }
  setstate(old);
  return ok;
}

// (callers hold rand_window_mutex())
bool fast_rand_usable() {
  static std::atomic<int> state{-1};   // -1 unknown, 0 no, 1 yes
  if (state.load() >= 0) return state.load() == 1;
  state.store(0);
  if (const char *e = std::getenv("STEREO_HIP_FAST_RAND")) if (std::atoi(e) == 0) return false;
  // a copy of the generator's state while it is parked, to undo what the check draws
  alignas(int32_t) static char park[256];
  static const size_t bytes_of_type[5] = {8, 32, 64, 128, 256};   // TYPE_0 .. TYPE_4 arrays incl. the info word
  char *old = initstate(1u, park, sizeof(park));
  if (!old) return false;
  int32_t info;
  std::memcpy(&info, old, sizeof(info));
  const int type = (int)(((info % 5) + 5) % 5);
  const size_t bytes = bytes_of_type[type];
  char saved[256];
  std::memcpy(saved, old, bytes);
  setstate(old);
  int32_t slow[8], fast[8];
  for (int i = 0; i < 8; ++i) slow[i] = rand();
  const int32_t slow_next = rand();
  auto restore = [&]() -> bool {
    char *cur = initstate(1u, park, sizeof(park));
    if (cur != old) { if (cur) setstate(cur); return false; }
    std::memcpy(old, saved, bytes);
    setstate(old);
    return true;
  };
  if (!restore()) return false;
  const bool drew = draw_rand_fast(fast, 8);
  const int32_t fast_next = rand();
  if (!restore()) return false;
  if (drew && std::memcmp(slow, fast, sizeof(slow)) == 0 && slow_next == fast_next) state.store(1);
  return state.load() == 1;
}

std::vector<int32_t> improve_permutation(int64_t N) {
  std::vector<int32_t> perm((size_t)N);
  for (int64_t i = 0; i < N; ++i) perm[i] = (int32_t)i;
  if (N < 2) return perm;
  std::vector<int32_t> r((size_t)(N - 1));
  {
    std::lock_guard<std::mutex> window(rand_window_mutex());   // one permutation's draws are contiguous in the stream
    if (!(fast_rand_usable() && draw_rand_fast(r.data(), N - 1)))
      for (int64_t i = 0; i < N - 1; ++i) r[i] = rand();   // (draw_rand_fast either drew everything or nothing)
  }
  for (int64_t i = 0; i < N - 1; ++i) {
    int64_t j = i + (int64_t)((r[i] / (1.0 + (double)RAND_MAX)) * (double)(N - i));
    if (j > N - 1) j = N - 1;
    std::swap(perm[i], perm[j]);
  }
  return perm;
}

}  // namespace

namespace {

// H if the edges are exactly those of a 4-connected H x W image numbered col * H + row (any order, either
// direction, no edge across a column end), else 0: the work-partition hint of stereo_rd_plan_set_grid.
int64_t grid_height_of(const uint32_t *conn, int64_t N, int64_t E) {
  int64_t H = 0;
  for (int64_t e = 0; e < E; ++e) {
    const int64_t a = conn[2 * e], b = conn[2 * e + 1], d = a < b ? b - a : a - b;
    if (d == 1) continue;
    if (H == 0) H = d;
    if (d != H) return 0;
  }
  if (H < 2 || N % H != 0) return 0;
  for (int64_t e = 0; e < E; ++e) {
    const int64_t a = conn[2 * e], b = conn[2 * e + 1], lo = a < b ? a : b, d = a < b ? b - a : a - b;
    if (d == 1 && lo % H == H - 1) return 0;
  }
  return H;
}

struct RdPlanCache {
  std::mutex mu;
  stereo_rd_plan *plan = nullptr;
  int64_t N = 0, E = 0;
  int device = -1;
  std::vector<uint32_t> conn;
};

RdPlanCache &rd_plan_cache() {
  static RdPlanCache *C = new RdPlanCache;   // (never destroyed: the HIP runtime may be gone before static destructors run)
  return *C;
}

int rd_through_cached_plan(const double *U0, const double *U1, const double *E00, const double *E01, const double *E10,
                           const double *E11, const uint32_t *conn, int64_t N, int64_t E, int improve, double *labelling,
                           double *energy, double *lower_bound, double *num_unlabelled, char *err, size_t errcap) {
  RdPlanCache *C = &rd_plan_cache();
  std::lock_guard<std::mutex> lock(C->mu);
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) return fail("stereo_rd: hipGetDevice failed", err, errcap);
  const bool hit = C->plan && C->N == N && C->E == E && C->device == dev &&
                   std::memcmp(C->conn.data(), conn, sizeof(uint32_t) * 2 * (size_t)E) == 0;
  if (!hit) {
    if (C->plan) { stereo_rd_plan_destroy(C->plan); C->plan = nullptr; }
    stereo_rd_plan *plan = nullptr;
    const int rc = stereo_rd_plan_create(N, E, conn, &plan, err, errcap);
    if (rc != 0) return rc;
    const int64_t H = grid_height_of(conn, N, E);
    if (H > 0 && N / H < INT32_MAX) {
      char e2[128] = {0};
      (void)stereo_rd_plan_set_grid(plan, (int)H, (int)(N / H), e2, sizeof(e2));   // a hint: results do not depend on it
    }
    C->plan = plan; C->N = N; C->E = E; C->device = dev;
    C->conn.assign(conn, conn + 2 * (size_t)E);
  }
  return stereo_rd_plan_solve(C->plan, U0, U1, E00, E01, E10, E11, improve, labelling, energy, lower_bound, num_unlabelled,
                              err, errcap);
}

}  // namespace

extern "C" void stereo_rd_cache_clear(void) {
  RdPlanCache &C = rd_plan_cache();
  std::lock_guard<std::mutex> lock(C.mu);
  if (C.plan) { stereo_rd_plan_destroy(C.plan); C.plan = nullptr; }
  C.conn.clear(); C.conn.shrink_to_fit();
}

extern "C" int stereo_hip_improve_permutation(int64_t N, int32_t *out) {
  if (N < 0 || (N > 0 && !out)) return 1;
  const std::vector<int32_t> perm = improve_permutation(N);
  if (N > 0) std::memcpy(out, perm.data(), sizeof(int32_t) * (size_t)N);
  return 0;
}

extern "C" int stereo_rd(const double *U0, const double *U1, const double *E00, const double *E01,
                         const double *E10, const double *E11, const uint32_t *conn, int64_t N, int64_t E,
                         int improve, double *labelling, double *energy, double *lower_bound,
                         double *num_unlabelled, char *err, size_t errcap) {
  if (!U0 || !U1 || !labelling || !energy || !lower_bound || !num_unlabelled || (E > 0 && (!E00 || !E01 || !E10 || !E11 || !conn)))
    return fail("stereo_rd: NULL argument", err, errcap);
  if (stereo_hip_device_count() < 1)
    return fail("stereo_rd: no HIP device available (the HIP path has no CPU fallback)", err, errcap);
  // The gateway is called once per fusion move with the SAME connectivity (dispmap_super.m:61-84): the plan of the
  // last connectivity seen is kept (edge grouping, slot layout, device buffers), so that from the second move on a
  // call costs what stereo_rd_plan_solve costs -- the upload of the six term arrays and the solve -- instead of
  // rebuilding the graph on the host every time (54 -> 2 ms at 450 x 375).  STEREO_HIP_RD_CACHE=0: build per call.
  if (E > 0 && N > 0) {
    const char *ce = std::getenv("STEREO_HIP_RD_CACHE");
    if (!ce || std::atoi(ce) != 0) return rd_through_cached_plan(U0, U1, E00, E01, E10, E11, conn, N, E, improve, labelling, energy,
                                                                   lower_bound, num_unlabelled, err, errcap);
  }
  try {
    QpboSolver S;
    std::string berr;
    const bool verbose = std::getenv("STEREO_HIP_QPBO_VERBOSE") != nullptr;
    auto now = [] { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
    const double t0 = now();
    if (!build_problem(U0, U1, E00, E01, E10, E11, conn, N, E, S.P, berr)) return fail(berr, err, errcap);
    const double t1 = now();
    S.upload();
    const double t2 = now();
    S.maxflow();
    const double t3 = now();
    if (verbose) std::fprintf(stderr, "[stereo_hip qpbo] build %.2f ms, upload %.2f ms, maxflow %.2f ms\n", t1 - t0, t2 - t1, t3 - t2);
    const int n = S.n;
    std::vector<int32_t> h(n);
    std::vector<double> snk(n);
    STEREO_HIP_CHECK(hipMemcpy(h.data(), S.g.h, sizeof(int32_t) * n, hipMemcpyDeviceToHost));
    STEREO_HIP_CHECK(hipMemcpy(snk.data(), S.d_snk.p, sizeof(double) * n, hipMemcpyDeviceToHost));
    std::vector<int> label(N);
    double unl = 0;
    for (int64_t i = 0; i < N; ++i) {
      const int li = h[i] < n ? 1 : 0, lm = h[i + N] < n ? 1 : 0;  // what_segment: 1 iff in the sink tree
      label[i] = li == lm ? -1 : li;
      if (label[i] < 0) unl += 1;
    }
    if (unl > 0) {
      std::vector<double> r(S.m);
      STEREO_HIP_CHECK(hipMemcpy(r.data(), S.d_r.p, sizeof(double) * S.m, hipMemcpyDeviceToHost));
      weak_persistencies(S.P, r, label);
      unl = 0;
      for (int64_t i = 0; i < N; ++i) if (label[i] < 0) unl += 1;
    }
    if (std::getenv("STEREO_HIP_QPBO_VERBOSE"))
      std::fprintf(stderr, "[stereo_hip qpbo] n=%d arcs=%d iterations=%lld global_relabels=%lld unlabelled=%g\n", S.n, S.m,
                   (long long)S.iterations, (long long)S.relabels, unl);
    *num_unlabelled = unl;  // rd_mex.cpp:83-88: counted before Improve
    // roof-dual bound: const + sum_i min(0, tr_i) + flow/2, flow = what reached the sink
    {
      double flow = 0, neg = 0;
      for (int v = 0; v < n; ++v) flow += S.snk0[v] - snk[v];
      for (int64_t i = 0; i < N; ++i) neg += S.P.tr[i] < 0 ? S.P.tr[i] : 0.0;
      *lower_bound = S.P.const0 + neg + flow / 2;
    }
    if (improve && unl > 0) {
      // QPBO::Improve() (QPBO_extra.cpp:1151-1233) from user labels 0: visit the nodes in a
      // rand() permutation (QPBO_extra.cpp:13-27); a node that is still not strongly labelled
      // is fixed to 0 by a large unary term and the flow is re-maximised incrementally.
      // only nodes without a strong label can ever need fixing; strong labels persist
      S.improve(improve_permutation(N), h);
      for (int64_t i = 0; i < N; ++i) {
        const int li = h[i] < n ? 1 : 0, lm = h[i + N] < n ? 1 : 0;
        label[i] = li == lm ? 0 : li;  // QPBO_extra.cpp:1210-1219: ambiguous -> user label (0)
      }
    }
    // energy of the labelling (unknown -> 0, QPBO.cpp:857), from the caller's own tables
    double en = 0;
    for (int64_t i = 0; i < N; ++i) { labelling[i] = label[i]; en += label[i] == 1 ? U1[i] : U0[i]; }
    for (int64_t e = 0; e < E; ++e) {
      const int xi = label[conn[2 * e]] == 1, xj = label[conn[2 * e + 1]] == 1;
      en += xi ? (xj ? E11[e] : E10[e]) : (xj ? E01[e] : E00[e]);
    }
    *energy = en;
    return 0;
  } catch (const HipError &e) {
    return fail(e.msg, err, errcap);
  } catch (const std::exception &e) {
    return fail(std::string("stereo_rd: ") + e.what(), err, errcap);
  }
}

// ------------------------------------------------------------------ plan API

struct stereo_rd_plan {
  int64_t N = 0, E = 0, npairs = 0;
  std::vector<int32_t> aptr;  // host copy (Improve, weak persistency)
  DevBuf<int32_t> d_pair_i, d_pair_j, d_pe_ptr, d_pe_edge, d_slots, d_slot_pair;
  DevBuf<uint32_t> d_conn;
  DevBuf<double> d_ci, d_cjs, d_konst, d_trv, d_terms, d_partial, d_in[6], d_snk0;
  DevBuf<int8_t> d_label, d_flab;
  DevBuf<int32_t> d_free, d_fids, d_fhead;   // unlabelled variables (unordered | sorted), heads of the free nodes' arcs
  DevBuf<uint8_t> d_fflag;
  QpboSolver S;
};

namespace {

double det_sum(const double *x, int64_t n, DevBuf<double> &partial) {
  if (n <= 0) return 0.0;
  const int64_t nb = (n + kQB * 8 - 1) / (kQB * 8);
  if (partial.n < (size_t)nb) partial.alloc(nb);
  hipLaunchKernelGGL(det_sum_kernel, dim3((unsigned)nb), dim3(kQB), 0, 0, x, n, partial.p);
  std::vector<double> h(nb);
  STEREO_HIP_CHECK(hipMemcpy(h.data(), partial.p, sizeof(double) * nb, hipMemcpyDeviceToHost));
  double s = 0;
  for (double v : h) s += v;  // fixed order
  return s;
}


// Several fixed-shape sums whose results the host only needs later: the kernels are queued as they become
// possible, one copy brings all block partials back (a synchronous copy per sum was eight host round trips per move).
struct DetSums {
  DevBuf<double> &buf;
  std::vector<std::pair<size_t, int64_t>> slots;   // (offset, #partials)
  std::vector<double> host;
  size_t used = 0;
  static int64_t blocks(int64_t n) { return n > 0 ? (n + kQB * 8 - 1) / (kQB * 8) : 0; }
  DetSums(DevBuf<double> &b, int64_t capacity) : buf(b) {
    if (buf.n < (size_t)std::max<int64_t>(capacity, 1)) buf.alloc((size_t)std::max<int64_t>(capacity, 1));
  }
  int queue(const double *x, int64_t n) {
    const int64_t nb = blocks(n);
    if (used + (size_t)nb > buf.n) throw HipError{"DetSums: capacity"};
    if (nb > 0) hipLaunchKernelGGL(det_sum_kernel, dim3((unsigned)nb), dim3(kQB), 0, 0, x, n, buf.p + used);
    slots.push_back({used, nb});
    used += (size_t)nb;
    return (int)slots.size() - 1;
  }
  void fetch() {
    host.resize(used);
    if (used) STEREO_HIP_CHECK(hipMemcpy(host.data(), buf.p, sizeof(double) * used, hipMemcpyDeviceToHost));
  }
  double get(int slot) const {
    double s = 0;
    for (int64_t k = 0; k < slots[slot].second; ++k) s += host[slots[slot].first + k];  // fixed order
    return s;
  }
};
}  // namespace

extern "C" int stereo_rd_plan_create(int64_t N, int64_t E, const uint32_t *conn, stereo_rd_plan **plan, char *err,
                                     size_t errcap) {
  if (!plan) return fail("stereo_rd_plan_create: plan is NULL", err, errcap);
  *plan = nullptr;
  if (N <= 0 || (E > 0 && !conn)) return fail("stereo_rd_plan_create: bad argument", err, errcap);
  if (2 * N >= INT32_MAX / 2 || E >= INT32_MAX / 4) return fail("stereo_rd: problem too large for 32-bit ids", err, errcap);
  if (stereo_hip_device_count() < 1)
    return fail("stereo_rd: no HIP device available (the HIP path has no CPU fallback)", err, errcap);
  try {
    std::unique_ptr<stereo_rd_plan> P(new stereo_rd_plan);
    P->N = N; P->E = E;
    std::vector<int64_t> order(E);
    std::iota(order.begin(), order.end(), 0);
    for (int64_t e = 0; e < E; ++e) {
      const uint32_t a = conn[2 * e], b = conn[2 * e + 1];
      if (a >= (uint64_t)N || b >= (uint64_t)N) return fail("connectivity index out of range", err, errcap);
      if (a == b) return fail("stereo_rd: self loops are not supported", err, errcap);
    }
    auto key = [&](int64_t e) {
      const uint32_t a = conn[2 * e], b = conn[2 * e + 1];
      return ((uint64_t)std::min(a, b) << 32) | std::max(a, b);
    };
    std::stable_sort(order.begin(), order.end(), [&](int64_t x, int64_t y) { return key(x) < key(y); });
    std::vector<int32_t> pi, pj, pe_ptr, pe_edge(E);
    for (int64_t k = 0; k < E; ++k) {
      const int64_t e = order[k];
      const uint32_t a = conn[2 * e], b = conn[2 * e + 1];
      const int32_t lo = (int32_t)std::min(a, b), hi = (int32_t)std::max(a, b);
      if (pi.empty() || pi.back() != lo || pj.back() != hi) { pi.push_back(lo); pj.push_back(hi); pe_ptr.push_back((int32_t)k); }
      pe_edge[k] = (int32_t)(e * 2 + ((int32_t)a == lo ? 0 : 1));
    }
    pe_ptr.push_back((int32_t)E);
    const int64_t np = (int64_t)pi.size(), n = 2 * N, m = 4 * np;
    P->npairs = np;
    // slot layout: every pair owns one outgoing arc at i, j, i', j'; arcs grouped by tail in pair order
    std::vector<int32_t> aptr(n + 1, 0);
    for (int64_t k = 0; k < np; ++k) { ++aptr[pi[k] + 1]; ++aptr[pj[k] + 1]; ++aptr[pi[k] + N + 1]; ++aptr[pj[k] + N + 1]; }
    for (int64_t v = 0; v < n; ++v) aptr[v + 1] += aptr[v];
    std::vector<int32_t> fill(aptr.begin(), aptr.end() - 1), slots(4 * np), slot_pair(m, 0);
    for (int64_t k = 0; k < np; ++k) {
      const int32_t i = pi[k], j = pj[k];
      slots[4 * k] = fill[i]++; slots[4 * k + 1] = fill[j]++;
      slots[4 * k + 2] = fill[i + N]++; slots[4 * k + 3] = fill[j + N]++;
      slot_pair[slots[4 * k]] = (int32_t)(2 * k); slot_pair[slots[4 * k + 1]] = (int32_t)(2 * k + 1);
      slot_pair[slots[4 * k + 2]] = (int32_t)(2 * k); slot_pair[slots[4 * k + 3]] = (int32_t)(2 * k + 1);
    }
    P->aptr = aptr;
    P->d_pair_i.upload(pi.data(), np); P->d_pair_j.upload(pj.data(), np);
    P->d_pe_ptr.upload(pe_ptr.data(), pe_ptr.size()); P->d_pe_edge.upload(pe_edge.data(), E);
    P->d_slots.upload(slots.data(), slots.size()); P->d_slot_pair.upload(slot_pair.data(), slot_pair.size());
    P->d_conn.upload(conn, 2 * E);
    P->d_ci.alloc(np); P->d_cjs.alloc(np); P->d_konst.alloc(np); P->d_trv.alloc(N);
    P->d_terms.alloc(std::max<int64_t>(N + E, n)); P->d_label.alloc(N); P->d_snk0.alloc(n);
    QpboSolver &S = P->S;
    S.P.N = N; S.P.aptr = aptr;
    S.n = (int)n; S.m = (int)m;
    S.d_aptr.upload(aptr.data(), aptr.size());
    S.max_degree = 0;
    for (size_t v = 0; v + 1 < aptr.size(); ++v) S.max_degree = std::max(S.max_degree, (int)(aptr[v + 1] - aptr[v]));
    S.d_head.alloc(m); S.d_rev.alloc(m); S.d_r.alloc(m); S.d_delta.alloc((size_t)2 * std::max<int64_t>(m, 1));
    S.d_ex.alloc(n); S.d_snk.alloc(n); S.d_h.alloc(n); S.d_h2.alloc(n); S.d_cnt.alloc(2048);
    S.g.n = (int)n; S.g.m = (int)m; S.g.aptr = S.d_aptr.p; S.g.head = S.d_head.p; S.g.rev = S.d_rev.p;
    S.g.r = S.d_r.p; S.g.delta = S.d_delta.p; S.g.ex = S.d_ex.p; S.g.snk = S.d_snk.p; S.g.h = S.d_h.p;
    S.g.h2 = S.d_h2.p; S.g.counters = S.d_cnt.p;
    S.set_tiling(N, 0, 0);
    STEREO_HIP_CHECK(hipDeviceSynchronize());
    *plan = P.release();
    return 0;
  } catch (const HipError &e) {
    return fail(e.msg, err, errcap);
  } catch (const std::exception &e) {
    return fail(std::string("stereo_rd_plan_create: ") + e.what(), err, errcap);
  }
}

extern "C" void stereo_rd_plan_destroy(stereo_rd_plan *plan) { delete plan; }

// inputs: six DEVICE arrays (U0, U1 of length N; E00, E01, E10, E11 of length E)
static int rd_plan_solve_device(stereo_rd_plan *P, const double *const in[6], int improve, double *labelling,
                                double *energy, double *lower_bound, double *num_unlabelled, char *err,
                                size_t errcap) {
  try {
    QpboSolver &S = P->S;
    const int64_t N = P->N, E = P->E, np = P->npairs;
    const int n = S.n;
    RdPlanDev d{};
    d.N = (int)N; d.npairs = (int)np; d.pair_i = P->d_pair_i.p; d.pair_j = P->d_pair_j.p;
    d.pe_ptr = P->d_pe_ptr.p; d.pe_edge = P->d_pe_edge.p; d.slots = P->d_slots.p; d.aptr = S.d_aptr.p;
    d.slot_pair = P->d_slot_pair.p;
    d.U0 = in[0]; d.U1 = in[1]; d.E00 = in[2]; d.E01 = in[3]; d.E10 = in[4]; d.E11 = in[5];
    d.head = S.d_head.p; d.rev = S.d_rev.p; d.r = S.d_r.p; d.ci = P->d_ci.p; d.cjs = P->d_cjs.p;
    d.konst = P->d_konst.p; d.ex = S.d_ex.p; d.snk = S.d_snk.p; d.trv = P->d_trv.p;
    // the two kernels may leave the sweep state of a previous move behind: restore the buffers
    S.g.h = S.d_h.p; S.g.h2 = S.d_h2.p;
    STEREO_HIP_CHECK(hipMemsetAsync(S.d_delta.p, 0, sizeof(double) * 2 * std::max(S.m, 1), 0));
    if (np > 0) hipLaunchKernelGGL(rd_pairs_kernel, dim3((unsigned)((np + kQB - 1) / kQB)), dim3(kQB), 0, 0, d);
    hipLaunchKernelGGL(rd_nodes_kernel, dim3((unsigned)((N + kQB - 1) / kQB)), dim3(kQB), 0, 0, d);
    STEREO_HIP_CHECK(hipMemcpyAsync(P->d_snk0.p, S.d_snk.p, sizeof(double) * n, hipMemcpyDeviceToDevice, 0));
    // constant of the normal form and sum_i min(0, tr_i): fixed-shape reductions, queued now and read back
    // together with the sums that follow the max-flow
    DetSums sums(P->d_partial, DetSums::blocks(np) + 2 * DetSums::blocks(N) + 2 * DetSums::blocks(n) + DetSums::blocks(N + E));
    const int s_konst = sums.queue(P->d_konst.p, np), s_u0 = sums.queue(in[0], N);
    const int s_neg = sums.queue(P->d_snk0.p, N);   // sum_i min(0, tr_i) = -(sink capacity of the unprimed half)
    S.iterations = 0; S.relabels = 0;
    const bool verbose = std::getenv("STEREO_HIP_QPBO_VERBOSE") != nullptr;
    auto now = [] { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
    if (verbose) STEREO_HIP_CHECK(hipDeviceSynchronize());
    const double tm0 = now();
    S.maxflow();
    if (verbose) {
      STEREO_HIP_CHECK(hipDeviceSynchronize());
      std::fprintf(stderr, "[stereo_hip qpbo plan] maxflow %.3f ms\n", now() - tm0);
    }
    // what reached the sink, node by node (snk0 - snk is exact for a node whose capacity barely moved -- an
    // out-of-range plane makes unaries of 4e7, dispmap_ncc.m:245 -- where the difference of the two totals
    // loses everything below ulp(1e12): 1e-3 on example_ncc.m's first move)
    hipLaunchKernelGGL(rd_absorbed_kernel, dim3((unsigned)((n + kQB - 1) / kQB)), dim3(kQB), 0, 0, n, P->d_snk0.p, S.d_snk.p,
                       P->d_terms.p);
    const int s_flow = sums.queue(P->d_terms.p, n);   // (d_terms, max(N + E, 2 N) doubles, is free until the energy terms below)
    // labels on the device; the host only sees the number of unlabelled nodes
    if ((int64_t)P->d_label.n < N) P->d_label.alloc(N);
    STEREO_HIP_CHECK(hipMemsetAsync(S.d_cnt.p + 3, 0, sizeof(int32_t), 0));
    if ((int64_t)P->d_free.n < N) P->d_free.alloc(N);
    hipLaunchKernelGGL(rd_labels_kernel, dim3((unsigned)((N + kQB - 1) / kQB)), dim3(kQB), 0, 0, N, n, S.g.h,
                       P->d_label.p, S.d_cnt.p + 3, P->d_free.p);
    // the energy of the strong labels, on the bet that no node stays unlabelled (otherwise it is summed again below)
    hipLaunchKernelGGL(rd_energy_terms_kernel, dim3((unsigned)((N + E + kQB - 1) / kQB)), dim3(kQB), 0, 0, N, E,
                       P->d_conn.p, S.g.h, n, in[0], in[1], in[2], in[3], in[4], in[5], P->d_label.p, P->d_terms.p);
    const int s_energy = sums.queue(P->d_terms.p, N + E);
    sums.fetch();
    int32_t unl32 = 0;
    STEREO_HIP_CHECK(hipMemcpy(&unl32, S.d_cnt.p + 3, sizeof(unl32), hipMemcpyDeviceToHost));
    const double konst = sums.get(s_konst) + sums.get(s_u0), neg = -sums.get(s_neg);
    *lower_bound = konst + neg + sums.get(s_flow) / 2;
    double unl = unl32;
    if (unl > 0) {
      const double tw0 = now();
      std::vector<int32_t> h;
      const int maxdeg = S.max_degree;
      const bool compact = maxdeg >= 1 && maxdeg <= 16 && !std::getenv("STEREO_HIP_QPBO_WEAK_FULL");
      if (compact) {
        // Weak persistency (QPBO_postprocessing.cpp:10-120) is a two-pass DFS over the FREE nodes only: their ids
        // come back, the arcs of exactly those nodes are gathered on the device, the pass runs on the host on that
        // small graph, and the new labels go back by id -- instead of the heights and the whole residual network.
        const int cnt = unl32;
        std::vector<int32_t> ids(cnt);
        STEREO_HIP_CHECK(hipMemcpy(ids.data(), P->d_free.p, sizeof(int32_t) * cnt, hipMemcpyDeviceToHost));
        std::sort(ids.begin(), ids.end());
        const size_t na = (size_t)2 * cnt * maxdeg;
        if (P->d_fids.n < (size_t)cnt) P->d_fids.alloc((size_t)cnt + cnt / 2 + 64);
        if (P->d_fhead.n < na) { P->d_fhead.alloc(na + na / 2 + 64); P->d_fflag.alloc(na + na / 2 + 64); }
        if (P->d_flab.n < (size_t)cnt) P->d_flab.alloc((size_t)cnt + cnt / 2 + 64);
        STEREO_HIP_CHECK(hipMemcpyAsync(P->d_fids.p, ids.data(), sizeof(int32_t) * cnt, hipMemcpyHostToDevice, 0));
        hipLaunchKernelGGL(rd_free_arcs_kernel, dim3((unsigned)((2 * cnt + kQB - 1) / kQB)), dim3(kQB), 0, 0, cnt, (int)N,
                           P->d_fids.p, S.d_aptr.p, S.d_head.p, S.d_rev.p, S.d_r.p, maxdeg, P->d_fhead.p, P->d_fflag.p);
        std::vector<int32_t> heads(na);
        std::vector<uint8_t> flags(na);
        STEREO_HIP_CHECK(hipMemcpy(heads.data(), P->d_fhead.p, sizeof(int32_t) * na, hipMemcpyDeviceToHost));
        STEREO_HIP_CHECK(hipMemcpy(flags.data(), P->d_fflag.p, na, hipMemcpyDeviceToHost));
        const double tw1 = now();
        std::vector<int8_t> lab;
        weak_persistencies_compact(cnt, (int)N, ids, maxdeg, heads, flags, lab);
        unl = 0;
        for (int t = 0; t < cnt; ++t) if (lab[t] < 0) unl += 1;
        STEREO_HIP_CHECK(hipMemcpyAsync(P->d_flab.p, lab.data(), cnt, hipMemcpyHostToDevice, 0));
        hipLaunchKernelGGL(rd_set_labels_kernel, dim3((unsigned)((cnt + kQB - 1) / kQB)), dim3(kQB), 0, 0, cnt, P->d_fids.p,
                           P->d_flab.p, P->d_label.p);
        STEREO_HIP_CHECK(hipStreamSynchronize(0));   // (the host vectors of the two asynchronous copies go out of scope)
        if (verbose) std::fprintf(stderr, "[stereo_hip qpbo plan] weak persistency: %d free variables, %.3f ms to fetch their arcs, %.3f ms on the host\n", cnt, tw1 - tw0, now() - tw1);
      } else {
        // (graphs with long arc lists: heights and the whole residual network go to the host)
        h.resize(n);
        STEREO_HIP_CHECK(hipMemcpy(h.data(), S.g.h, sizeof(int32_t) * n, hipMemcpyDeviceToHost));
        std::vector<int> label(N);
        for (int64_t i = 0; i < N; ++i) {
          const int li = h[i] < n ? 1 : 0, lm = h[i + N] < n ? 1 : 0;
          label[i] = li == lm ? -1 : li;
        }
        S.P.head.resize(S.m); S.P.rev.resize(S.m);
        std::vector<double> r(S.m);
        STEREO_HIP_CHECK(hipMemcpy(S.P.head.data(), S.d_head.p, sizeof(int32_t) * S.m, hipMemcpyDeviceToHost));
        STEREO_HIP_CHECK(hipMemcpy(S.P.rev.data(), S.d_rev.p, sizeof(int32_t) * S.m, hipMemcpyDeviceToHost));
        STEREO_HIP_CHECK(hipMemcpy(r.data(), S.d_r.p, sizeof(double) * S.m, hipMemcpyDeviceToHost));
        const double tw1 = now();
        weak_persistencies(S.P, r, label);
        unl = 0;
        for (int64_t i = 0; i < N; ++i) if (label[i] < 0) unl += 1;
        std::vector<int8_t> l8(N);
        for (int64_t i = 0; i < N; ++i) l8[i] = (int8_t)label[i];
        P->d_label.upload(l8.data(), N);
        if (verbose) std::fprintf(stderr, "[stereo_hip qpbo plan] weak persistency: %.3f ms to fetch the residual network, %.3f ms on the host\n", tw1 - tw0, now() - tw1);
      }
      *num_unlabelled = unl;  // rd_mex.cpp:83-88: counted before Improve
      if (improve && unl > 0) {
        const double ti0 = now();
        const std::vector<int32_t> perm = improve_permutation(N);
        if (verbose) std::fprintf(stderr, "[stereo_hip qpbo plan] Improve: permutation %.3f ms\n", now() - ti0);
        S.improve(perm, h);   // (returns the final heights)
        std::vector<int8_t> l8(N);
        for (int64_t i = 0; i < N; ++i) {
          const int li = h[i] < n ? 1 : 0, lm = h[i + N] < n ? 1 : 0;
          l8[i] = (int8_t)(li == lm ? 0 : li);   // QPBO_extra.cpp:1210-1219: ambiguous -> user label (0)
        }
        P->d_label.upload(l8.data(), N);
        if (verbose) std::fprintf(stderr, "[stereo_hip qpbo plan] Improve: %.3f ms\n", now() - ti0);
      }
    }
    *num_unlabelled = unl;
    if (std::getenv("STEREO_HIP_QPBO_VERBOSE"))
      std::fprintf(stderr, "[stereo_hip qpbo plan] n=%d arcs=%d iterations=%lld global_relabels=%lld unlabelled=%g\n", S.n,
                   S.m, (long long)S.iterations, (long long)S.relabels, unl);
    if (labelling) {
      std::vector<int8_t> l8(N);
      STEREO_HIP_CHECK(hipMemcpy(l8.data(), P->d_label.p, N, hipMemcpyDeviceToHost));
      for (int64_t i = 0; i < N; ++i) labelling[i] = l8[i];
    }
    if (unl32 > 0) {   // labels changed on the host (weak persistency, Improve): the energy of the final labelling
      hipLaunchKernelGGL(rd_energy_terms_kernel, dim3((unsigned)((N + E + kQB - 1) / kQB)), dim3(kQB), 0, 0, N, E,
                         P->d_conn.p, S.g.h, n, in[0], in[1], in[2], in[3], in[4], in[5], P->d_label.p, P->d_terms.p);
      *energy = det_sum(P->d_terms.p, N + E, P->d_partial);
    } else {
      *energy = sums.get(s_energy);
    }
    STEREO_HIP_CHECK(hipGetLastError());
    return 0;
  } catch (const HipError &e) {
    return fail(e.msg, err, errcap);
  } catch (const std::exception &e) {
    return fail(std::string("stereo_rd: ") + e.what(), err, errcap);
  }
}

extern "C" const int8_t *stereo_rd_plan_device_labels(stereo_rd_plan *P) { return P ? P->d_label.p : nullptr; }

extern "C" int stereo_rd_plan_set_grid(stereo_rd_plan *P, int H, int W, char *err, size_t errcap) {
  if (!P) return fail("stereo_rd_plan_set_grid: NULL plan", err, errcap);
  if (H < 1 || W < 1 || (int64_t)H * W != P->N) return fail("stereo_rd_plan_set_grid: H * W must equal N", err, errcap);
  try {
    P->S.set_tiling(P->N, H, W);
    STEREO_HIP_CHECK(hipDeviceSynchronize());
    return 0;
  } catch (const HipError &e) {
    return fail(e.msg, err, errcap);
  }
}

extern "C" int stereo_rd_plan_solve(stereo_rd_plan *P, const double *U0, const double *U1, const double *E00,
                                    const double *E01, const double *E10, const double *E11, int improve,
                                    double *labelling, double *energy, double *lower_bound,
                                    double *num_unlabelled, char *err, size_t errcap) {
  if (!P || !U0 || !U1 || !labelling || !energy || !lower_bound || !num_unlabelled ||
      (P->E > 0 && (!E00 || !E01 || !E10 || !E11)))
    return fail("stereo_rd_plan_solve: NULL argument", err, errcap);
  try {
    const double *src[6] = {U0, U1, E00, E01, E10, E11};
    const double *in[6];
    for (int k = 0; k < 6; ++k) {
      P->d_in[k].upload(src[k], k < 2 ? P->N : P->E);
      in[k] = P->d_in[k].p;
    }
    return rd_plan_solve_device(P, in, improve, labelling, energy, lower_bound, num_unlabelled, err, errcap);
  } catch (const HipError &e) {
    return fail(e.msg, err, errcap);
  }
}

extern "C" int stereo_rd_plan_solve_device(stereo_rd_plan *P, const double *d_U0, const double *d_U1,
                                           const double *d_E00, const double *d_E01, const double *d_E10,
                                           const double *d_E11, int improve, double *labelling, double *energy,
                                           double *lower_bound, double *num_unlabelled, char *err,
                                           size_t errcap) {
  if (!P || !d_U0 || !d_U1 || !energy || !lower_bound || !num_unlabelled)
    return fail("stereo_rd_plan_solve_device: NULL argument", err, errcap);
  const double *in[6] = {d_U0, d_U1, d_E00, d_E01, d_E10, d_E11};
  return rd_plan_solve_device(P, in, improve, labelling, energy, lower_bound, num_unlabelled, err, errcap);
}
