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
#include <cmath>
#include <cstdint>
#include <cstdlib>
#include <memory>
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
  int32_t *counters;  // [0] active nodes, [1] frontier size (next), [2] changed flag
};

__global__ __launch_bounds__(kQB) void qpbo_push_kernel(QpboDev g) {
  const int v = blockIdx.x * kQB + threadIdx.x;
  if (v >= g.n) return;
  double e = g.ex[v];
  const int hv = g.h[v];
  if (!(e > 0) || hv >= g.n) return;
  if (hv == 1) {
    const double s = g.snk[v];
    if (s > 0) {
      const double d = e < s ? e : s;
      g.snk[v] = s - d;
      e -= d;
    }
  }
  const int a0 = g.aptr[v], a1 = g.aptr[v + 1];
  for (int a = a0; a < a1 && e > 0; ++a) {
    const double ra = g.r[a];
    if (ra > 0 && hv == g.h[g.head[a]] + 1) {
      const double d = e < ra ? e : ra;
      g.r[a] = ra - d;
      g.r[g.rev[a]] += d;
      g.delta[a] = d;
      e -= d;
    }
  }
  g.ex[v] = e;
}

// arcs are stored grouped by tail node, so "arc a of node v" is simply index a
__global__ __launch_bounds__(kQB) void qpbo_gather_relabel_kernel(QpboDev g) {
  const int v = blockIdx.x * kQB + threadIdx.x;
  if (v >= g.n) return;
  double e = g.ex[v];
  const int a0 = g.aptr[v], a1 = g.aptr[v + 1];
  for (int a = a0; a < a1; ++a) {
    const int b = g.rev[a];
    const double d = g.delta[b];
    if (d != 0) { e += d; g.delta[b] = 0; }
  }
  g.ex[v] = e;
  int hv = g.h[v];
  if (e > 0 && hv < g.n) {
    int hmin = g.n;
    if (g.snk[v] > 0) hmin = 0;
    for (int a = a0; a < a1; ++a)
      if (g.r[a] > 0) {
        const int hw = g.h[g.head[a]];
        hmin = hw < hmin ? hw : hmin;
      }
    if (hmin + 1 > hv) hv = hmin + 1 < g.n ? hmin + 1 : g.n;
    if (hv < g.n) atomicAdd(g.counters, 1);
  }
  g.h2[v] = hv;
}

__global__ __launch_bounds__(kQB) void qpbo_bfs_init_kernel(QpboDev g, int32_t *frontier) {
  const int v = blockIdx.x * kQB + threadIdx.x;
  if (v >= g.n) return;
  if (g.snk[v] > 0) {
    g.h[v] = 1;
    frontier[atomicAdd(g.counters + 1, 1)] = v;
  } else {
    g.h[v] = g.n;
  }
}

__global__ __launch_bounds__(kQB) void qpbo_bfs_step_kernel(QpboDev g, const int32_t *frontier, int count,
                                                           int32_t *next, int level) {
  const int t = blockIdx.x * kQB + threadIdx.x;
  if (t >= count) return;
  const int w = frontier[t];
  const int a0 = g.aptr[w], a1 = g.aptr[w + 1];
  for (int a = a0; a < a1; ++a) {
    // arc head[a] -> w is rev[a]: it must be residual for head[a] to reach the sink through w
    if (g.r[g.rev[a]] > 0) {
      const int v = g.head[a];
      if (atomicCAS(g.h + v, g.n, level + 1) == g.n) next[atomicAdd(g.counters + 1, 1)] = v;
    }
  }
}

__global__ __launch_bounds__(kQB) void qpbo_count_active_kernel(QpboDev g) {
  const int v = blockIdx.x * kQB + threadIdx.x;
  if (v >= g.n) return;
  if (g.ex[v] > 0 && g.h[v] < g.n) atomicAdd(g.counters, 1);
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
  DevBuf<int32_t> d_aptr, d_head, d_rev, d_h, d_h2, d_front0, d_front1, d_cnt;
  DevBuf<double> d_r, d_delta, d_ex, d_snk;
  std::vector<double> snk0;
  QpboDev g{};
  int64_t iterations = 0, relabels = 0;

  void upload() {
    n = (int)(2 * P.N); m = (int)P.head.size();
    d_aptr.upload(P.aptr.data(), P.aptr.size());
    d_head.upload(P.head.data(), P.head.size());
    d_rev.upload(P.rev.data(), P.rev.size());
    d_r.upload(P.cap.data(), P.cap.size());
    d_delta.alloc(std::max(m, 1));
    STEREO_HIP_CHECK(hipMemset(d_delta.p, 0, sizeof(double) * std::max(m, 1)));
    std::vector<double> ex(n), snk(n);
    for (int v = 0; v < n; ++v) {  // QPBO_maxflow.cpp:135-150: tr_cap > 0 source arc, < 0 sink arc
      ex[v] = P.tr[v] > 0 ? P.tr[v] : 0.0;
      snk[v] = P.tr[v] < 0 ? -P.tr[v] : 0.0;
    }
    snk0 = snk;
    d_ex.upload(ex.data(), n); d_snk.upload(snk.data(), n);
    d_h.alloc(n); d_h2.alloc(n); d_front0.alloc(n); d_front1.alloc(n); d_cnt.alloc(4);
    STEREO_HIP_CHECK(hipMemset(d_cnt.p, 0, sizeof(int32_t) * 4));
    g.n = n; g.m = m; g.aptr = d_aptr.p; g.head = d_head.p; g.rev = d_rev.p; g.r = d_r.p;
    g.delta = d_delta.p; g.ex = d_ex.p; g.snk = d_snk.p; g.h = d_h.p; g.h2 = d_h2.p; g.counters = d_cnt.p;
    STEREO_HIP_CHECK(hipDeviceSynchronize());
  }

  int grid() const { return (n + kQB - 1) / kQB; }

  // exact distances to the sink in the residual graph (frontier BFS); returns #active nodes
  int global_relabel() {
    int32_t cnt[4] = {0, 0, 0, 0};
    STEREO_HIP_CHECK(hipMemset(d_cnt.p, 0, sizeof(cnt)));
    hipLaunchKernelGGL(qpbo_bfs_init_kernel, dim3(grid()), dim3(kQB), 0, 0, g, d_front0.p);
    int32_t *cur = d_front0.p, *nxt = d_front1.p;
    for (int level = 1;; ++level) {
      STEREO_HIP_CHECK(hipMemcpy(cnt, d_cnt.p, sizeof(cnt), hipMemcpyDeviceToHost));
      const int count = cnt[1];
      if (count == 0) break;
      STEREO_HIP_CHECK(hipMemset(d_cnt.p + 1, 0, sizeof(int32_t)));
      hipLaunchKernelGGL(qpbo_bfs_step_kernel, dim3((count + kQB - 1) / kQB), dim3(kQB), 0, 0, g, cur, count, nxt, level);
      std::swap(cur, nxt);
    }
    STEREO_HIP_CHECK(hipMemset(d_cnt.p, 0, sizeof(int32_t)));
    hipLaunchKernelGGL(qpbo_count_active_kernel, dim3(grid()), dim3(kQB), 0, 0, g);
    STEREO_HIP_CHECK(hipMemcpy(cnt, d_cnt.p, sizeof(cnt), hipMemcpyDeviceToHost));
    ++relabels;
    return cnt[0];
  }

  // AddUnaryTerm(i, 0, INFTY) with INFTY = 1 + max over the two saturation sums of node i
  // (QPBO_extra.cpp:241-254, :1185-1199), applied to the push-relabel state: more source
  // capacity at i, more sink capacity at its mate.
  void fix_to_zero(int i) {
    const int im = i + (int)P.N;
    double ex2[2], sn2[2];
    STEREO_HIP_CHECK(hipMemcpy(&ex2[0], d_ex.p + i, sizeof(double), hipMemcpyDeviceToHost));
    STEREO_HIP_CHECK(hipMemcpy(&ex2[1], d_ex.p + im, sizeof(double), hipMemcpyDeviceToHost));
    STEREO_HIP_CHECK(hipMemcpy(&sn2[0], d_snk.p + i, sizeof(double), hipMemcpyDeviceToHost));
    STEREO_HIP_CHECK(hipMemcpy(&sn2[1], d_snk.p + im, sizeof(double), hipMemcpyDeviceToHost));
    const int a0 = P.aptr[i], a1 = P.aptr[i + 1];
    std::vector<double> rr(std::max(a1 - a0, 1));
    const double tcap = ex2[0] - sn2[0];
    double c1 = -tcap, c2 = tcap;
    for (int a = a0; a < a1; ++a) {
      double ra, rb;
      STEREO_HIP_CHECK(hipMemcpy(&ra, d_r.p + a, sizeof(double), hipMemcpyDeviceToHost));
      STEREO_HIP_CHECK(hipMemcpy(&rb, d_r.p + P.rev[a], sizeof(double), hipMemcpyDeviceToHost));
      c1 += ra; c2 += rb;
    }
    const double INFTY = (c1 > c2 ? c1 : c2) + 1;
    double t0 = ex2[0] - sn2[0] + INFTY, t1 = ex2[1] - sn2[1] - INFTY;
    const double nex0 = t0 > 0 ? t0 : 0, nsn0 = t0 < 0 ? -t0 : 0;
    const double nex1 = t1 > 0 ? t1 : 0, nsn1 = t1 < 0 ? -t1 : 0;
    STEREO_HIP_CHECK(hipMemcpy(d_ex.p + i, &nex0, sizeof(double), hipMemcpyHostToDevice));
    STEREO_HIP_CHECK(hipMemcpy(d_snk.p + i, &nsn0, sizeof(double), hipMemcpyHostToDevice));
    STEREO_HIP_CHECK(hipMemcpy(d_ex.p + im, &nex1, sizeof(double), hipMemcpyHostToDevice));
    STEREO_HIP_CHECK(hipMemcpy(d_snk.p + im, &nsn1, sizeof(double), hipMemcpyHostToDevice));
  }

  void maxflow() {
    int active = global_relabel();
    const int check_every = 32;
    int relabel_every = 256;
    if (const char *e = std::getenv("STEREO_HIP_QPBO_RELABEL_EVERY")) relabel_every = std::max(1, std::atoi(e));
    int since_relabel = 0;
    while (active > 0) {
      for (int k = 0; k < check_every; ++k) {
        hipLaunchKernelGGL(qpbo_push_kernel, dim3(grid()), dim3(kQB), 0, 0, g);
        if (k == check_every - 1) STEREO_HIP_CHECK(hipMemsetAsync(d_cnt.p, 0, sizeof(int32_t), 0));
        hipLaunchKernelGGL(qpbo_gather_relabel_kernel, dim3(grid()), dim3(kQB), 0, 0, g);
        std::swap(g.h, g.h2);
      }
      iterations += check_every;
      since_relabel += check_every;
      int32_t cnt = 0;
      STEREO_HIP_CHECK(hipMemcpy(&cnt, d_cnt.p, sizeof(cnt), hipMemcpyDeviceToHost));
      active = cnt;
      if (active > 0 && since_relabel >= relabel_every) {
        active = global_relabel();
        since_relabel = 0;
      }
    }
    // heights may be stale lower bounds: one exact BFS defines T
    global_relabel();
    STEREO_HIP_CHECK(hipGetLastError());
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

}  // namespace

extern "C" int stereo_rd(const double *U0, const double *U1, const double *E00, const double *E01,
                         const double *E10, const double *E11, const uint32_t *conn, int64_t N, int64_t E,
                         int improve, double *labelling, double *energy, double *lower_bound,
                         double *num_unlabelled, char *err, size_t errcap) {
  if (!U0 || !U1 || !labelling || !energy || !lower_bound || !num_unlabelled || (E > 0 && (!E00 || !E01 || !E10 || !E11 || !conn)))
    return fail("stereo_rd: NULL argument", err, errcap);
  if (stereo_hip_device_count() < 1)
    return fail("stereo_rd: no HIP device available (the HIP path has no CPU fallback)", err, errcap);
  try {
    QpboSolver S;
    std::string berr;
    if (!build_problem(U0, U1, E00, E01, E10, E11, conn, N, E, S.P, berr)) return fail(berr, err, errcap);
    S.upload();
    S.maxflow();
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
      std::vector<int32_t> perm(N);
      for (int64_t i = 0; i < N; ++i) perm[i] = (int32_t)i;
      for (int64_t i = 0; i < N - 1; ++i) {
        int64_t j = i + (int64_t)((rand() / (1.0 + (double)RAND_MAX)) * (double)(N - i));
        if (j > N - 1) j = N - 1;
        std::swap(perm[i], perm[j]);
      }
      // only nodes without a strong label can ever need fixing; strong labels persist
      for (int64_t pidx = 0; pidx < N; ++pidx) {
        const int32_t i = perm[pidx];
        if (h[i] < n || h[i + N] < n) {
          if ((h[i] < n) != (h[i + N] < n)) continue;  // strongly labelled
        }
        S.fix_to_zero(i);
        S.maxflow();
        STEREO_HIP_CHECK(hipMemcpy(h.data(), S.g.h, sizeof(int32_t) * n, hipMemcpyDeviceToHost));
      }
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
