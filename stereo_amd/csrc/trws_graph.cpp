// Host-side graph analysis for the TRW-S path.  See trws_graph.h.
#include "trws_graph.h"

#include "../../include/stereo_hip.h"
#include "common.h"

#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <numeric>
#include <queue>
#include <thread>

namespace stereo {
namespace {

// Boundary set of the ordering heuristic (ordering.cpp:70-152).  The reference
// keeps a LIFO-prepended doubly linked list and scans it for the FIRST node of
// minimum remaining degree, which is O(|boundary|) per pick.  The same pick is
// "minimum degree, ties broken by most recent insertion": one max-heap of
// insertion stamps per degree value, with lazy invalidation.
class Boundary {
 public:
  explicit Boundary(int max_deg) : buckets_(max_deg + 1), cur_(max_deg + 1) {}
  void push(int deg, int64_t stamp, int32_t node) {
    buckets_[deg].emplace(stamp, node);
    if (deg < cur_) cur_ = deg;
    ++live_;
  }
  // entry is current iff the node is still in the boundary with this degree/stamp
  template <class Valid>
  bool pop(Valid valid, int32_t &node) {
    const int nb = (int)buckets_.size();
    while (cur_ < nb) {
      auto &h = buckets_[cur_];
      while (!h.empty()) {
        auto top = h.top();
        h.pop();
        if (valid(top.second, cur_, top.first)) {
          node = top.second;
          return true;
        }
      }
      ++cur_;
    }
    return false;
  }

 private:
  std::vector<std::priority_queue<std::pair<int64_t, int32_t>>> buckets_;
  int cur_;
  int64_t live_ = 0;
};

}  // namespace

int spec_segment_length() {
  int L = 16;
  if (const char *e = std::getenv("STEREO_HIP_TRWS_SPEC_SEG")) L = std::atoi(e);
  return std::max(4, std::min(L, 48));
}

#define TICK(name) do { if (std::getenv("STEREO_HIP_GRAPH_VERBOSE")) { auto now_ = std::chrono::steady_clock::now(); std::fprintf(stderr, "[graph] -> %s: %.1f ms\n", name, std::chrono::duration<double, std::milli>(now_ - tick_).count()); tick_ = now_; } } while (0)
bool build_trws_graph(int64_t N, int64_t E, const uint32_t *conn, TrwsGraph &g,
                      std::string &err, int64_t max_resident_runs, const int32_t *owner_in, int nstrips,
                      int64_t certainly_resident, int ordering) {
  auto tick_ = std::chrono::steady_clock::now();
  if (N <= 0 || E < 0) { err = "build_trws_graph: empty problem"; return false; }
  if (N >= INT32_MAX || E >= INT32_MAX) { err = "build_trws_graph: more than 2^31 nodes/edges"; return false; }
  g = TrwsGraph();
  g.N = N; g.E = E;
  if (nstrips < 1) { err = "build_trws_graph: nstrips must be >= 1"; return false; }
  if (nstrips > 1 && !owner_in) { err = "build_trws_graph: strips need an owner per node"; return false; }
  g.nstrips = nstrips;
  if (nstrips > 1) {
    g.owner.assign(owner_in, owner_in + N);
    for (int64_t i = 0; i < N; ++i)
      if (g.owner[i] < 0 || g.owner[i] >= nstrips) { err = "build_trws_graph: owner out of range"; return false; }
  }
  const int32_t *own = nstrips > 1 ? g.owner.data() : nullptr;  // per node
  g.tail.resize(E); g.head.resize(E); g.mdir.assign(E, 0);
  std::vector<int32_t> firstF(N, -1), firstB(N, -1), nextF(E), nextB(E);
  std::vector<int32_t> deg(N, 0);
  // AddEdge: prepend to the tail's forward and the head's backward list.
  for (int64_t e = 0; e < E; ++e) {
    uint32_t a = conn[2 * e], b = conn[2 * e + 1];
    if (a >= (uint64_t)N || b >= (uint64_t)N) { err = "connectivity index out of range"; return false; }
    if (a == b) { err = "self loops are not supported"; return false; }
    if (own && std::abs(own[a] - own[b]) > 1) { err = "strips must form a chain: an edge joins strips that are not neighbours"; return false; }
    g.tail[e] = (int32_t)a; g.head[e] = (int32_t)b;
    nextF[e] = firstF[a]; firstF[a] = (int32_t)e;
    nextB[e] = firstB[b]; firstB[b] = (int32_t)e;
    ++deg[a]; ++deg[b];
  }
  TICK("0");
  // ---- SetAutomaticOrdering (or, as a labelled option, the node index order)
  g.order.resize(N); g.rank.assign(N, -1);
  if (ordering == 1) {
    for (int64_t i = 0; i < N; ++i) { g.order[i] = (int32_t)i; g.rank[i] = (int32_t)i; }
  } else {
    const int max_deg = *std::max_element(deg.begin(), deg.end());
    std::vector<uint8_t> where(N, 2);  // 2 untouched list, 1 boundary, 0 ordered
    std::vector<int64_t> stamp(N, 0);
    // untouched nodes never change degree, so the outer "first node of minimum
    // degree in index order" is a cursor over nodes sorted by (degree, index)
    std::vector<int32_t> by_deg(N);
    std::iota(by_deg.begin(), by_deg.end(), 0);
    std::stable_sort(by_deg.begin(), by_deg.end(),
                     [&](int32_t x, int32_t y) { return deg[x] < deg[y]; });
    int64_t cursor = 0, counter = 0, count = 0;
    Boundary bnd(max_deg);
    auto valid = [&](int32_t n, int d, int64_t s) { return where[n] == 1 && deg[n] == d && stamp[n] == s; };
    while (count < N) {
      while (cursor < N && where[by_deg[cursor]] != 2) ++cursor;
      if (cursor >= N) { err = "ordering: internal error"; return false; }
      int32_t seed = by_deg[cursor];
      where[seed] = 1; stamp[seed] = ++counter;
      bnd.push(deg[seed], stamp[seed], seed);
      int32_t i;
      while (bnd.pop(valid, i)) {
        where[i] = 0; g.rank[i] = (int32_t)count; g.order[count++] = i;
        for (int pass = 0; pass < 2; ++pass) {
          for (int32_t e = pass == 0 ? firstF[i] : firstB[i]; e >= 0;
               e = pass == 0 ? nextF[e] : nextB[e]) {
            int32_t j = pass == 0 ? g.head[e] : g.tail[e];
            if (where[j] == 0) continue;
            --deg[j];
            if (where[j] == 2) { where[j] = 1; stamp[j] = ++counter; }
            bnd.push(deg[j], stamp[j], j);
          }
        }
      }
    }
  }
  TICK("1");
  // ---- CompleteGraphConstruction: orient low -> high rank, rebuild lists
  std::fill(firstB.begin(), firstB.end(), -1);
  for (int64_t r = 0; r < N; ++r) {
    int32_t i = g.order[r], eprev = -1;
    for (int32_t e = firstF[i]; e >= 0;) {
      int32_t j = g.head[e];
      if (g.rank[i] < g.rank[j]) {
        nextB[e] = firstB[j]; firstB[j] = e;
        eprev = e; e = nextF[e];
      } else {
        int32_t enext = nextF[e];
        g.mdir[e] ^= 1; g.tail[e] = j; g.head[e] = i;
        if (eprev >= 0) nextF[eprev] = enext; else firstF[i] = enext;
        nextF[e] = firstF[j]; firstF[j] = e;
        nextB[e] = firstB[i]; firstB[i] = e;
        e = enext;
      }
    }
  }
  TICK("2");
  // ---- flatten to CSR by rank, gamma, levels, lower-bound term positions
  g.fptr.assign(N + 1, 0); g.bptr.assign(N + 1, 0);
  g.fidx.resize(E); g.bidx.resize(E); g.gamma.resize(N);
  int64_t pf = 0, pb = 0;
  std::vector<int32_t> level(N, 0);
  int32_t nlev = 0;
  for (int64_t r = 0; r < N; ++r) {
    int32_t i = g.order[r];
    g.fptr[r] = (int32_t)pf; g.bptr[r] = (int32_t)pb;
    for (int32_t e = firstF[i]; e >= 0; e = nextF[e]) g.fidx[pf++] = e;
    int32_t lv = 0;
    for (int32_t e = firstB[i]; e >= 0; e = nextB[e]) {
      g.bidx[pb++] = e;
      lv = std::max(lv, level[g.rank[g.tail[e]]] + 1);
    }
    level[r] = lv; nlev = std::max(nlev, lv + 1);
    int nf = (int)(pf - g.fptr[r]), nbk = (int)(pb - g.bptr[r]);
    int ni = std::max(nf, nbk);
    g.gamma[r] = ni > 0 ? (double)1 / ni : 1.0;  // isolated node: no edge ever reads gamma
  }
  g.fptr[N] = (int32_t)pf; g.bptr[N] = (int32_t)pb;
  g.level_ptr.assign(nlev + 1, 0);
  for (int64_t r = 0; r < N; ++r) ++g.level_ptr[level[r] + 1];
  for (int32_t l = 0; l < nlev; ++l) {
    g.max_level_nodes = std::max<int64_t>(g.max_level_nodes, g.level_ptr[l + 1]);
    g.level_ptr[l + 1] += g.level_ptr[l];
  }
  g.level_ranks.resize(N);
  {
    std::vector<int32_t> fill(g.level_ptr.begin(), g.level_ptr.end() - 1);
    for (int64_t r = 0; r < N; ++r) g.level_ranks[fill[level[r]]++] = (int32_t)r;
  }
  g.lb_pos_node.resize(N); g.lb_pos_edge.assign(E, -1);
  g.strip_lb_terms.assign(nstrips, 0); g.strip_nodes.assign(nstrips, 0); g.e_pos.resize(N);
  for (int64_t r = N - 1; r >= 0; --r) {
    int64_t &pos = g.strip_lb_terms[own ? own[g.order[r]] : 0];  // the node that computes a term owns it
    g.lb_pos_node[r] = (int32_t)pos++;
    for (int32_t k = g.bptr[r]; k < g.bptr[r + 1]; ++k) g.lb_pos_edge[g.bidx[k]] = (int32_t)pos++;
  }
  for (int64_t r = 0; r < N; ++r) g.e_pos[r] = (int32_t)g.strip_nodes[own ? own[g.order[r]] : 0]++;
  g.lb_terms = 0;
  for (int64_t v : g.strip_lb_terms) g.lb_terms = std::max(g.lb_terms, v);
  if (nstrips == 1) g.lb_terms = g.strip_lb_terms[0];
  TICK("3");
  // ---- persistent sweep schedules
  // cut == false: a run ends only where the node does not hang on one of the two previous
  // visits.  cut == true (used when there are more runs than resident workgroups): a run also
  // ends in front of a node whose dependency level jumps (it will wait long for a foreign
  // node -- e.g. the last node of a grid row waits for the border chain -- and would pin a
  // workgroup meanwhile), and runs are dispensed by the level of their first node.
  auto build_runs = [&](int d, bool cut) {
    TrwsGraph::Sweep &S = g.sweep[d];
    // incoming / outgoing lists of this direction, by rank
    const std::vector<int32_t> &iptr = d == 0 ? g.bptr : g.fptr, &iidx = d == 0 ? g.bidx : g.fidx;
    const std::vector<int32_t> &optr = d == 0 ? g.fptr : g.bptr, &oidx = d == 0 ? g.fidx : g.bidx;
    S.run_ptr.clear(); S.dep_ptr.assign(N + 1, 0); S.dep_rank.clear(); S.in_slot.assign(E, -1);
    S.run_order.clear();
    std::vector<std::vector<int32_t>> tmp_deps;  // filled per rank in processing order
    tmp_deps.resize(N);
    std::vector<int32_t> lev(N, 0);  // dependency level within this sweep direction
    constexpr int32_t kJump = 8;
    int64_t run_start = 0;
    for (int64_t p = 0; p < N; ++p) {
      const int32_t r = d == 0 ? (int32_t)p : (int32_t)(N - 1 - p);
      // ranks visited one and two steps earlier (hand-over through LDS is kept for two visits)
      const int32_t near1 = d == 0 ? r - 1 : r + 1, near2 = d == 0 ? r - 2 : r + 2;
      bool chained = false;
      std::vector<int32_t> &deps = tmp_deps[r];
      // pass 1: does this node hang on one of the last two visits of the current run?
      int32_t lv = 0;
      for (int32_t k = iptr[r]; k < iptr[r + 1]; ++k) {
        const int32_t other = g.rank[d == 0 ? g.tail[iidx[k]] : g.head[iidx[k]]];
        if (((p - 1 >= run_start && other == near1) || (p - 2 >= run_start && other == near2)) &&
            (!own || own[g.order[other]] == own[g.order[r]])) chained = true;
        lv = std::max(lv, lev[other] + 1);
      }
      lev[r] = lv;
      if (cut && chained && p >= 1 && lv > lev[near1] + kJump) chained = false;
      if (!chained) { S.run_ptr.push_back((int32_t)p); run_start = p; }
      for (int32_t k = iptr[r]; k < iptr[r + 1]; ++k) {
        const int32_t e = iidx[k];
        const int32_t other = g.rank[d == 0 ? g.tail[e] : g.head[e]];
        int dist = 0;
        if (p - 1 >= run_start && other == near1) dist = 1;
        else if (p - 2 >= run_start && other == near2) dist = 2;
        if (own && own[g.order[other]] != own[g.order[r]]) dist = 0;
        if (dist) {
          // slot of the edge in that node's outgoing list; in_slot = slot + 8 * (dist - 1)
          for (int32_t w = optr[other]; w < optr[other + 1] && w - optr[other] < TrwsGraph::kMaxSlots; ++w)
            if (oidx[w] == e) S.in_slot[k] = (int8_t)((w - optr[other]) + 8 * (dist - 1));
        } else if (std::find(deps.begin(), deps.end(), other) == deps.end()) {
          deps.push_back(other);
        }
      }
    }
    S.run_ptr.push_back((int32_t)N);
    int64_t dp = 0;
    for (int64_t r = 0; r < N; ++r) {
      S.dep_ptr[r] = (int32_t)dp;
      for (int32_t x : tmp_deps[r]) { S.dep_rank.push_back(x); ++dp; }
    }
    S.dep_ptr[N] = (int32_t)dp;
    if (!cut) return;
    // dispense runs by the level of their first node; keep the order only if every foreign
    // dependency then lies in a run dispensed earlier (otherwise workgroups could all be
    // waiting for a run nobody has picked up yet)
    const int64_t R = (int64_t)S.run_ptr.size() - 1;
    std::vector<int32_t> order(R);
    for (int64_t k = 0; k < R; ++k) order[k] = (int32_t)k;
    auto first_lev = [&](int32_t k) { const int64_t p = S.run_ptr[k]; return lev[d == 0 ? p : N - 1 - p]; };
    std::stable_sort(order.begin(), order.end(), [&](int32_t x, int32_t y) { return first_lev(x) < first_lev(y); });
    std::vector<int32_t> ticket_of_run(R), run_of_pos(N);
    for (int64_t t = 0; t < R; ++t) ticket_of_run[order[t]] = (int32_t)t;
    for (int64_t k = 0; k < R; ++k)
      for (int64_t p = S.run_ptr[k]; p < S.run_ptr[k + 1]; ++p) run_of_pos[p] = (int32_t)k;
    bool safe = true;
    for (int64_t r = 0; r < N && safe; ++r) {
      const int32_t mine = run_of_pos[d == 0 ? r : N - 1 - r];
      for (int32_t x : tmp_deps[r]) {
        const int32_t theirs = run_of_pos[d == 0 ? x : N - 1 - x];
        if (theirs != mine && ticket_of_run[theirs] > ticket_of_run[mine]) { safe = false; break; }
      }
    }
    if (safe) S.run_order = order;
  };
  for (int d = 0; d < 2; ++d) {
    build_runs(d, false);
    if (max_resident_runs > 0 && (int64_t)g.sweep[d].run_ptr.size() - 1 > max_resident_runs) {
      build_runs(d, true);
      // a cut turns the hand-over from the previous visit into a foreign dependency; the fast
      // kernels take at most four per node
      bool ok = true;
      for (int64_t r = 0; r < N && ok; ++r) ok = g.sweep[d].dep_ptr[r + 1] - g.sweep[d].dep_ptr[r] <= 4;
      if (!ok) build_runs(d, false);
    }
  }
  for (int d = 0; d < 2 && own; ++d) {
    TrwsGraph::Sweep &S = g.sweep[d];
    S.run_strip.clear();
    for (size_t k = 0; k + 1 < S.run_ptr.size(); ++k) {
      const int64_t p = S.run_ptr[k];
      S.run_strip.push_back(own[g.order[d == 0 ? p : N - 1 - p]]);
    }
  }
  TICK("4");
  // ---- descriptors of the fast kernel (layout: trws.hip NodeDesc)
  g.fast_ok = true;
  for (int64_t r = 0; r < N && g.fast_ok; ++r) {
    if ((g.fptr[r + 1] - g.fptr[r]) + (g.bptr[r + 1] - g.bptr[r]) > 8) g.fast_ok = false;
    for (int d = 0; d < 2; ++d)
      if (g.sweep[d].dep_ptr[r + 1] - g.sweep[d].dep_ptr[r] > 4) g.fast_ok = false;
  }
  if (g.fast_ok) {
    constexpr int W = TrwsGraph::kDescWords;
    // the two sweep directions are independent of each other: one host thread each
    auto build_direction = [&](int d) {
      auto tick_ = std::chrono::steady_clock::now();
#define DTICK(name) do { if (d == 0) TICK(name); } while (0)
      TrwsGraph::Sweep &S = g.sweep[d];
      const std::vector<int32_t> &iptr = d == 0 ? g.bptr : g.fptr, &iidx = d == 0 ? g.bidx : g.fidx;
      const std::vector<int32_t> &optr = d == 0 ? g.fptr : g.bptr, &oidx = d == 0 ? g.fidx : g.bidx;
      auto position = [&](int32_t r) -> int64_t { return d == 0 ? (int64_t)r : N - 1 - (int64_t)r; };
      auto other_end = [&](int32_t e_in) -> int32_t { return g.rank[d == 0 ? g.tail[e_in] : g.head[e_in]]; };
      // ---- chain schedule: a node extends the run of the node visited two steps or one step
      // earlier if it depends on it and that node is still the last one of its run; two steps
      // first, which is what separates two interleaved rows (s0 s1 s2 s3 ...: s3 hangs on s1 AND
      // on s2, s4 only on s2) into the runs s0 s1 s3 s5 ... and s2 s4 s6 ...
      std::vector<int32_t> lev(N, 0), run_of(N, -1), pred(N, -1), next_of(N, -1), run_tail, run_head, first_lev;
      const bool cut = max_resident_runs > 0 && (int64_t)S.run_ptr.size() - 1 > max_resident_runs;
      constexpr int32_t kJump = 8;
      for (int64_t p = 0; p < N; ++p) {
        const int32_t r = d == 0 ? (int32_t)p : (int32_t)(N - 1 - p);
        int32_t lv = 0, best = -1;
        for (int32_t k = iptr[r]; k < iptr[r + 1]; ++k) {
          const int32_t o = other_end(iidx[k]);
          lv = std::max(lv, lev[o] + 1);
          const int64_t back = p - position(o);
          if ((back != 1 && back != 2) || run_tail[run_of[o]] != o) continue;
          if (own && own[g.order[o]] != own[g.order[r]]) continue;  // a run stays inside one strip
          if (best < 0 || position(o) < position(best)) best = o;
        }
        lev[r] = lv;
        if (cut && best >= 0 && lv > lev[best] + kJump) best = -1;
        if (best >= 0) {
          run_of[r] = run_of[best]; next_of[best] = r; run_tail[run_of[r]] = r; pred[r] = best;
        } else {
          run_of[r] = (int32_t)run_head.size(); run_head.push_back(r); run_tail.push_back(r); first_lev.push_back(lv);
        }
      }
      const int64_t R = (int64_t)run_head.size();
      DTICK("dir0 chain schedule");
      // foreign dependencies per rank (everything but the predecessor in the run)
      struct Deps {  // at most kMaxSlots incoming edges per node in this branch (fast_ok)
        int32_t v[TrwsGraph::kMaxSlots]; int32_t n = 0;
        const int32_t *begin() const { return v; }
        const int32_t *end() const { return v + n; }
        size_t size() const { return (size_t)n; }
        int32_t operator[](int k) const { return v[k]; }
        void push_back(int32_t x) { v[n++] = x; }
        void assign(const int32_t *a, const int32_t *b) { n = 0; for (; a != b; ++a) v[n++] = *a; }
      };
      std::vector<Deps> deps(N);
      bool ok = true;
      for (int64_t r = 0; r < N && ok; ++r) {
        for (int32_t k = iptr[r]; k < iptr[r + 1]; ++k) {
          const int32_t o = other_end(iidx[k]);
          if (o != pred[r] && std::find(deps[r].begin(), deps[r].end(), o) == deps[r].end()) deps[r].push_back(o);
        }
        ok = deps[r].size() <= 4;
      }
      DTICK("dir0 dependencies");
      // ticket order.  Runs are numbered by the position of their first node; a dependency can
      // then live in a run with a LARGER number (the two interleaved rows need each other).  That
      // is harmless while every run has its own resident workgroup.  With fewer workgroups than
      // runs it must be shown that waiting never blocks the dispenser: accepted if a run only
      // looks ahead to the very next ticket and that one looks ahead to nobody (the smallest
      // unfinished ticket and its successor are always held, so both make progress).
      std::vector<int32_t> order(R), ticket_of_run(R);
      for (int64_t k = 0; k < R; ++k) order[k] = (int32_t)k;
      if (cut) std::stable_sort(order.begin(), order.end(), [&](int32_t x, int32_t y) { return first_lev[x] < first_lev[y]; });
      // With strips every strip has its own dispenser and its own (smaller) set of workgroups: the
      // tickets that count are the positions among the strip's OWN runs, a dependency in another
      // strip's run is served by that strip's workgroups, and the test is made whatever the run count
      // (a strip may be launched with fewer workgroups than it has runs: logical strips that share a
      // device, max_workgroups, a partitioned GPU).
      auto look_ahead_ok = [&]() {
        std::vector<int32_t> strip_of_run(R, 0), seen(std::max(nstrips, 1), 0);
        if (own) for (int64_t k = 0; k < R; ++k) strip_of_run[k] = own[g.order[run_head[k]]];
        for (int64_t t = 0; t < R; ++t) ticket_of_run[order[t]] = seen[strip_of_run[order[t]]]++;
        std::vector<std::vector<int32_t>> ahead(std::max(nstrips, 1));  // per strip and ticket: farthest ticket it waits for, minus its own
        for (int s2 = 0; s2 < std::max(nstrips, 1); ++s2) ahead[s2].assign(seen[s2], 0);
        for (int64_t r = 0; r < N; ++r)
          for (int32_t x : deps[r]) {
            const int32_t rm = run_of[r], rt = run_of[x];
            if (strip_of_run[rm] != strip_of_run[rt]) continue;
            const int32_t mine = ticket_of_run[rm], theirs = ticket_of_run[rt];
            ahead[strip_of_run[rm]][mine] = std::max(ahead[strip_of_run[rm]][mine], theirs - mine);
          }
        for (const auto &a : ahead)
          for (size_t t = 0; t < a.size(); ++t)
            if (a[t] > 1 || (a[t] == 1 && t + 1 < a.size() && a[t + 1] > 0)) return false;
        return true;
      };
      // (checked whenever there are more runs than CUs: one workgroup per CU is all that is certain
      // to be resident, whatever max_resident_runs the caller derived from its kernel's LDS use)
      const int64_t resident = max_resident_runs > 0 ? std::min(max_resident_runs, std::max<int64_t>(certainly_resident, 1)) : 0;
      if (ok && ((resident > 0 && R > resident) || nstrips > 1) && !look_ahead_ok()) {
        if (cut) {  // try creation order before giving up
          for (int64_t k = 0; k < R; ++k) order[k] = (int32_t)k;
          ok = look_ahead_ok();
        } else {
          ok = false;
        }
      }
      S.chain_rank.clear(); S.chain_run_ptr.clear(); S.chain_run_order.clear();
      std::vector<int32_t> pred2(N, -1);  // rank visited two steps earlier in the same run (rank-contiguous fallback only)
      if (ok) {
        for (int64_t k = 0; k < R; ++k) {
          S.chain_run_ptr.push_back((int32_t)S.chain_rank.size());
          for (int32_t r = run_head[k]; r >= 0; r = next_of[r]) S.chain_rank.push_back(r);
        }
        S.chain_run_ptr.push_back((int32_t)S.chain_rank.size());
        bool identity = true;
        for (int64_t k = 0; k < R; ++k) identity = identity && order[k] == (int32_t)k;
        if (!identity) S.chain_run_order = order;
      } else {
        // fall back to the rank-contiguous runs of build_runs (hand-over from one or two visits back)
        S.chain_rank.resize(N);
        for (int64_t p = 0; p < N; ++p) S.chain_rank[p] = d == 0 ? (int32_t)p : (int32_t)(N - 1 - p);
        S.chain_run_ptr = S.run_ptr; S.chain_run_order = S.run_order;
        for (int64_t r = 0; r < N; ++r) { deps[r].assign(S.dep_rank.data() + S.dep_ptr[r], S.dep_rank.data() + S.dep_ptr[r + 1]); pred[r] = -1; }
        for (size_t k = 0; k + 1 < S.chain_run_ptr.size(); ++k)
          for (int64_t p = S.chain_run_ptr[k]; p < S.chain_run_ptr[k + 1]; ++p) {
            if (p - 1 >= S.chain_run_ptr[k]) pred[S.chain_rank[p]] = S.chain_rank[p - 1];
            if (p - 2 >= S.chain_run_ptr[k]) pred2[S.chain_rank[p]] = S.chain_rank[p - 2];
          }
      }
      DTICK("dir0 tickets");
      // ---- descriptors, in schedule order (every position is independent of the others: host threads)
      S.desc.assign((size_t)N * W, 0);
      DTICK("dir0 descriptor allocation");
      auto describe = [&](int64_t pa, int64_t pb) {
      for (int64_t p = pa; p < pb; ++p) {
        const int32_t r = S.chain_rank[p];
        int32_t *D = &S.desc[(size_t)p * W];
        const int nout = optr[r + 1] - optr[r], nin = iptr[r + 1] - iptr[r];
        const int nd = (int)deps[r].size();
        uint32_t md = 0;
        for (int k = 0; k < 8; ++k) {
          int32_t e = 0, slot = -1, lbe = 0, xn = 0;
          if (k < nout) {
            e = oidx[optr[r] + k];
            lbe = g.lb_pos_edge[e];
          } else if (k < nout + nin) {
            const int32_t ik = iptr[r] + (k - nout);
            e = iidx[ik];
            // slot of the edge in the outgoing list of the node visited one (0..7) or two (8..15) steps earlier
            const int32_t o = other_end(e);
            const int dist = (pred[r] >= 0 && o == pred[r]) ? 1 : (pred2[r] >= 0 && o == pred2[r]) ? 2 : 0;
            if (dist)
              for (int32_t w = optr[o]; w < optr[o + 1] && w - optr[o] < TrwsGraph::kMaxSlots; ++w)
                if (oidx[w] == e) slot = (w - optr[o]) + 8 * (dist - 1);
            xn = d == 0 ? g.tail[e] : g.head[e];  // the other endpoint: its label feeds the primal
          }
          if (k < nout + nin && g.mdir[e]) md |= 1u << k;
          D[4 + k] = e; D[12 + k] = slot; D[24 + k] = lbe; D[32 + k] = xn;
        }
        D[0] = g.order[r];
        D[1] = r;
        // bit 12: a loader may wait for this node's foreign dependencies while the node two visits
        // earlier in the run is still being computed (its result only becomes visible one visit
        // later): true if every dependency comes before that node in this sweep's order -- what is
        // waited for can then not depend on anything this workgroup still holds back.  False where
        // two chains feed each other (the interleaved last rows).
        const int32_t pm = pred[r], pm2 = pm >= 0 ? pred[pm] : -1;
        const int32_t bound = pm2 >= 0 ? pm2 : pm >= 0 ? pm : r;
        bool ahead = true;
        for (int k = 0; k < nd; ++k) ahead = ahead && (d == 0 ? deps[r][k] < bound : deps[r][k] > bound);
        D[2] = (int32_t)((uint32_t)nout | ((uint32_t)nin << 4) | ((uint32_t)nd << 8) | ((uint32_t)ahead << 12) | (md << 16));
        D[3] = g.lb_pos_node[r];
        for (int k = 0; k < 4; ++k) D[20 + k] = k < nd ? deps[r][k] : 0;
        // strips: which outgoing messages (and whose copy of the flag / label) live in a neighbour's memory
        uint32_t remote = 0;
        if (own) {
          const int32_t mine = own[g.order[r]];
          for (int k = 0; k < nout && k < 8; ++k) {
            const int32_t e = oidx[optr[r] + k];
            const int32_t theirs = own[d == 0 ? g.head[e] : g.tail[e]];
            if (theirs == mine) continue;
            remote |= 1u << k;
            if (theirs > mine) remote |= (1u << (8 + k)) | (1u << 17); else remote |= 1u << 16;
          }
        }
        D[kDescRemote] = (int32_t)remote;
        D[kDescEpos] = g.e_pos[r];
        // slots once more, one byte each (0xff = none), for the compute waves: words 41, 42
        uint32_t pk[2] = {0, 0};
        for (int k = 0; k < 8; ++k) pk[k >> 2] |= (uint32_t)(uint8_t)(int8_t)D[12 + k] << (8 * (k & 3));
        D[41] = (int32_t)pk[0]; D[42] = (int32_t)pk[1];
        uint32_t fetch = 0;
        for (int k = nout; k < nout + nin && k < 8; ++k)
          if (D[12 + k] < 0) fetch |= 1u << k;
        D[kDescFetch] = (int32_t)fetch;
        // twins: outgoing messages k and k' that go to the SAME neighbour (the reference's neighbourhood holds every
        // pair of pixels as two directed edges, dispmap_super.m:279-302, and the orientation step turns both the same
        // way): nibble k of word 56 = k' (k itself without a twin).  Pairs only, mutual; what makes twins carry the same
        // message -- equal weights, shared positions, equal old messages -- is the kernel's to check at run time.
        uint32_t twin = 0;
        {
          int tw[8];
          for (int k = 0; k < 8; ++k) tw[k] = k;
          for (int k = 0; k < nout && k < 8; ++k) {
            if (tw[k] != k) continue;
            const int32_t ek = oidx[optr[r] + k];
            const int32_t to_k = d == 0 ? g.head[ek] : g.tail[ek];
            for (int k2 = k + 1; k2 < nout && k2 < 8; ++k2) {
              const int32_t e2 = oidx[optr[r] + k2];
              if (tw[k2] == k2 && (d == 0 ? g.head[e2] : g.tail[e2]) == to_k) { tw[k] = k2; tw[k2] = k; break; }
            }
          }
          for (int k = 0; k < 8; ++k) twin |= (uint32_t)tw[k] << (4 * k);
        }
        D[kDescTwin] = (int32_t)twin;
      }
      };
      {
        const unsigned hw = std::max(1u, std::thread::hardware_concurrency());
        const int64_t T = std::max<int64_t>(1, std::min<int64_t>({(int64_t)hw / 2, 32, N / 4096 + 1}));
        std::vector<std::thread> pool;
        for (int64_t t = 1; t < T; ++t) pool.emplace_back(describe, N * t / T, N * (t + 1) / T);
        describe(0, N / T);
        for (auto &th : pool) th.join();
      }
      DTICK("dir0 descriptors");
      // Completion flags are raised either in the middle of the next visit (costs a store
      // drain on that run's critical path, but the dependent run can follow closely) or
      // lazily at its end (free).  A run is "lazy" if nobody else reads its flags before it
      // has finished anyway: no node of another run depends on any node but its last.
      const int64_t RR = (int64_t)S.chain_run_ptr.size() - 1;
      std::vector<int32_t> run_at(N);
      for (int64_t k = 0; k < RR; ++k)
        for (int64_t p = S.chain_run_ptr[k]; p < S.chain_run_ptr[k + 1]; ++p) run_at[S.chain_rank[p]] = (int32_t)k;
      std::vector<uint8_t> eager(RR, 0);
      for (int64_t r = 0; r < N; ++r)
        for (int32_t x : deps[r]) {
          const int32_t kx = run_at[x];
          if (x != S.chain_rank[S.chain_run_ptr[kx + 1] - 1]) eager[kx] = 1;
        }
      for (int64_t k = 0; k < RR; ++k)
        for (int64_t p = S.chain_run_ptr[k]; p < S.chain_run_ptr[k + 1]; ++p) S.desc[(size_t)p * W + 40] = eager[k];
      S.chain_run_strip.clear();
      if (own)
        for (int64_t k = 0; k < RR; ++k) S.chain_run_strip.push_back(own[g.order[S.chain_rank[S.chain_run_ptr[k]]]]);
      // ---- speculative schedule of the one long serial run (trws_graph.h: Sweep::Spec)
      S.spec = TrwsGraph::Sweep::Spec();
      if (ok && !own && RR >= 2) {
        TrwsGraph::Sweep::Spec sp;
        int64_t best = -1, len1 = 0, len2 = 0;
        for (int64_t k = 0; k < RR; ++k) {
          const int64_t len = S.chain_run_ptr[k + 1] - S.chain_run_ptr[k];
          if (len > len1) { len2 = len1; len1 = len; best = k; } else if (len > len2) len2 = len;
        }
        const int L = spec_segment_length();
        sp.run = (int32_t)best; sp.c0 = S.chain_run_ptr[best]; sp.c1 = S.chain_run_ptr[best + 1];
        sp.seg_len = L; sp.nseg = (int32_t)(len1 / L); sp.max_len = (int32_t)(len1 - (int64_t)(sp.nseg - 1) * L);
        bool fine = sp.nseg >= 8 && sp.nseg < (1 << 20) && len1 + 8 >= 2 * len2;
        auto seg_of = [&](int64_t p) { return (int32_t)std::min<int64_t>((p - sp.c0) / L, sp.nseg - 1); };
        std::vector<int64_t> pos_of;
        if (fine) {
          pos_of.assign(N, -1);
          for (int64_t p = sp.c0; p < sp.c1; ++p) pos_of[S.chain_rank[p]] = p;
        }
        for (int64_t p = sp.c0; p < sp.c1 && fine; ++p) {
          const int32_t *D = &S.desc[(size_t)p * W];
          const int nout = D[2] & 15, nin = (D[2] >> 4) & 15, nd = (D[2] >> 8) & 15, ntot = nout + nin;
          if (nout > 4 || nin > 4) { fine = false; break; }
          int nfresh = 0, kfirst = ntot, slots[2] = {-1, -1};
          for (int k = nout; k < ntot; ++k) {
            const int sl = D[12 + k];
            if (sl < 0) continue;
            if (sl >= 4) { fine = false; break; }   // only what the node in front hands over, from its first four messages
            if (nfresh == 0) kfirst = k;
            ++nfresh;
            if (slots[0] < 0 || slots[0] == sl) slots[0] = sl;
            else if (slots[1] < 0 || slots[1] == sl) slots[1] = sl;
            else fine = false;
          }
          if (p == sp.c0 ? nfresh != 0 : (nfresh < 1)) fine = false;
          if (ntot - kfirst > 4 || (ntot - kfirst) - nfresh > 3) fine = false;
          // a dependency inside the run must have committed before the runner gets here: an earlier segment
          for (int k = 0; k < nd && fine; ++k) {
            const int32_t x = D[20 + k];
            if (pos_of[x] >= 0 && seg_of(pos_of[x]) >= seg_of(p)) fine = false;
          }
        }
        if (fine) {
          for (int64_t k = 0; k < RR; ++k) {
            if (k == best) for (int32_t q = 0; q < sp.nseg; ++q) { sp.run_ptr.push_back(sp.c0 + q * L); sp.kind.push_back(1 + q); }
            else { sp.run_ptr.push_back(S.chain_run_ptr[k]); sp.kind.push_back(0); }
          }
          sp.run_ptr.push_back(S.chain_run_ptr[RR]);
          // tickets: the runner's first, whatever the direction (the workgroup that draws it serves it before anything
          // else, trws_pipe.hip; it waits for what it needs, holding one CU of 256), then the chain schedule's order
          // with the cut run's ticket replaced by its segments'
          sp.run_order.push_back(-1);
          for (int64_t t = 0; t < RR; ++t) {
            const int32_t k = S.chain_run_order.empty() ? (int32_t)t : S.chain_run_order[t];
            if (k == best) for (int32_t q = 0; q < sp.nseg; ++q) sp.run_order.push_back((int32_t)best + q);
            else sp.run_order.push_back(k < best ? k : k + sp.nseg - 1);
          }
          sp.ok = true;
          S.spec = std::move(sp);
        }
      }
    };
#undef DTICK
    std::thread backward([&] { build_direction(1); });
    build_direction(0);
    backward.join();
  }
  TICK("end");
  return true;
}


namespace {
// local ids of strip s: node_l / edge_l are -1 where the strip holds nothing
void strip_ids(const TrwsGraph &g, int s, std::vector<int32_t> &node_l, std::vector<int32_t> &edge_l,
               std::vector<int32_t> *nodes, std::vector<int32_t> *edges, int64_t *n_own) {
  const int64_t N = (int64_t)g.owner.size(), E = (int64_t)g.tail.size();
  node_l.assign(N, -1); edge_l.assign(E, -1);
  std::vector<uint8_t> halo(N, 0);
  int32_t el = 0;
  for (int64_t e = 0; e < E; ++e) {
    const bool a = g.owner[g.tail[e]] == s, b = g.owner[g.head[e]] == s;
    if (!a && !b) continue;
    edge_l[e] = el++;
    if (edges) edges->push_back((int32_t)e);
    if (!a) halo[g.tail[e]] = 1;
    if (!b) halo[g.head[e]] = 1;
  }
  int32_t nl = 0;
  for (int64_t i = 0; i < N; ++i)
    if (g.owner[i] == s) { node_l[i] = nl++; if (nodes) nodes->push_back((int32_t)i); }
  if (n_own) *n_own = nl;
  for (int64_t i = 0; i < N; ++i)
    if (halo[i]) { node_l[i] = nl++; if (nodes) nodes->push_back((int32_t)i); }
}
}  // namespace

bool build_strip_layout(const TrwsGraph &g, int strip, StripLayout &out, std::string &err) {
  constexpr int W = TrwsGraph::kDescWords;
  out = StripLayout();
  std::vector<int32_t> node_l, edge_l, node_p[2], edge_p[2];
  strip_ids(g, strip, node_l, edge_l, &out.nodes, &out.edges, &out.n_own);
  if (strip > 0) strip_ids(g, strip - 1, node_p[0], edge_p[0], nullptr, nullptr, nullptr);
  if (strip + 1 < g.nstrips) strip_ids(g, strip + 1, node_p[1], edge_p[1], nullptr, nullptr, nullptr);
  bool sound = true;
  for (int d = 0; d < 2; ++d) {
    const TrwsGraph::Sweep &S = g.sweep[d];
    const int64_t R = (int64_t)S.chain_run_ptr.size() - 1;
    out.run_ptr[d].assign(1, 0);
    for (int64_t t = 0; t < R; ++t) {
      const int32_t run = S.chain_run_order.empty() ? (int32_t)t : S.chain_run_order[t];
      if (S.chain_run_strip[run] != strip) continue;
      for (int64_t q = S.chain_run_ptr[run]; q < S.chain_run_ptr[run + 1]; ++q) {
        const int32_t *G = &S.desc[(size_t)q * W];
        const size_t at = out.desc[d].size();
        out.desc[d].insert(out.desc[d].end(), G, G + W);
        int32_t *D = &out.desc[d][at];
        const int nout = G[2] & 15, nin = (G[2] >> 4) & 15, nd = (G[2] >> 8) & 15;
        const uint32_t rem = (uint32_t)G[kDescRemote];
        D[0] = node_l[G[0]];
        D[1] = D[0];  // the flag of a node sits at its local node id
        for (int k = 0; k < 8; ++k) {
          if (k < nout + nin) D[4 + k] = edge_l[G[4 + k]];
          if (k >= nout && k < nout + nin) D[32 + k] = node_l[G[32 + k]];
          if (k < nout && ((rem >> k) & 1)) D[kDescPeerEdge + k] = edge_p[(rem >> (8 + k)) & 1][G[4 + k]];
        }
        for (int k = 0; k < nd && k < 4; ++k) D[20 + k] = node_l[g.order[G[20 + k]]];
        if (rem & (1u << 16)) { D[kDescPeerNode] = node_p[0][G[0]]; out.need_peer[0] = true; }
        if (rem & (1u << 17)) { D[kDescPeerNode + 1] = node_p[1][G[0]]; out.need_peer[1] = true; }
        for (int k = 0; k < 64; ++k)
          if ((k <= 1 || (k >= 4 && k < 4 + nout + nin) || (k >= 20 && k < 20 + nd) || (k >= 32 + nout && k < 32 + nout + nin) ||
               k >= kDescPeerEdge) && D[k] < 0) sound = false;
      }
      out.run_ptr[d].push_back((int32_t)(out.desc[d].size() / W));
    }
  }
  if (!sound) err = "stereo_trws: a strip refers to a node or edge outside its halo (strips must be consecutive in the visiting order)";
  return sound;
}

}  // namespace stereo

extern "C" int stereo_trws_analyze(int64_t N, int64_t E, const uint32_t *conn, int64_t *rank,
                                   int64_t *tail, int64_t *head, int32_t *mdir, int64_t *fwd_ptr,
                                   int64_t *fwd_idx, int64_t *bwd_ptr, int64_t *bwd_idx,
                                   int64_t *level, char *err, size_t errcap) {
  stereo::TrwsGraph g;
  std::string gerr;
  if (!conn && E > 0) return stereo::fail("stereo_trws_analyze: NULL connectivity", err, errcap);
  if (!stereo::build_trws_graph(N, E, conn, g, gerr)) return stereo::fail(gerr, err, errcap);
  const int L = (int)g.level_ptr.size() - 1;
  if (level)
    for (int l = 0; l < L; ++l)
      for (int32_t k = g.level_ptr[l]; k < g.level_ptr[l + 1]; ++k) level[g.order[g.level_ranks[k]]] = l;
  int64_t pf = 0, pb = 0;
  for (int64_t i = 0; i < N; ++i) {
    const int32_t r = g.rank[i];
    if (rank) rank[i] = r;
    if (fwd_ptr) fwd_ptr[i] = pf;
    if (bwd_ptr) bwd_ptr[i] = pb;
    for (int32_t k = g.fptr[r]; k < g.fptr[r + 1]; ++k, ++pf) if (fwd_idx) fwd_idx[pf] = g.fidx[k];
    for (int32_t k = g.bptr[r]; k < g.bptr[r + 1]; ++k, ++pb) if (bwd_idx) bwd_idx[pb] = g.bidx[k];
  }
  if (fwd_ptr) fwd_ptr[N] = pf;
  if (bwd_ptr) bwd_ptr[N] = pb;
  for (int64_t e = 0; e < E; ++e) {
    if (tail) tail[e] = g.tail[e];
    if (head) head[e] = g.head[e];
    if (mdir) mdir[e] = g.mdir[e];
  }
  return 0;
}

static int schedule_impl(int64_t N, int64_t E, const uint32_t *conn, int64_t max_resident_runs,
                         int direction, const int32_t *owner, int nstrips, int64_t *rank_at, int64_t *run_ptr,
                         int64_t *nruns, int64_t *ticket_run, int64_t *pred_rank, int64_t *dep_ptr,
                         int64_t *dep_rank, int64_t *run_strip, int64_t *remote, const char *who, char *err,
                         size_t errcap) {
  stereo::TrwsGraph g;
  std::string gerr;
  if (!conn && E > 0) return stereo::fail(std::string(who) + ": NULL connectivity", err, errcap);
  if (direction != 0 && direction != 1) return stereo::fail(std::string(who) + ": direction must be 0 or 1", err, errcap);
  if (!stereo::build_trws_graph(N, E, conn, g, gerr, max_resident_runs, owner, nstrips)) return stereo::fail(gerr, err, errcap);
  if (!g.fast_ok) return stereo::fail(std::string(who) + ": graph not eligible for the descriptor-driven kernels", err, errcap);
  const stereo::TrwsGraph::Sweep &S = g.sweep[direction];
  constexpr int W = stereo::TrwsGraph::kDescWords;
  const int64_t R = (int64_t)S.chain_run_ptr.size() - 1;
  if (nruns) *nruns = R;
  for (int64_t p = 0; p < N; ++p) if (rank_at) rank_at[p] = S.chain_rank[p];
  for (int64_t k = 0; k <= R; ++k) if (run_ptr) run_ptr[k] = S.chain_run_ptr[k];
  for (int64_t t = 0; t < R; ++t) if (ticket_run) ticket_run[t] = S.chain_run_order.empty() ? t : S.chain_run_order[t];
  for (int64_t k = 0; k < R; ++k) if (run_strip) run_strip[k] = S.chain_run_strip.empty() ? 0 : S.chain_run_strip[k];
  // predecessor and dependencies as the kernels see them: from the descriptors
  int64_t dp = 0;
  std::vector<int64_t> pos_of(N);
  for (int64_t p = 0; p < N; ++p) pos_of[S.chain_rank[p]] = p;
  for (int64_t r = 0; r < N; ++r) {
    const int32_t *D = &S.desc[(size_t)pos_of[r] * W];
    const int nout = D[2] & 15, nin = (D[2] >> 4) & 15, nd = (D[2] >> 8) & 15;
    int64_t pr = -1;
    for (int k = nout; k < nout + nin; ++k)
      if (D[12 + k] >= 0 && D[12 + k] < 8) pr = g.rank[D[32 + k]];
    if (pred_rank) pred_rank[r] = pr;
    if (remote) remote[r] = (uint32_t)D[stereo::kDescRemote];
    if (dep_ptr) dep_ptr[r] = dp;
    for (int k = 0; k < nd; ++k, ++dp) if (dep_rank) dep_rank[dp] = D[20 + k];
  }
  if (dep_ptr) dep_ptr[N] = dp;
  return 0;
}

extern "C" int stereo_trws_schedule(int64_t N, int64_t E, const uint32_t *conn, int64_t max_resident_runs,
                                    int direction, int64_t *rank_at, int64_t *run_ptr, int64_t *nruns,
                                    int64_t *ticket_run, int64_t *pred_rank, int64_t *dep_ptr,
                                    int64_t *dep_rank, char *err, size_t errcap) {
  return schedule_impl(N, E, conn, max_resident_runs, direction, nullptr, 1, rank_at, run_ptr, nruns, ticket_run,
                       pred_rank, dep_ptr, dep_rank, nullptr, nullptr, "stereo_trws_schedule", err, errcap);
}

extern "C" int stereo_trws_schedule_strips(int64_t N, int64_t E, const uint32_t *conn, int64_t max_resident_runs,
                                           int direction, const int32_t *owner, int nstrips, int64_t *rank_at,
                                           int64_t *run_ptr, int64_t *nruns, int64_t *ticket_run,
                                           int64_t *pred_rank, int64_t *dep_ptr, int64_t *dep_rank,
                                           int64_t *run_strip, int64_t *remote, char *err, size_t errcap) {
  if (nstrips > 1 && !owner) return stereo::fail("stereo_trws_schedule_strips: NULL owner", err, errcap);
  return schedule_impl(N, E, conn, max_resident_runs, direction, owner, nstrips, rank_at, run_ptr, nruns, ticket_run,
                       pred_rank, dep_ptr, dep_rank, run_strip, remote, "stereo_trws_schedule_strips", err, errcap);
}

// Host-only view of the speculative schedule (trws_graph.h: Sweep::Spec), for CPU tests of its dependency structure.
// info[0..5] = ok, cut run (index in the chain schedule), c0, c1, segment length, segments; the arrays (may be NULL)
// take the schedule with the cut run as segments: run_ptr (runs + 1), kind (runs), ticket_run (tickets = runs + 1,
// -1 = the runner); *nruns = runs.  Together with stereo_trws_schedule (positions, dependencies) that is everything
// the kernels walk.
extern "C" int stereo_trws_spec_schedule(int64_t N, int64_t E, const uint32_t *conn, int direction, int64_t *info,
                                         int64_t *nruns, int64_t *run_ptr, int64_t *kind, int64_t *ticket_run, char *err,
                                         size_t errcap) {
  if (!conn || !info || (direction != 0 && direction != 1)) return stereo::fail("stereo_trws_spec_schedule: bad argument", err, errcap);
  stereo::TrwsGraph g;
  std::string gerr;
  if (!stereo::build_trws_graph(N, E, conn, g, gerr)) return stereo::fail(gerr, err, errcap);
  const stereo::TrwsGraph::Sweep::Spec &sp = g.sweep[direction].spec;
  info[0] = sp.ok ? 1 : 0; info[1] = sp.run; info[2] = sp.c0; info[3] = sp.c1; info[4] = sp.seg_len; info[5] = sp.nseg;
  if (nruns) *nruns = (int64_t)sp.kind.size();
  for (size_t k = 0; run_ptr && k < sp.run_ptr.size(); ++k) run_ptr[k] = sp.run_ptr[k];
  for (size_t k = 0; kind && k < sp.kind.size(); ++k) kind[k] = sp.kind[k];
  for (size_t k = 0; ticket_run && k < sp.run_order.size(); ++k) ticket_run[k] = sp.run_order[k];
  return 0;
}

// Host-only view of what one strip stores and of its renumbered descriptors (no device needed):
// lets a CPU test check that the ids a strip writes into its neighbours' arrays are the ids the
// neighbours use themselves.
extern "C" int stereo_trws_strip_layout_host(int64_t N, int64_t E, const uint32_t *conn, const int32_t *owner,
                                             int nstrips, int strip, int direction, int64_t *n_nodes, int64_t *n_own,
                                             int64_t *n_edges, int64_t *n_visits, int32_t *nodes, int32_t *edges,
                                             int32_t *desc, char *err, size_t errcap) {
  if (!conn || !owner || nstrips < 1 || strip < 0 || strip >= nstrips || (direction != 0 && direction != 1))
    return stereo::fail("stereo_trws_strip_layout_host: bad argument", err, errcap);
  if (nstrips < 2)  // (one strip is the plain plan: no owner table is kept for it)
    return stereo::fail("stereo_trws_strip_layout_host: a strip layout needs at least two strips", err, errcap);
  try {
    stereo::TrwsGraph g;
    std::string gerr;
    if (!stereo::build_trws_graph(N, E, conn, g, gerr, 0, owner, nstrips)) return stereo::fail(gerr, err, errcap);
    if (!g.fast_ok) return stereo::fail("stereo_trws_strip_layout_host: graph outside the descriptor-driven kernels' range", err, errcap);
    stereo::StripLayout L;
    if (!stereo::build_strip_layout(g, strip, L, gerr)) return stereo::fail(gerr, err, errcap);
    if (n_nodes) *n_nodes = (int64_t)L.nodes.size();
    if (n_own) *n_own = L.n_own;
    if (n_edges) *n_edges = (int64_t)L.edges.size();
    if (n_visits) *n_visits = (int64_t)(L.desc[direction].size() / stereo::TrwsGraph::kDescWords);
    if (nodes) std::copy(L.nodes.begin(), L.nodes.end(), nodes);
    if (edges) std::copy(L.edges.begin(), L.edges.end(), edges);
    if (desc) std::copy(L.desc[direction].begin(), L.desc[direction].end(), desc);
    return 0;
  } catch (const std::exception &e) {
    return stereo::fail(std::string("stereo_trws_strip_layout_host: ") + e.what(), err, errcap);
  }
}
