/* TEST INFRASTRUCTURE ONLY -- CPU restatement of the reference's roof-duality path
 * (cpp/rd_mex.cpp + cpp/QPBO-v1.3.src).  Only tests/, smoke() and bench.py's
 * cpu_baseline leg may load this; the product never does.
 *
 * Restated, with the reference lines they follow:
 *   rd_mex.cpp:55-100                 gateway sequence
 *   QPBO.h:760-807                    ComputeWeights (normal form of one 2x2 table)
 *   QPBO.cpp:408-507                  AddPairwiseTerm: one arc pair per directed term,
 *                                     supermodular terms routed to the mate node
 *   QPBO.h:615-624                    AddUnaryTerm
 *   QPBO.cpp:676-728                  TransformToSecondStage(false): mirrored half
 *   QPBO.cpp:786-816, QPBO_extra.cpp:137-239  MergeParallelEdges incl. its arithmetic
 *   QPBO.cpp:818-845                  Solve: labels from the min cut, -1 if equal to mate's
 *   QPBO_postprocessing.cpp:10-120    ComputeWeakPersistencies (two-pass DFS)
 *   QPBO_extra.cpp:13-27,241-254,1151-1233  Improve with the libc rand() permutation
 *   QPBO.cpp:847-917                  energy / lower bound
 * NOT restated: the Boykov-Kolmogorov search-tree max-flow (QPBO_maxflow.cpp).  Any
 * maximum flow yields the same strong labels; this file uses a plain FIFO
 * push-relabel with exact-distance relabelling.  Consequences, stated in the tests:
 * weak-persistency labels between incomparable components (a DFS-order artefact in the
 * reference) and the last bits of energy / lower bound may differ from the reference.
 *
 * Parity status: PINNED against oracle/_ref/libref_qpbo.so (the reference's own QPBO
 * library) in tests/test_oracle_qpbo.py and against tests/golden/rd_runs.npz.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

typedef struct {
  int64_t N;      /* variables; node v + N is the mate of v */
  int64_t na;     /* arcs, stored as pairs (a, a^1) */
  int64_t *head, *next, *first; /* adjacency lists: first[2N], next[na] */
  double *r, *tr;               /* residual capacities, terminal capacities (2N) */
  char *alive;                  /* arc pair still present (merge removes pairs) */
  double zero_energy;
} qg_t;

static void weights(double A, double B, double C, double D, double *ci, double *cj, double *cij,
                    double *cji) {
  *ci = D - A; B -= A; C -= D;
  if (B < 0) { *ci += -B; *cj = B; *cji = B + C; *cij = 0; }
  else if (C < 0) { *ci += C; *cj = -C; *cij = B + C; *cji = 0; }
  else { *cj = 0; *cij = B; *cji = C; }
}

static void set_from(qg_t *g, int64_t a, int64_t v) { g->next[a] = g->first[v]; g->first[v] = a; }
static void remove_from(qg_t *g, int64_t a, int64_t v) {
  if (g->first[v] == a) { g->first[v] = g->next[a]; return; }
  for (int64_t b = g->first[v]; b >= 0; b = g->next[b])
    if (g->next[b] == a) { g->next[b] = g->next[a]; return; }
}
static int64_t mate(const qg_t *g, int64_t v) { return v < g->N ? v + g->N : v - g->N; }
static int64_t tail_of(const qg_t *g, int64_t a) { return g->head[a ^ 1]; }

/* Arc numbering: directed term e owns arcs 4e (i -> j or j'), 4e+1 (its reverse) in the
 * first half and 4e+2, 4e+3 (the mirror pair) in the second half. */
static int64_t arc_mate(int64_t a) { return a ^ 2; }

static int build(qg_t *g, const double *U0, const double *U1, const double *E00, const double *E01,
                 const double *E10, const double *E11, const uint32_t *conn, int64_t N, int64_t E) {
  memset(g, 0, sizeof(*g));
  g->N = N; g->na = 4 * E;
  g->head = malloc(sizeof(int64_t) * (4 * E + 4)); g->next = malloc(sizeof(int64_t) * (4 * E + 4));
  g->first = malloc(sizeof(int64_t) * (2 * N + 2)); g->r = calloc(4 * E + 4, sizeof(double));
  g->tr = calloc(2 * N + 2, sizeof(double)); g->alive = calloc(4 * E + 4, 1);
  if (!g->head || !g->next || !g->first || !g->r || !g->tr || !g->alive) return 1;
  for (int64_t v = 0; v < 2 * N; ++v) g->first[v] = -1;
  /* stage 0: AddPairwiseTerm in edge order; submodular arcs enter the lists now */
  char *super = calloc(E + 1, 1);
  for (int64_t e = 0; e < E; ++e) {
    const int64_t i = conn[2 * e], j = conn[2 * e + 1];
    if (i >= N || j >= N || i == j) { free(super); return 2; }
    double ci, cj, cij, cji;
    const int64_t a = 4 * e, ar = 4 * e + 1;
    if (E01[e] + E10[e] >= E00[e] + E11[e]) {
      weights(E00[e], E01[e], E10[e], E11[e], &ci, &cj, &cij, &cji);
      g->head[a] = j; set_from(g, a, i); set_from(g, ar, j);
      g->tr[j] += cj;
    } else {
      super[e] = 1;
      weights(E01[e], E00[e], E11[e], E10[e], &ci, &cj, &cij, &cji);
      g->head[a] = j + N;
      g->tr[j] -= cj;
    }
    g->head[ar] = i;
    g->tr[i] += ci;
    g->r[a] = cij; g->r[ar] = cji;
    g->alive[a] = g->alive[ar] = 1;
    g->zero_energy += E00[e];
  }
  for (int64_t u = 0; u < N; ++u) { g->tr[u] += U1[u] - U0[u]; g->zero_energy += U0[u]; }
  /* TransformToSecondStage(false), QPBO.cpp:684-728 */
  for (int64_t v = 0; v < N; ++v) g->tr[v + N] = -g->tr[v];
  for (int64_t e = 0; e < E; ++e) {
    const int64_t a0 = 4 * e, a1 = 4 * e + 2;
    const int64_t i0 = g->head[a0 ^ 1];
    if (!super[e]) {
      const int64_t i1 = i0 + N, j1 = g->head[a0] + N;
      set_from(g, a1, j1); set_from(g, a1 ^ 1, i1);
      g->head[a1] = i1; g->head[a1 ^ 1] = j1;
    } else {
      const int64_t i1 = i0 + N, j1 = g->head[a0], j0 = j1 - N;
      set_from(g, a0, i0); set_from(g, a0 ^ 1, j1);
      set_from(g, a1, j0); set_from(g, a1 ^ 1, i1);
      g->head[a1] = i1; g->head[a1 ^ 1] = j0;
    }
    g->r[a1] = g->r[a0]; g->r[a1 ^ 1] = g->r[a0 ^ 1];
    g->alive[a1] = g->alive[a1 ^ 1] = 1;
  }
  free(super);
  return 0;
}

/* QPBO_extra.cpp:137-239.  a1, a2: arcs out of the same node i0 (first half) to j or mate(j).
 * Returns 1 if a1 survives, 0 if the roles were swapped. */
static int merge_pair(qg_t *g, int64_t a1, int64_t a2) {
  int x;
  int64_t m1 = arc_mate(a1), m2 = arc_mate(a2);
  const int64_t i0 = tail_of(g, a1), i1 = mate(g, i0);
  if (g->head[a1] == g->head[a2]) {
    g->r[a1] += g->r[a2]; g->r[a1 ^ 1] += g->r[a2 ^ 1];
    g->r[m1] += g->r[m2]; g->r[m1 ^ 1] += g->r[m2 ^ 1];
    x = 1;
  } else {
    double delta = g->r[m1] - g->r[a1];
    g->tr[tail_of(g, m1)] -= delta; g->tr[g->head[m1]] += delta;
    delta = g->r[m2] - g->r[a2];
    g->tr[tail_of(g, m2)] -= delta; g->tr[g->head[m2]] += delta;
    if (g->r[a1] + g->r[a1 ^ 1] >= g->r[a2] + g->r[a2 ^ 1]) x = 1;
    else { int64_t t = a1; a1 = a2; a2 = t; t = m1; m1 = m2; m2 = t; x = 0; }
    const int64_t j0 = g->head[a1], j1 = g->head[a2];
    const double ci = g->r[a2 ^ 1] - g->r[a2], cij = -g->r[a2], cji = -g->r[a2 ^ 1];
    g->tr[i0] += ci; g->tr[i1] -= ci;
    g->r[a1] += cij; g->r[a1 ^ 1] += cji;
    if (g->r[a1] < 0) {
      delta = g->r[a1]; g->r[a1] = 0; g->r[a1 ^ 1] += delta;
      g->tr[i0] -= delta; g->tr[i1] += delta; g->tr[j0] += delta; g->tr[j1] -= delta;
    }
    if (g->r[a1 ^ 1] < 0) {
      delta = g->r[a1 ^ 1]; g->r[a1 ^ 1] = 0; g->r[a1] += delta;
      g->tr[j0] -= delta; g->tr[j1] += delta; g->tr[i0] += delta; g->tr[i1] -= delta;
    }
    g->r[m1] = g->r[a1]; g->r[m1 ^ 1] = g->r[a1 ^ 1];
  }
  remove_from(g, a2, i0); remove_from(g, a2 ^ 1, g->head[a2]);
  remove_from(g, m2, tail_of(g, m2)); remove_from(g, m2 ^ 1, i1);
  g->alive[a2] = g->alive[a2 ^ 1] = g->alive[m2] = g->alive[m2 ^ 1] = 0;
  return x;
}

/* QPBO.cpp:786-816 */
static void merge_parallel(qg_t *g) {
  int64_t *parent = malloc(sizeof(int64_t) * (g->N + 1));
  for (int64_t i = 0; i < g->N; ++i) {
    for (int64_t a = g->first[i]; a >= 0; a = g->next[a]) {
      int64_t j = g->head[a]; if (j >= g->N) j -= g->N;
      parent[j] = a;
    }
    for (int64_t a = g->first[i], an; a >= 0; a = an) {
      an = g->next[a];
      int64_t j = g->head[a]; if (j >= g->N) j -= g->N;
      if (parent[j] == a) continue;
      if (merge_pair(g, parent[j], a) == 0) { parent[j] = a; an = g->next[a]; }
    }
  }
  free(parent);
}

/* FIFO push-relabel on the doubled graph; on return h[v] < n iff v reaches the sink. */
static void maxflow(qg_t *g, double *ex, double *snk, int64_t *h) {
  const int64_t n = 2 * g->N;
  int64_t *queue = malloc(sizeof(int64_t) * (n + 1));
  char *inq = calloc(n + 1, 1);
  for (int rounds = 0;; ++rounds) {
    /* exact distances to the sink by BFS over reverse residual arcs */
    int64_t qh = 0, qt = 0;
    for (int64_t v = 0; v < n; ++v) { h[v] = n; if (snk[v] > 0) { h[v] = 1; queue[qt++] = v; } }
    while (qh < qt) {
      const int64_t w = queue[qh++];
      for (int64_t a = g->first[w]; a >= 0; a = g->next[a]) {
        const int64_t v = g->head[a];
        if (g->r[a ^ 1] > 0 && h[v] == n) { h[v] = h[w] + 1; queue[qt++] = v; }
      }
    }
    /* discharge active nodes until none is left below height n, then re-check with a BFS */
    qh = qt = 0;
    int64_t cap = n + 1, cnt = 0;
    memset(inq, 0, n + 1);
    for (int64_t v = 0; v < n; ++v) if (ex[v] > 0 && h[v] < n) { queue[qt] = v; qt = (qt + 1) % cap; inq[v] = 1; ++cnt; }
    if (cnt == 0) break;
    int64_t work = 0;
    while (cnt > 0 && work < 8 * n) {
      const int64_t v = queue[qh]; qh = (qh + 1) % cap; --cnt; inq[v] = 0; ++work;
      while (ex[v] > 0 && h[v] < n) {
        if (h[v] == 1 && snk[v] > 0) {
          const double d = ex[v] < snk[v] ? ex[v] : snk[v];
          ex[v] -= d; snk[v] -= d;
          if (!(ex[v] > 0)) break;
        }
        int64_t hmin = n;
        if (snk[v] > 0) hmin = 0;
        for (int64_t a = g->first[v]; a >= 0 && ex[v] > 0; a = g->next[a]) {
          if (!(g->r[a] > 0)) continue;
          const int64_t w = g->head[a];
          if (h[v] == h[w] + 1) {
            const double d = ex[v] < g->r[a] ? ex[v] : g->r[a];
            g->r[a] -= d; g->r[a ^ 1] += d; ex[v] -= d; ex[w] += d;
            if (!inq[w] && h[w] < n) { queue[qt] = w; qt = (qt + 1) % cap; inq[w] = 1; ++cnt; }
          }
          if (g->r[a] > 0 && h[w] < hmin) hmin = h[w];
        }
        if (ex[v] > 0) h[v] = hmin + 1 < n ? hmin + 1 : n;
      }
    }
  }
  free(queue); free(inq);
}

/* QPBO_postprocessing.cpp:10-120 */
static void weak(qg_t *g, int *label) {
  const int64_t N = g->N, n = 2 * N;
  int64_t *parent = malloc(sizeof(int64_t) * n), *cur = malloc(sizeof(int64_t) * n);
  int64_t *region = malloc(sizeof(int64_t) * n), *stack = malloc(sizeof(int64_t) * n);
  char *seen = malloc(n);
  int64_t sp = 0;
  for (int64_t i = 0; i < N; ++i) {
    const int un = label[i] < 0;
    seen[i] = seen[i + N] = !un; region[i] = region[i + N] = un ? -1 : 0;
  }
  for (int64_t s = 0; s < n; ++s) {
    if (seen[s]) continue;
    int64_t i = s; seen[i] = 1; parent[i] = i; cur[i] = g->first[i];
    for (;;) {
      if (cur[i] < 0) { stack[sp++] = i; if (parent[i] == i) break; i = parent[i]; cur[i] = g->next[cur[i]]; continue; }
      const int64_t j = g->head[cur[i]];
      if (!(g->r[cur[i]] > 0) || seen[j]) { cur[i] = g->next[cur[i]]; continue; }
      seen[j] = 1; parent[j] = i; i = j; cur[i] = g->first[i];
    }
  }
  int64_t comp = 0;
  while (sp > 0) {
    int64_t i = stack[--sp];
    if (region[i] > 0) continue;
    region[i] = ++comp; parent[i] = i; cur[i] = g->first[i];
    for (;;) {
      if (cur[i] < 0) { if (parent[i] == i) break; i = parent[i]; cur[i] = g->next[cur[i]]; continue; }
      const int64_t j = g->head[cur[i]];
      if (!(g->r[cur[i] ^ 1] > 0) || region[j] >= 0) { cur[i] = g->next[cur[i]]; continue; }
      parent[j] = i; i = j; cur[i] = g->first[i]; region[i] = comp;
    }
  }
  for (int64_t i = 0; i < N; ++i)
    if (label[i] < 0) {
      if (region[i] > region[i + N]) label[i] = 0;
      else if (region[i] < region[i + N]) label[i] = 1;
    }
  free(parent); free(cur); free(region); free(stack); free(seen);
}

static void strong_labels(const qg_t *g, const int64_t *h, int *label) {
  const int64_t n = 2 * g->N;
  for (int64_t i = 0; i < g->N; ++i) {
    const int li = h[i] < n, lm = h[i + g->N] < n;
    label[i] = li == lm ? -1 : li;
  }
}

/* stage: 0 full gateway behaviour, 1 stop after Solve().  Energy is evaluated from the
 * caller's tables; lower_bound = const + sum_i min(0,t_i) + flow/2 with the constant and
 * terminals of the merged normal form (roof-dual bound, flow invariant). */
int oracle_rd(const double *U0, const double *U1, const double *E00, const double *E01,
              const double *E10, const double *E11, const uint32_t *conn, int64_t N, int64_t E,
              int improve, int stage, double *labelling, double *energy, double *lower_bound,
              double *num_unlabelled) {
  qg_t g;
  int rc = build(&g, U0, U1, E00, E01, E10, E11, conn, N, E);
  if (rc) return rc;
  merge_parallel(&g);
  const int64_t n = 2 * N;
  double *ex = malloc(sizeof(double) * n), *snk = malloc(sizeof(double) * n);
  int64_t *h = malloc(sizeof(int64_t) * n);
  int *label = malloc(sizeof(int) * (N + 1));
  /* QPBO.cpp:899-917 evaluated on the merged graph BEFORE any flow:
   * 2 LB0 = 2 zero + sum_i min(0, tr_i - tr_i') - sum over supermodular pairs (r_a + r_mate);
   * every unit of flow that reaches the sink of the doubled graph raises 2 LB by one. */
  double lb2 = 2 * g.zero_energy, cap_in = 0, left = 0;
  for (int64_t i = 0; i < N; ++i) { const double t = g.tr[i] - g.tr[i + N]; if (t < 0) lb2 += t; }
  for (int64_t a = 0; a < g.na; a += 4)
    if (g.alive[a] && g.head[a] >= N) lb2 -= g.r[a] + g.r[a + 2];
  for (int64_t v = 0; v < n; ++v) { ex[v] = g.tr[v] > 0 ? g.tr[v] : 0; snk[v] = g.tr[v] < 0 ? -g.tr[v] : 0; cap_in += snk[v]; }
  maxflow(&g, ex, snk, h);
  strong_labels(&g, h, label);
  for (int64_t v = 0; v < n; ++v) left += snk[v];
  *lower_bound = (lb2 + (cap_in - left)) / 2;
  if (stage == 0) weak(&g, label);
  double unl = 0;
  for (int64_t i = 0; i < N; ++i) if (label[i] < 0) unl += 1;
  *num_unlabelled = unl;
  if (stage == 0 && improve && unl > 0) {
    int64_t *perm = malloc(sizeof(int64_t) * N);
    for (int64_t i = 0; i < N; ++i) perm[i] = i;
    for (int64_t i = 0; i < N - 1; ++i) {
      int64_t j = i + (int64_t)((rand() / (1.0 + (double)RAND_MAX)) * (double)(N - i));
      if (j > N - 1) j = N - 1;
      const int64_t k = perm[j]; perm[j] = perm[i]; perm[i] = k;
    }
    for (int64_t p = 0; p < N; ++p) {
      const int64_t i = perm[p];
      if ((h[i] < n) != (h[i + N] < n)) continue;
      double c1 = -(ex[i] - snk[i]), c2 = ex[i] - snk[i];
      for (int64_t a = g.first[i]; a >= 0; a = g.next[a]) { c1 += g.r[a]; c2 += g.r[a ^ 1]; }
      const double INFTY = (c1 > c2 ? c1 : c2) + 1;
      double t0 = ex[i] - snk[i] + INFTY, t1 = ex[i + N] - snk[i + N] - INFTY;
      ex[i] = t0 > 0 ? t0 : 0; snk[i] = t0 < 0 ? -t0 : 0;
      ex[i + N] = t1 > 0 ? t1 : 0; snk[i + N] = t1 < 0 ? -t1 : 0;
      maxflow(&g, ex, snk, h);
    }
    for (int64_t i = 0; i < N; ++i) {
      const int li = h[i] < n, lm = h[i + N] < n;
      label[i] = li == lm ? 0 : li;
    }
    free(perm);
  }
  double en = 0;
  for (int64_t i = 0; i < N; ++i) { labelling[i] = label[i]; en += label[i] == 1 ? U1[i] : U0[i]; }
  for (int64_t e = 0; e < E; ++e) {
    const int xi = label[conn[2 * e]] == 1, xj = label[conn[2 * e + 1]] == 1;
    en += xi ? (xj ? E11[e] : E10[e]) : (xj ? E01[e] : E00[e]);
  }
  *energy = en;
  free(ex); free(snk); free(h); free(label);
  free(g.head); free(g.next); free(g.first); free(g.r); free(g.tr); free(g.alive);
  return 0;
}
