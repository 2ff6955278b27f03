/* TEST INFRASTRUCTURE ONLY -- runs the CPU restatement (trws_oracle.c, qpbo_oracle.c, sort_oracle.cpp)
 * under AddressSanitizer + UndefinedBehaviorSanitizer on small problems with ties, zero weights,
 * duplicate positions and frustrated terms (`make -C oracle sanitize`; tests/test_sanitizers.py).
 * the reference has no sanitizer story; the restatement that every parity test leans
 * on gets one. */
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

int oracle_trws(int kernel, const double *unary, const uint32_t *conn, const double *q, const double *qprim,
                const double *alphas, double tol, double maxiter, double max_relgap, int K, int64_t N, int64_t E,
                int mode, void *msg_fn, void *col_fn, double *labelling, double *energy, double *lower_bound,
                double *iterations, double *trace);
void oracle_set_ordering(int ordering);
int oracle_rd(const double *U0, const double *U1, const double *E00, const double *E01, const double *E10,
              const double *E11, const uint32_t *conn, int64_t N, int64_t E, int improve, int stage,
              double *labelling, double *energy, double *lower_bound, double *num_unlabelled);
int oracle_trws_structure(int64_t N, int64_t E, const uint32_t *conn, int64_t *rank, int64_t *tail, int64_t *head,
                          int32_t *dir, int64_t *fptr, int64_t *fidx, int64_t *bptr, int64_t *bidx);

static uint64_t state = 88172645463325252ull;
static double rnd(void) {  /* xorshift, [0, 1) */
  state ^= state << 13; state ^= state >> 7; state ^= state << 17;
  return (double)(state >> 11) / 9007199254740992.0;
}

static int64_t grid(int H, int W, uint32_t *conn) {  /* dispmap_super.m:279-302, zero based */
  int64_t e = 0;
  for (int c = 0; c < W; ++c) for (int r = 0; r + 1 < H; ++r) { conn[2 * e] = c * H + r; conn[2 * e + 1] = c * H + r + 1; ++e; }
  for (int c = 0; c < W; ++c) for (int r = 0; r + 1 < H; ++r) { conn[2 * e] = c * H + r + 1; conn[2 * e + 1] = c * H + r; ++e; }
  for (int c = 0; c + 1 < W; ++c) for (int r = 0; r < H; ++r) { conn[2 * e] = c * H + r; conn[2 * e + 1] = (c + 1) * H + r; ++e; }
  for (int c = 0; c + 1 < W; ++c) for (int r = 0; r < H; ++r) { conn[2 * e] = (c + 1) * H + r; conn[2 * e + 1] = c * H + r; ++e; }
  return e;
}

int main(void) {
  int fails = 0;
  const int shapes[4][3] = {{6, 7, 5}, {1, 9, 4}, {5, 4, 20}, {2, 2, 3}};
  for (int s = 0; s < 4; ++s) {
    const int H = shapes[s][0], W = shapes[s][1], K = shapes[s][2];
    const int64_t N = (int64_t)H * W;
    uint32_t *conn = malloc(sizeof(uint32_t) * 2 * (4 * N + 4));
    const int64_t E = grid(H, W, conn);
    double *unary = malloc(sizeof(double) * N * K), *q = malloc(sizeof(double) * (E * K + 1)),
           *qp = malloc(sizeof(double) * (E * K + 1)), *al = malloc(sizeof(double) * (E + 1)), *lab = malloc(sizeof(double) * N);
    for (int64_t i = 0; i < N * K; ++i) unary[i] = (double)(int)(rnd() * 10);       /* integer costs: ties */
    for (int64_t i = 0; i < E * K; ++i) { q[i] = (double)(int)(rnd() * K); qp[i] = q[i] + (rnd() < 0.5 ? 0.0 : 0.25); }
    for (int64_t e = 0; e < E; ++e) al[e] = rnd() < 0.1 ? 0.0 : 1.0 + (double)(int)(rnd() * 2);
    for (int kernel = 1; kernel <= 2; ++kernel)
      for (int mode = 0; mode <= 1; ++mode)
        for (int ordering = 0; ordering <= 1; ++ordering) {
          double en, lb, it;
          oracle_set_ordering(ordering);
          const int rc = oracle_trws(kernel, unary, conn, q, qp, al, 2.0, 4, 0.0, K, N, E, mode, 0, 0, lab, &en, &lb, &it, 0);
          oracle_set_ordering(0);
          if (rc || !(en >= lb - 1e-9)) { printf("trws shape %d kernel %d mode %d: rc %d en %g lb %g\n", s, kernel, mode, rc, en, lb); ++fails; }
        }
    /* a frustrated binary problem: unlabelled nodes, weak persistency, Improve */
    double *U0 = calloc(N, sizeof(double)), *U1 = malloc(sizeof(double) * N), *E00 = calloc(E + 1, sizeof(double)),
           *E01 = malloc(sizeof(double) * (E + 1)), *E11 = calloc(E + 1, sizeof(double));
    for (int64_t i = 0; i < N; ++i) U1[i] = (double)(int)(rnd() * 7) - 3;
    for (int64_t e = 0; e < E; ++e) E01[e] = (double)(int)(rnd() * 9) - 4;
    for (int improve = 0; improve <= 1; ++improve) {
      double en, lb, nu;
      srand(7);
      const int rc = oracle_rd(U0, U1, E00, E01, E01, E11, conn, N, E, improve, 0, lab, &en, &lb, &nu);
      if (rc) { printf("rd shape %d improve %d: rc %d\n", s, improve, rc); ++fails; }
    }
    int64_t *rank = malloc(sizeof(int64_t) * N), *tail = malloc(sizeof(int64_t) * (E + 1)), *head = malloc(sizeof(int64_t) * (E + 1)),
            *fptr = malloc(sizeof(int64_t) * (N + 1)), *fidx = malloc(sizeof(int64_t) * (E + 1)), *bptr = malloc(sizeof(int64_t) * (N + 1)),
            *bidx = malloc(sizeof(int64_t) * (E + 1));
    int32_t *dir = malloc(sizeof(int32_t) * (E + 1));
    if (oracle_trws_structure(N, E, conn, rank, tail, head, dir, fptr, fidx, bptr, bidx)) { printf("structure shape %d failed\n", s); ++fails; }
    free(conn); free(unary); free(q); free(qp); free(al); free(lab); free(U0); free(U1); free(E00); free(E01); free(E11);
    free(rank); free(tail); free(head); free(fptr); free(fidx); free(bptr); free(bidx); free(dir);
  }
  printf(fails ? "SANITIZE_CHECK_FAILED\n" : "SANITIZE_CHECK_OK\n");
  return fails ? 1 : 0;
}
