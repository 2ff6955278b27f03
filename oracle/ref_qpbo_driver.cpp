// TEST INFRASTRUCTURE ONLY -- not part of the product path.
//
// Driver that links the reference's own QPBO v1.3 sources (compiled where they
// lie under /root/reference/cpp/QPBO-v1.3.src, never copied) into
// oracle/_ref/libref_qpbo.so and exposes the roof-duality gateway as a C
// function.  The QPBO library needs nothing the image lacks (libc only), so it
// is built as is.  The reference's mex gateway (cpp/rd_mex.cpp) cannot be built
// here (no mex.h in this image and none is faked); this file restates the
// ~40 lines of gateway logic that sit between MATLAB and the library:
//
//   rd_mex.cpp:55-56   QPBO<double> graph(N, E); AddNode(N)
//   rd_mex.cpp:58-59   AddPairwiseTerm(conn(0,p), conn(1,p), E00,E01,E10,E11) for every p
//   rd_mex.cpp:61-62   AddUnaryTerm(u, U0(u), U1(u)) for every u
//   rd_mex.cpp:65      MergeParallelEdges()
//   rd_mex.cpp:68-69   Solve(); ComputeWeakPersistencies()
//   rd_mex.cpp:83-88   num_unlabelled counted before Improve
//   rd_mex.cpp:91-92   Improve() only if improve && num_unlabelled > 0
//   rd_mex.cpp:96-100  labels, ComputeTwiceEnergy()/2, ComputeTwiceLowerBound()/2
#include <cstdint>
#include <cstdlib>
#include <stdexcept>
#include <string>

#include "QPBO.h"

namespace {
void throw_error(char* msg) { throw std::runtime_error(msg ? msg : "QPBO error"); }
}  // namespace

extern "C" {

// libc rand() state is process global and Improve() draws its permutation from
// it (QPBO_extra.cpp:13-27); tests seed it explicitly.
void ref_srand(unsigned seed) { srand(seed); }

// stage: 0 = full gateway behaviour; 1 = stop after Solve() (strong persistency
// only, no weak persistencies) -- used to pin the flow-invariant part.
int ref_rd_stage(const double* U0, const double* U1, const double* E00, const double* E01,
                 const double* E10, const double* E11, const uint32_t* conn, int64_t N,
                 int64_t E, int improve, int stage, double* labelling, double* energy,
                 double* lower_bound, double* num_unlabelled) {
  try {
    QPBO<double> graph((int)N, (int)E, throw_error);
    graph.AddNode((int)N);
    for (int64_t p = 0; p < E; ++p)
      graph.AddPairwiseTerm((int)conn[2 * p], (int)conn[2 * p + 1], E00[p], E01[p], E10[p],
                            E11[p]);
    for (int64_t u = 0; u < N; ++u) graph.AddUnaryTerm((int)u, U0[u], U1[u]);
    graph.MergeParallelEdges();
    graph.Solve();
    if (stage == 0) graph.ComputeWeakPersistencies();
    double unl = 0;
    for (int64_t u = 0; u < N; ++u)
      if (graph.GetLabel((int)u) < 0) unl += 1;
    *num_unlabelled = unl;
    if (stage == 0 && improve && unl > 0) graph.Improve();
    for (int64_t u = 0; u < N; ++u) labelling[u] = graph.GetLabel((int)u);
    *energy = graph.ComputeTwiceEnergy() / 2;
    *lower_bound = graph.ComputeTwiceLowerBound() / 2;
    return 0;
  } catch (const std::exception&) {
    return 1;
  }
}

int ref_rd(const double* U0, const double* U1, const double* E00, const double* E01,
           const double* E10, const double* E11, const uint32_t* conn, int64_t N, int64_t E,
           int improve, double* labelling, double* energy, double* lower_bound,
           double* num_unlabelled) {
  return ref_rd_stage(U0, U1, E00, E01, E10, E11, conn, N, E, improve, 0, labelling, energy,
                      lower_bound, num_unlabelled);
}

}  // extern "C"
