// TEST INFRASTRUCTURE ONLY -- not part of the product path.
//
// Links the reference's own TRW-S *type* classes (the per-edge message update,
// the hot loop of the path) into oracle/_ref/libref_trws_types.so:
//   /root/reference/cpp/trw-s/typeStereoLinear.h     (UpdateMessage :329-487, AddColumn :491-518)
//   /root/reference/cpp/trw-s/typeStereoQuadratic.h  (UpdateMessage :329-501, AddColumn :505-531)
// Both headers are self-contained (std headers only) and are compiled where they
// lie.  The rest of the reference TRW-S library (MRFEnergy.h:11, minimize.cpp:6)
// includes <mex.h>, which this image lacks and which is NOT faked: the MRFEnergy
// core (ordering, orientation, sweeps) is therefore unbuildable here and is
// covered by the restatement in oracle/trws_oracle.c only.
//
// The type classes keep their members private and befriend
// `MRFEnergy<Type>`; the harness below is a class template of that name whose
// only job is to call Edge::Initialize / Swap / UpdateMessage / AddColumn on a
// caller-supplied message.  The sort that produces the per-edge index arrays
// restates trws_mex.cpp:84-119 (std::sort of (value,index) pairs after every
// push_back, unsorted values + sort permutation stacked as [q ; qprim]).
#include <algorithm>
#include <cstdlib>
#include <cstring>
#include <utility>
#include <vector>

#include "typeStereoLinear.h"
#include "typeStereoQuadratic.h"

template <class T>
class MRFEnergy {
 public:
  typedef typename T::REAL REAL;
  typedef typename T::Edge Edge;
  typedef typename T::Vector Vector;

  struct Built {
    char* edge_mem;
    char* src_mem;
    char* buf;
    Edge* edge;
    Vector* src;
  };

  static Built build(int K, const double* q, const double* qprim, double alpha, double lambda,
                     int mdir) {
    typedef std::pair<double, int> Pair;
    struct Cmp {
      bool operator()(const Pair& a, const Pair& b) const { return a.first < b.first; }
    };
    std::vector<Pair> qp, qpp;
    for (int j = 0; j < K; ++j) {
      qp.push_back(Pair(q[j], j));
      qpp.push_back(Pair(qprim[j], j));
      std::sort(qp.begin(), qp.end(), Cmp());
      std::sort(qpp.begin(), qpp.end(), Cmp());
    }
    std::vector<REAL> data(2 * K);
    std::vector<int> inds(2 * K);
    for (int j = 0; j < K; ++j) {
      data[j] = q[j];
      inds[j] = qp[j].second;
      data[K + j] = qprim[j];
      inds[K + j] = qpp[j].second;
    }
    typename T::GlobalSize Kg(K);
    typename T::LocalSize Kl(K);
    typename T::EdgeData ed(lambda, alpha, data.data(), inds.data());
    Built b;
    int esz = Edge::GetSizeInBytes(Kg, Kl, Kl, ed);
    b.edge_mem = (char*)calloc(1, esz + 64);
    b.edge = (Edge*)b.edge_mem;
    int vsz = Vector::GetSizeInBytes(Kg, Kl);
    b.src_mem = (char*)calloc(1, vsz + 64);
    b.src = (Vector*)b.src_mem;
    b.buf = (char*)calloc(1, Edge::GetBufSizeInBytes(vsz) + 64);
    b.edge->Initialize(Kg, Kl, Kl, ed, b.src, b.src);
    if (mdir) b.edge->Swap(Kg, Kl, Kl);
    return b;
  }
  static void destroy(Built& b) {
    free(b.edge_mem);
    free(b.src_mem);
    free(b.buf);
  }

  static double update(int K, const double* Di, double gamma, double* msg, const double* q,
                       const double* qprim, double alpha, double lambda, int dir, int mdir) {
    Built b = build(K, q, qprim, alpha, lambda, mdir);
    typename T::GlobalSize Kg(K);
    typename T::LocalSize Kl(K);
    typename T::NodeData nd(const_cast<double*>(Di));
    b.src->Initialize(Kg, Kl, nd);
    typename T::NodeData md(msg);
    b.edge->GetMessagePtr()->Initialize(Kg, Kl, md);
    REAL v = b.edge->UpdateMessage(Kg, Kl, Kl, b.src, gamma, dir, b.buf);
    for (int k = 0; k < K; ++k) msg[k] = b.edge->GetMessagePtr()->GetValue(Kg, Kl, k);
    destroy(b);
    return v;
  }

  static void add_column(int K, const double* q, const double* qprim, double alpha,
                         double lambda, int ksource, double* dest, int dir, int mdir) {
    Built b = build(K, q, qprim, alpha, lambda, mdir);
    typename T::GlobalSize Kg(K);
    typename T::LocalSize Kl(K);
    typename T::NodeData nd(dest);
    b.src->Initialize(Kg, Kl, nd);
    b.edge->AddColumn(Kg, Kl, Kl, ksource, b.src, dir);
    for (int k = 0; k < K; ++k) dest[k] = b.src->GetValue(Kg, Kl, k);
    destroy(b);
  }

  // Vector::ComputeMin (first minimum wins) and ComputeAndSubtractMin.
  static double vec_min(int K, const double* v, int* kmin) {
    typename T::GlobalSize Kg(K);
    typename T::LocalSize Kl(K);
    char* mem = (char*)calloc(1, Vector::GetSizeInBytes(Kg, Kl) + 64);
    Vector* vec = (Vector*)mem;
    typename T::NodeData nd(const_cast<double*>(v));
    vec->Initialize(Kg, Kl, nd);
    typename T::Label km;
    REAL r = vec->ComputeMin(Kg, Kl, km);
    *kmin = km;
    free(mem);
    return r;
  }
};

extern "C" {

// kernel 1 = TypeStereoLinear, 2 = TypeStereoQuadratic (trws_mex.cpp:156-163).
// q / qprim are the gateway's K-vectors q(:,e) and qprim(:,e) of one edge;
// mdir is the edge's Swap() parity, dir the sweep direction (0 fwd, 1 bwd).
// msg: in = message currently stored on the edge, out = new message (normalised).
double ref_update_message(int kernel, int K, const double* Di, double gamma, double* msg,
                          const double* q, const double* qprim, double alpha, double lambda,
                          int dir, int mdir) {
  if (kernel == 1)
    return MRFEnergy<TypeStereoLinear>::update(K, Di, gamma, msg, q, qprim, alpha, lambda, dir,
                                               mdir);
  return MRFEnergy<TypeStereoQuadratic>::update(K, Di, gamma, msg, q, qprim, alpha, lambda, dir,
                                                mdir);
}

void ref_add_column(int kernel, int K, const double* q, const double* qprim, double alpha,
                    double lambda, int ksource, double* dest, int dir, int mdir) {
  if (kernel == 1)
    MRFEnergy<TypeStereoLinear>::add_column(K, q, qprim, alpha, lambda, ksource, dest, dir, mdir);
  else
    MRFEnergy<TypeStereoQuadratic>::add_column(K, q, qprim, alpha, lambda, ksource, dest, dir,
                                               mdir);
}

double ref_vec_min(int K, const double* v, int* kmin) {
  return MRFEnergy<TypeStereoLinear>::vec_min(K, v, kmin);
}

}  // extern "C"
