// TEST INFRASTRUCTURE ONLY -- the order in which the reference's gateway hands equal positions to
// the message code.  cpp/trws_mex.cpp:84-97 builds the ascending order of q(:,p) / qprim(:,p) by
// pushing one (value, index) pair at a time and calling std::sort on the whole vector after every
// push, with a comparator that looks at the value only (:16-20).  std::sort is not stable: up to 16
// elements libstdc++ runs a plain insertion sort, which leaves equal values in index order, beyond
// that its introsort (median-of-three partitions) may swap them -- deterministically for a given
// library.  Restated here with the same calls, so that it follows whatever std::sort of the
// toolchain in use does, exactly like a reference built with that toolchain.
#include <algorithm>
#include <cstdint>
#include <utility>
#include <vector>

extern "C" void oracle_gateway_order(const double *v, int K, int32_t *out) {
  typedef std::pair<double, int> Pair;
  struct Cmp {
    bool operator()(const Pair &a, const Pair &b) const { return a.first < b.first; }
  };
  std::vector<Pair> p;
  for (int j = 0; j < K; ++j) {
    p.push_back(Pair(v[j], j));
    std::sort(p.begin(), p.end(), Cmp());
  }
  for (int j = 0; j < K; ++j) out[j] = p[j].second;
}
