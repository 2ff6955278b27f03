// Development tool (MI355X box): wave_sort2 of trws_dev.h (the flat path's two key sorts) against std::sort on random and
// adversarial key sets (duplicates, already sorted, reversed).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -Iinclude tools/micro_sort.hip -o /tmp/micro_sort && /tmp/micro_sort
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <random>
#include <vector>

#include "../stereo_amd/csrc/trws_dev.h"

using namespace stereo;

__global__ void sort_kernel(unsigned *a, unsigned *b, int sets) {
  const int lane = threadIdx.x;
  for (int s = blockIdx.x; s < sets; s += gridDim.x) {
    unsigned x = a[s * 64 + lane], y = b[s * 64 + lane];
    wave_sort2(x, y, lane);
    a[s * 64 + lane] = x; b[s * 64 + lane] = y;
  }
}

int main() {
  const int sets = 20000;
  std::mt19937_64 rng(7);
  std::vector<unsigned> a(sets * 64), b(sets * 64);
  for (int s = 0; s < sets; ++s) {
    const int kind = s % 5;
    for (int i = 0; i < 64; ++i) {
      unsigned u = (unsigned)rng(), v = (unsigned)rng();
      if (kind == 1) { u &= 7; v &= 3; }               // many duplicates
      if (kind == 2) { u = i * 1000u; v = (63 - i); }   // sorted / reversed
      if (kind == 3) { u = i < 60 ? u : 0xFFFFFFFFu; v = i < 60 ? v : 0xFFFFFFFFu; }   // the idle lanes' keys
      if (kind == 4) { u = (u & 1) ? 0u : 0xFFFFFFFFu; v = u; }
      a[s * 64 + i] = u; b[s * 64 + i] = v;
    }
  }
  std::vector<unsigned> wa(a), wb(b);
  for (int s = 0; s < sets; ++s) { std::sort(wa.begin() + s * 64, wa.begin() + s * 64 + 64); std::sort(wb.begin() + s * 64, wb.begin() + s * 64 + 64); }
  unsigned *da, *db;
  hipMalloc(&da, a.size() * 4); hipMalloc(&db, b.size() * 4);
  hipMemcpy(da, a.data(), a.size() * 4, hipMemcpyHostToDevice); hipMemcpy(db, b.data(), b.size() * 4, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(sort_kernel, dim3(256), dim3(64), 0, 0, da, db, sets);
  hipMemcpy(a.data(), da, a.size() * 4, hipMemcpyDeviceToHost); hipMemcpy(b.data(), db, b.size() * 4, hipMemcpyDeviceToHost);
  size_t bad = 0;
  for (size_t i = 0; i < a.size(); ++i) bad += (a[i] != wa[i]) + (b[i] != wb[i]);
  std::printf("micro_sort: %d sets of 2 x 64 keys, %zu wrong entries\n", sets, bad);
  return bad != 0;
}
