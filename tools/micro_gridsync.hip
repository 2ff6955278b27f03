// Development tool: cost of the grid barrier used by qpbo_maxflow_kernel (one agent-scope atomic
// counter, __syncthreads on both sides) for several grid sizes, cooperative launch.
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/micro_gridsync tools/micro_gridsync.hip && /tmp/micro_gridsync
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>

template <int SLEEP>
__global__ __launch_bounds__(1024) void k(int *ctl, int rounds) {
  unsigned gen = 0;
  for (int r = 0; r < rounds; ++r) {
    __syncthreads();
    if (threadIdx.x == 0) {
      ++gen;
      __hip_atomic_fetch_add(ctl, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      const int target = (int)(gen * gridDim.x);
      while (__hip_atomic_load(ctl, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
        if (SLEEP >= 0) __builtin_amdgcn_s_sleep(SLEEP);
      }
    }
    __syncthreads();
  }
}

template <int SLEEP>
void run(int blocks, int threads) {
  int *ctl; hipMalloc(&ctl, 4);
  int rounds = 2000;
  void *args[] = {&ctl, &rounds};
  for (int rep = 0; rep < 2; ++rep) {
    hipMemset(ctl, 0, 4);
    hipDeviceSynchronize();
    auto t0 = std::chrono::steady_clock::now();
    if (hipLaunchCooperativeKernel((const void *)k<SLEEP>, dim3(blocks), dim3(threads), args, 0, 0) != hipSuccess) { printf("launch failed\n"); return; }
    hipDeviceSynchronize();
    double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
    if (rep) printf("blocks %4d x %4d threads, s_sleep(%d): %.2f us per barrier\n", blocks, threads, SLEEP, us / rounds);
  }
  hipFree(ctl);
}

int main() {
  run<1>(256, 1024); run<1>(330, 1024); run<1>(512, 1024); run<0>(330, 1024); run<-1>(330, 1024);
  run<1>(256, 256); run<1>(1024, 256);
  return 0;
}
