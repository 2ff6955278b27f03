// Development tool: latency of a flag hand-over between two workgroups (one round trip = A stores,
// B sees it and stores back, A sees that) with agent-scope (sc1) accesses, and with L2-scope
// accesses (sc0 load, plain store) for workgroups that sit on the same XCD.
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/micro_pingpong tools/micro_pingpong.hip && /tmp/micro_pingpong
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>

__device__ __forceinline__ int xcc_id() {
  int v;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(v));
  return v & 0xf;
}
template <int MODE>
__device__ __forceinline__ int ld(const int *p) {
  if (MODE == 0) return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  int v;
  asm volatile("global_load_dword %0, %1, off sc0\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
  return v;
}
template <int MODE>
__device__ __forceinline__ void st(int *p, int v) {
  if (MODE == 0) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); return; }
  asm volatile("global_store_dword %0, %1, off sc0\n\ts_waitcnt vmcnt(0)" ::"v"(p), "v"(v) : "memory");
}

template <int MODE>
__global__ void k(int *flags, int *xcc, int a, int b, int rounds) {
  const int me = blockIdx.x;
  if (threadIdx.x == 0) xcc[me] = xcc_id();
  if (me != a && me != b) return;
  if (threadIdx.x != 0) return;
  int *mine = flags + (me == a ? 0 : 32), *theirs = flags + (me == a ? 32 : 0);
  for (int r = 1; r <= rounds; ++r) {
    int spins = 0;
    if (me == a) {
      st<MODE>(mine, r);
      while (ld<MODE>(theirs) < r) { if (++spins > 2000000) { flags[64] = r; return; } }  // bounded
    } else {
      while (ld<MODE>(theirs) < r) { if (++spins > 2000000) { flags[64] = -r; return; } }
      st<MODE>(mine, r);
    }
  }
}

template <int MODE>
void run(int a, int b, const char *what) {
  int *flags, *xcc; hipMalloc(&flags, 512); hipMalloc(&xcc, 64 * 4);
  hipMemset(flags, 0, 512);
  const int rounds = 20000;
  hipDeviceSynchronize();
  auto t0 = std::chrono::steady_clock::now();
  hipLaunchKernelGGL(k<MODE>, dim3(64), dim3(64), 0, 0, flags, xcc, a, b, rounds);
  hipDeviceSynchronize();
  double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
  int h[64]; hipMemcpy(h, xcc, sizeof(h), hipMemcpyDeviceToHost);
  int gave_up = 0; hipMemcpy(&gave_up, flags + 64, 4, hipMemcpyDeviceToHost);
  if (gave_up) printf("  (gave up waiting in round %d: the two workgroups do not see each other's stores)\n", gave_up);
  fflush(stdout);
  printf("%-34s blocks %2d (xcc %d) <-> %2d (xcc %d): %.3f us per round trip\n", what, a, h[a], b, h[b], us / rounds);
  hipFree(flags); hipFree(xcc);
}

int main() {
  setvbuf(stdout, nullptr, _IONBF, 0);
  run<0>(0, 1, "agent scope (sc1), other XCD");
  run<0>(0, 8, "agent scope (sc1), same XCD");
  // (an "L2 scope" variant -- sc0 load, plain store -- did not hand data over at all on this part: the
  //  agent-scope path is the only one; measured 1.16 us across XCDs, 0.82 us inside one)
  return 0;
}
