// Development tool: what a "visit" of the chunk-parallel compute waves costs in isolation -- four
// waves, each: 9 LDS row reads + adds, H table write, counter barrier, 17-offset window loop (x2
// messages), counter barrier, final write -- with and without four more waves that idle the way
// the loader / storer / primal waves do (spinning with s_sleep on a global flag, or LDS traffic).
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/micro_chunk tools/micro_chunk.hip && /tmp/micro_chunk
#include <hip/hip_runtime.h>
#include <cstdio>

__device__ __forceinline__ double wmin(double v) {
  for (int o = 32; o > 0; o >>= 1) { double t = __shfl_xor(v, o); v = t < v ? t : v; }
  return v;
}

template <int MODE>  // 0: compute waves only; 1: + 4 waves spinning on a global flag with s_sleep(1); 2: + 4 waves in s_barrier only
__global__ __launch_bounds__(512) void k(double *out, long long *cyc, int iters, int *gflag, int use_barrier) {
  __shared__ double rows[9 * 260];
  __shared__ double tab[2][288];
  __shared__ double red[2][4];
  __shared__ int cnt;
  extern __shared__ double aux[];  // 4 x 9 x 260 doubles for the aux waves
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  for (int i = tid; i < 9 * 260; i += 512) rows[i] = i * 0.001;
  for (int i = tid; i < 2 * 288; i += 512) tab[0][i] = 1e300;
  if (tid == 0) cnt = 0;
  __syncthreads();
  const int off = wave * 64 + lane;
  int arrivals = 0;
  auto csync = [&]() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    arrivals += 4;
    if (lane == 0) __hip_atomic_fetch_add(&cnt, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    while (__hip_atomic_load(&cnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) - arrivals < 0) __builtin_amdgcn_s_sleep(0);
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
  };
  long long t0 = __builtin_readcyclecounter();
  double acc = 0;
  for (int it = 0; it < iters; ++it) {
    if (wave < 4) {
      double di = rows[off];
#pragma unroll
      for (int j = 1; j < 9; ++j) di += rows[j * 260 + off];
      for (int m = 0; m < 2; ++m) {
        const double h = 0.25 * di - rows[(1 + m) * 260 + off] + it;
        tab[m][16 + off] = h;
        const double lm = wmin(h);
        if (lane == 0) red[m][wave] = lm;
      }
      csync();
      double o[2];
      for (int m = 0; m < 2; ++m) {
        const double hmin = fmin(fmin(red[m][0], red[m][1]), fmin(red[m][2], red[m][3]));
        double m1 = 1e300;
        for (int d0 = -8; d0 <= 8; d0 += 8) {
          double hs[8];
#pragma unroll
          for (int u = 0; u < 8; ++u) hs[u] = tab[m][16 + off + (d0 + u <= 8 ? d0 + u : 8)];
#pragma unroll
          for (int u = 0; u < 8; ++u) { const double c = 1.5 * fabs((double)(d0 + u)) + hs[u]; m1 = fmin(m1, d0 + u <= 8 ? c : 1e300); }
        }
        o[m] = fmin(m1, hmin + 12.0);
      }
      csync();
      acc += o[0] + o[1];
      rows[(3 + (it & 1)) * 260 + off] = acc * 1e-9;
    } else if (MODE == 1) {
      int spins = 0;
      while (__hip_atomic_load(gflag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < 1 && ++spins < 8) __builtin_amdgcn_s_sleep(1);
    } else if (MODE == 2) {
      // what a loader wave does: ~130 v_readlane (descriptor decode), nine 2 KB rows from memory into LDS
      int w = gflag[64 + lane + (it & 7) * 64];
      int sacc = 0;
#pragma unroll
      for (int r = 0; r < 64; ++r) sacc += __builtin_amdgcn_readlane(w, r);
#pragma unroll
      for (int r = 0; r < 64; ++r) sacc ^= __builtin_amdgcn_readlane(w + sacc, r);
      double v[9][4];
#pragma unroll
      for (int j = 0; j < 9; ++j)
#pragma unroll
        for (int c = 0; c < 4; ++c) v[j][c] = out[512 + ((sacc & 7) + j) * 256 + c * 64 + lane];
#pragma unroll
      for (int j = 0; j < 9; ++j)
#pragma unroll
        for (int c = 0; c < 4; ++c) aux[(wave - 4) * 9 * 260 + j * 260 + c * 64 + lane] = v[j][c];
    } else if (MODE == 3) {
      int w = gflag[64 + lane + (it & 7) * 64];
      int sacc = 0;
#pragma unroll
      for (int r = 0; r < 64; ++r) sacc += __builtin_amdgcn_readlane(w, r);
#pragma unroll
      for (int r = 0; r < 64; ++r) sacc ^= __builtin_amdgcn_readlane(w + sacc, r);
      if (sacc == 12345) out[0] = 1;
    } else if (MODE == 4) {
      double v[9][4];
#pragma unroll
      for (int j = 0; j < 9; ++j)
#pragma unroll
        for (int c = 0; c < 4; ++c) v[j][c] = out[512 + ((it & 7) + j) * 256 + c * 64 + lane];
#pragma unroll
      for (int j = 0; j < 9; ++j)
#pragma unroll
        for (int c = 0; c < 4; ++c) aux[(wave - 4) * 9 * 260 + j * 260 + c * 64 + lane] = v[j][c];
    }
    if (use_barrier) __syncthreads();
  }
  long long t1 = __builtin_readcyclecounter();
  if (tid == 0) cyc[0] = t1 - t0;
  out[tid] = acc;
}

template <int MODE>
void run(const char *what, int use_barrier) {
  double *out; long long *cyc; int *gflag;
  hipMalloc(&out, (512 + 32 * 256) * 8); hipMemset(out, 0, (512 + 32 * 256) * 8); hipMalloc(&cyc, 8); hipMalloc(&gflag, 4 * 1024); hipMemset(gflag, 0, 4 * 1024);
  const int iters = 20000;
  hipFuncSetAttribute((const void *)k<MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, 4 * 9 * 260 * 8);
  hipLaunchKernelGGL(k<MODE>, dim3(1), dim3(512), 4 * 9 * 260 * 8, 0, out, cyc, iters, gflag, use_barrier);
  hipDeviceSynchronize();
  long long h; hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost);
  printf("%-60s %.0f cycles per visit\n", what, (double)h / iters);
  hipFree(out); hipFree(cyc); hipFree(gflag);
}

int main() {
  run<0>("4 compute waves, 4 idle waves exit the loop body at once", 0);
  run<0>("the same + one s_barrier of all 8 waves per visit", 1);
  run<1>("4 compute + 4 waves polling a global flag (s_sleep 1), barrier", 1);
  run<2>("4 compute + 4 loader-like waves (readlanes + 18 KB to LDS), barrier", 1);
  run<3>("4 compute + 4 waves doing 128 v_readlane each, barrier", 1);
  run<4>("4 compute + 4 waves moving 18 KB from memory to LDS each, barrier", 1);
  return 0;
}
