// Development tool: issue cost (cycles per wave64 instruction, one wave alone on a SIMD) of the fp64
// VALU operations the message kernels are made of, and of a dependent LDS round trip.
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/micro_valu tools/micro_valu.hip && /tmp/micro_valu
#include <hip/hip_runtime.h>
#include <cstdio>

template <int OP>
__global__ void k(double *out, long long *cyc, int iters) {
  __shared__ double lds[256];
  double a[8];
  for (int i = 0; i < 8; ++i) a[i] = threadIdx.x * 0.5 + i;
  lds[threadIdx.x] = threadIdx.x;
  __syncthreads();
  double b = out[threadIdx.x];
  long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      if (OP == 0) a[i] = a[i] + b;
      if (OP == 1) a[i] = fmin(a[i], b + i);
      if (OP == 2) a[i] = fmax(a[i], b);
      if (OP == 3) a[i] = a[i] * b;
      if (OP == 4) a[i] = a[i] < b ? a[i] : b + i;        // cmp + cndmask x2
      if (OP == 5) a[i] = lds[((int)a[i]) & 255];          // dependent LDS read
      if (OP == 6) asm("v_min_f64 %0, %1, %2" : "=v"(a[i]) : "v"(a[i]), "v"(b));  // without the IEEE canonicalise
      if (OP == 7) { int x = __double2loint(a[i]); x = x + (int)threadIdx.x; a[i] = __hiloint2double(__double2hiint(a[i]), x); }
      if (OP == 8) { int x = __double2loint(a[i]); x = x < 77 ? x : (int)threadIdx.x; a[i] = __hiloint2double(__double2hiint(a[i]), x); }
      if (OP == 9) { int x = __double2loint(a[i]); x = __builtin_amdgcn_update_dpp(x, x, 0xB1, 0xF, 0xF, false); a[i] = __hiloint2double(__double2hiint(a[i]), x); }
      if (OP == 10) a[i] = __builtin_fma(a[i], b, b);
      if (OP == 11) a[0] = a[0] + b;                       // ONE dependent chain of v_add_f64 (8 links per trip)
      if (OP == 12) a[0] = a[0] * b + a[1];                // dependent mul -> add chain (two instructions per link, contraction off)
      if (OP == 13) { a[i & 1] = a[i & 1] + b; }           // two independent chains
    }
  }
  long long t1 = __builtin_readcyclecounter();
  double s = 0;
  for (int i = 0; i < 8; ++i) s += a[i];
  out[threadIdx.x] = s;
  if (threadIdx.x == 0) cyc[0] = t1 - t0;
  atomicMax((unsigned long long *)cyc + 1, (unsigned long long)(t1 - t0));  // the slowest wave
}

int main() {
  double *out; long long *cyc;
  hipMalloc(&out, 1024 * 8); hipMalloc(&cyc, 16);
  hipMemset(out, 0, 1024 * 8);
  const char *names[] = {"v_add_f64", "v_min_f64(+add)", "v_max_f64", "v_mul_f64", "cmp+select(+add)", "dependent ds_read", "asm v_min_f64", "v_add_u32", "v_min_i32 / cmp+cndmask b32", "v_mov_dpp", "v_fma_f64", "v_add_f64 ONE dependent chain", "v_mul+v_add dependent (per link of 2)", "v_add_f64 two chains"};
  const int iters = 2000;
  long long h;
#define RUN(OP) hipLaunchKernelGGL(k<OP>, dim3(1), dim3(64), 0, 0, out, cyc, iters); hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost); \
  printf("%-22s %.2f cycles per instruction group (8 independent chains)\n", names[OP], (double)h / (iters * 8.0));
  RUN(0) RUN(1) RUN(2) RUN(3) RUN(4) RUN(5) RUN(6) RUN(7) RUN(8) RUN(9) RUN(10) RUN(11) RUN(12) RUN(13)
  // how the cost per wave changes with more waves on the CU (1 wave per SIMD at 256 threads, 2 at 512, 4 at 1024)
  for (int threads = 64; threads <= 1024; threads *= 2) {
    long long hh[2];
    hipMemset(cyc, 0, 16);
    hipLaunchKernelGGL(k<0>, dim3(1), dim3(threads), 0, 0, out, cyc, iters); hipMemcpy(hh, cyc, 16, hipMemcpyDeviceToHost);
    printf("v_add_f64, %4d threads in the workgroup: %.2f cycles per instruction of wave 0, %.2f of the slowest wave\n", threads, (double)hh[0] / (iters * 8.0), (double)hh[1] / (iters * 8.0));
    hipMemset(cyc, 0, 16);
    hipLaunchKernelGGL(k<7>, dim3(1), dim3(threads), 0, 0, out, cyc, iters); hipMemcpy(hh, cyc, 16, hipMemcpyDeviceToHost);
    printf("v_add_u32, %4d threads in the workgroup: %.2f cycles per instruction of wave 0, %.2f of the slowest wave\n", threads, (double)hh[0] / (iters * 8.0), (double)hh[1] / (iters * 8.0));
  }
  return 0;
}
