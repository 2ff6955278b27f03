// Microbenchmark (development tool): cycles per message update for the three
// message paths, one wave, data in LDS/L2.  Build & run on the GPU box:
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -Iinclude tools/micro_message.hip \
//         stereo_amd/csrc/trws_graph.cpp -o /tmp/micro && /tmp/micro
#include "../stereo_amd/csrc/trws.hip"

using namespace stereo;

template <int KERNEL, int MODE>
__global__ void micro_kernel(DevParams p, int reps, long long *cycles, int K) {
  extern __shared__ __attribute__((aligned(16))) double lds[];
  const int lane = threadIdx.x & 63;
  double *Di = lds;
  double *scratch = lds + p.Kp + 8;
  for (int k = lane; k < K; k += 64) Di[k] = p.unary[k];
  __syncthreads();
  long long t0 = __builtin_readcyclecounter();
  long long w0 = wall_clock64();
  double acc = 0;
  for (int i = 0; i < reps; ++i) {
    acc += update_message<KERNEL, false, MODE, false>(p, 0, Di, 0.25, scratch, nullptr, lane);
    // keep the message from collapsing: restore something data dependent but cheap
    if (lane < K) p.msg[lane] = p.unary[K + lane] + 1e-9 * acc;
  }
  long long t1 = __builtin_readcyclecounter();
  long long w1 = wall_clock64();
  if (threadIdx.x == 0) { cycles[0] = t1 - t0; cycles[1] = w1 - w0; cycles[2] = (long long)acc; }
}

int main() {
  for (int K : {16, 60, 64, 128}) {
    const int Kp = (K + 1) & ~1;
    std::vector<double> unary(2 * K), pos(K), alpha(1, 1.0);
    srand(1);
    for (auto &v : unary) v = 40.0 * rand() / RAND_MAX;
    for (int k = 0; k < K; ++k) pos[k] = k;
    std::vector<uint16_t> perm(K);
    for (int k = 0; k < K; ++k) perm[k] = k;
    uint8_t mdir = 0;
    DevBuf<double> d_unary, d_pos, d_alpha, d_msg;
    DevBuf<uint16_t> d_perm;
    DevBuf<uint8_t> d_mdir;
    DevBuf<long long> d_cyc;
    d_unary.upload(unary.data(), unary.size()); d_pos.upload(pos.data(), K); d_alpha.upload(alpha.data(), 1);
    d_msg.alloc(K); d_perm.upload(perm.data(), K); d_mdir.upload(&mdir, 1); d_cyc.alloc(4);
    hipMemset(d_msg.p, 0, sizeof(double) * K);
    DevParams p{};
    p.K = K; p.Kp = Kp; p.kernel = 1; p.lambda = 8.0; p.unary = d_unary.p; p.msg = d_msg.p;
    p.pos = d_pos.p; p.perm_pos = d_perm.p; p.alpha = d_alpha.p; p.mdir = d_mdir.p;
    const size_t lds = sizeof(double) * (Kp + 8 + kWaveVecs * Kp + 16);
    const int reps = 2000;
    long long cyc[3];
    for (int mode = 0; mode < 2; ++mode) {
      for (int rep = 0; rep < 2; ++rep) {
        if (mode == 0) hipLaunchKernelGGL((micro_kernel<1, 0>), dim3(1), dim3(64), lds, 0, p, reps, d_cyc.p, K);
        else hipLaunchKernelGGL((micro_kernel<1, 1>), dim3(1), dim3(64), lds, 0, p, reps, d_cyc.p, K);
        hipDeviceSynchronize();
      }
      hipMemcpy(cyc, d_cyc.p, sizeof(cyc), hipMemcpyDeviceToHost);
      int wc_khz = 0; hipDeviceGetAttribute(&wc_khz, hipDeviceAttributeWallClockRate, 0);
      const double secs = (double)cyc[1] / (wc_khz * 1e3);
      printf("K=%d mode=%s: %.0f ticks/message, %.2f us/message, tick rate %.0f MHz (wall clock %d kHz)\n", K,
             mode == 0 ? "exact" : "minplus", (double)cyc[0] / reps, secs / reps * 1e6, cyc[0] / secs / 1e6, wc_khz);
    }
  }
  return 0;
}
