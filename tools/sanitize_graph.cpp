// Development / CI tool: the product's host-side graph analysis (stereo_amd/csrc/trws_graph.cpp: node
// order, orientation, lists, dataflow schedule, strip cuts, descriptors) under AddressSanitizer +
// UndefinedBehaviorSanitizer, on grids of awkward shapes, random multigraphs and row strips.
//   g++ -std=c++17 -O1 -g -fsanitize=address,undefined -fno-sanitize-recover=undefined \
//       -D__HIP_PLATFORM_AMD__ -I/opt/rocm/include -Iinclude tools/sanitize_graph.cpp stereo_amd/csrc/trws_graph.cpp
#include <cstdint>
#include <cstdio>
#include <string>
#include <vector>

#include "../include/stereo_hip.h"

namespace stereo {
std::string &last_error() {
  static thread_local std::string s;
  return s;
}
}  // namespace stereo

static uint64_t state = 0x9E3779B97F4A7C15ull;
static uint32_t rnd(uint32_t n) {
  state ^= state << 13; state ^= state >> 7; state ^= state << 17;
  return (uint32_t)((state >> 33) % n);
}

static std::vector<uint32_t> grid(int H, int W) {
  std::vector<uint32_t> c;
  auto add = [&](int a, int b) { c.push_back(a); c.push_back(b); };
  for (int x = 0; x < W; ++x) for (int r = 0; r + 1 < H; ++r) add(x * H + r, x * H + r + 1);
  for (int x = 0; x < W; ++x) for (int r = 0; r + 1 < H; ++r) add(x * H + r + 1, x * H + r);
  for (int x = 0; x + 1 < W; ++x) for (int r = 0; r < H; ++r) add(x * H + r, (x + 1) * H + r);
  for (int x = 0; x + 1 < W; ++x) for (int r = 0; r < H; ++r) add((x + 1) * H + r, x * H + r);
  return c;
}

int main() {
  char err[512];
  int fails = 0;
  const int shapes[][2] = {{6, 8}, {1, 9}, {9, 1}, {2, 2}, {37, 5}, {5, 41}, {300, 6}, {24, 31}};
  for (auto &s : shapes) {
    const int H = s[0], W = s[1];
    const int64_t N = (int64_t)H * W;
    std::vector<uint32_t> conn = grid(H, W);
    const int64_t E = (int64_t)conn.size() / 2;
    std::vector<int64_t> rank(N), tail(E), head(E), fp(N + 1), fi(E), bp(N + 1), bi(E), level(N);
    std::vector<int32_t> mdir(E);
    if (stereo_trws_analyze(N, E, conn.data(), rank.data(), tail.data(), head.data(), mdir.data(), fp.data(), fi.data(),
                            bp.data(), bi.data(), level.data(), err, sizeof(err))) { std::printf("analyze %dx%d: %s\n", H, W, err); ++fails; }
    for (int G = 1; G <= 4 && G <= H; ++G) {
      std::vector<int32_t> owner(N);
      for (int64_t i = 0; i < N; ++i) owner[i] = (int32_t)(((i % H) * G) / H);
      for (int d = 0; d < 2; ++d)
        for (int64_t resident : {0, 4}) {
          std::vector<int64_t> rank_at(N), run_ptr(N + 1), ticket(N), pred(N), dep_ptr(N + 1), dep(4 * N), strip(N), remote(N);
          int64_t nruns = 0;
          const int rc = stereo_trws_schedule_strips(N, E, conn.data(), resident, d, owner.data(), G, rank_at.data(), run_ptr.data(),
                                                     &nruns, ticket.data(), pred.data(), dep_ptr.data(), dep.data(), strip.data(),
                                                     remote.data(), err, sizeof(err));
          if (rc) { std::printf("schedule %dx%d G %d dir %d: %s\n", H, W, G, d, err); ++fails; }
        }
    }
  }
  for (int trial = 0; trial < 200; ++trial) {  // random multigraphs (the generic kernels' territory)
    const int64_t N = 3 + rnd(40);
    std::vector<uint32_t> conn;
    const int64_t want = 1 + rnd((uint32_t)(3 * N));
    for (int64_t e = 0; e < want; ++e) {
      const uint32_t a = rnd((uint32_t)N), b = rnd((uint32_t)N);
      if (a != b) { conn.push_back(a); conn.push_back(b); }
    }
    const int64_t E = (int64_t)conn.size() / 2;
    if (!E) continue;
    std::vector<int64_t> rank(N), level(N);
    if (stereo_trws_analyze(N, E, conn.data(), rank.data(), nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr,
                            level.data(), err, sizeof(err))) { std::printf("random graph %d: %s\n", trial, err); ++fails; }
  }
  // invalid input is reported, not crashed on
  {
    uint32_t bad[4] = {0, 7, 1, 1};
    if (!stereo_trws_analyze(3, 2, bad, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, err, sizeof(err))) ++fails;
  }
  std::printf(fails ? "SANITIZE_GRAPH_FAILED\n" : "SANITIZE_GRAPH_OK\n");
  return fails ? 1 : 0;
}
