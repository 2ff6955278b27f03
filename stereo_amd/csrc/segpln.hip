// SegPln proposals of dispmap_globalstereo (dispmap_globalstereo.m:60-201, LO-RANSAC :417-466) on the device:
// the winner-takes-all disparity map by window matching (stereo_segpln_wta) and, over a caller-supplied
// segmentation (the mean-shift / Felzenszwalb segmenters are out of scope), one robustly fitted
// plane per segment (stereo_segpln_planes).  Part of libstereo_hip.so.
//
// Parity: MATLAB's mldivide and its random stream are outside the reference tree, so the restatement in
// oracle/terms.py (segpln_wta, segpln_planes) IS the definition the kernels are tested against -- the plane
// fits bit for bit (every product, sum and quotient below is made in the oracle's order, no contraction; the
// random triples come from the same counter-based generator on both sides), the matching scores to 1e-12
// (exp / log differ in the last bit between libm and the device).
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdint>
#include <future>
#include <memory>
#include <stdexcept>
#include <vector>

#include "../../include/stereo_hip.h"
#include "common.h"

namespace stereo {
namespace {

constexpr int kTile = 16;  // output pixels per workgroup side (window matching)

// vgg_interp2.cxx:245-322, 'linear' branch, one channel plane A (H x W column major), 1-based X (column) / Y (row)
__device__ __forceinline__ double interp2_linear(const double *A, int H, int W, double X, double Y, double oobv) {
  const double dw = (double)W, dh = (double)H;
  double o = oobv;
  if (X >= 1 && Y >= 1) {
    if (X < dw) {
      if (Y < dh) {
        const int xi = (int)X, yi = (int)Y;
        const double u = X - xi, v = Y - yi;
        const size_t k = (size_t)H * (xi - 1) + yi - 1;
        o = A[k] + (A[k + H] - A[k]) * u;
        o += ((A[k + 1] - o) + (A[k + H + 1] - A[k + 1]) * u) * v;
      } else if (Y == dh) {
        const int xi = (int)X;
        const double u = X - xi;
        const size_t k = (size_t)H * xi - 1;
        o = A[k] + (A[k + H] - A[k]) * u;
      }
    } else if (X == dw) {
      if (Y < dh) {
        const int yi = (int)Y;
        const double v = Y - yi;
        const size_t k = (size_t)H * (W - 1) + yi - 1;
        o = A[k] + (A[k + 1] - A[k]) * v;
      } else if (Y == dh) {
        o = A[(size_t)H * W - 1];
      }
    }
  }
  return o;
}

struct WtaArgs {
  const double *images;  // n images, each H x W x C column major (MATLAB layout), one after the other
  const double *P;       // 3 x 4 x n column major
  const double *disps;
  int n, H, W, C, nd, window;
  double col_thresh, min_corr;
  double *out;           // (H - 2 window) x (W - 2 window) column major: disparity of the winner, 0 below min_corr
};

// dispmap_globalstereo.m:76-112.  A workgroup owns a kTile x kTile block of the 'valid' output; per disparity and
// image it evaluates ephoto of the colour difference on the block + its halo (LDS), box-filters it columns first,
// then rows (conv2(filt, filt', ., 'valid'), :99) and adds the images up; the running FIRST maximum of the
// normalised score stays in registers.
__global__ __launch_bounds__(kTile * kTile) void segpln_wta_kernel(WtaArgs a) {
  extern __shared__ double lds[];
  const int wn = 2 * a.window + 1, T = kTile + 2 * a.window;
  double *Yt = lds;            // T x T (column major: [c * T + r])
  double *tt = lds + T * T;    // kTile rows x T columns ([c * kTile + r])
  const int Hv = a.H - 2 * a.window, Wv = a.W - 2 * a.window;
  const int r0 = blockIdx.x * kTile, c0 = blockIdx.y * kTile;   // block origin in the valid output = in the image
  const int tr = threadIdx.x % kTile, tc = threadIdx.x / kTile;
  const size_t Npx = (size_t)a.H * a.W;
  const double inv = 1.0 / (double)wn;
  const double ct = -1.0 / (a.col_thresh * a.C);
  // normaliser: ephoto(-1000 - Rvec) of the FIRST pixel times the number of images (:105-106)
  double x1;
  {
    double s = 0;
    for (int c = 0; c < a.C; ++c) {
      const double r = fmin(fmax(floor(a.images[(size_t)c * Npx] + 0.5), 0.0), 255.0);
      const double f = -1000.0 - r;
      s = s + f * f;
    }
    x1 = (log(2.0) - log(exp(s * ct) + 1.0)) * a.n;
  }
  double best = -__builtin_huge_val(), bestd = 0;
  for (int b = 0; b < a.nd; ++b) {
    double o = 0;
    for (int im = 0; im < a.n; ++im) {
      const double *Pm = a.P + 12 * im;   // P(:, :, im): element (i, j) at Pm[i + 3 j]
      const double *img = a.images + (size_t)im * Npx * a.C;
      __syncthreads();
      for (int t = threadIdx.x; t < T * T; t += kTile * kTile) {
        const int rr = r0 + t % T, cc = c0 + t / T;   // image pixel (0-based)
        double y = 0;
        if (rr < a.H && cc < a.W) {
          const double x = (double)(cc + 1), yy = (double)(rr + 1);
          // X = WC * P(:,1:3,a)' (row [x y 1] times the transposed 3 x 3 block), d = disps(b) * P(:,4,a)
          const double X0 = (x * Pm[0] + yy * Pm[3]) + 1.0 * Pm[6];
          const double X1 = (x * Pm[1] + yy * Pm[4]) + 1.0 * Pm[7];
          const double X2 = (x * Pm[2] + yy * Pm[5]) + 1.0 * Pm[8];
          const double dv = a.disps[b];
          const double Z = 1.0 / (X2 + dv * Pm[11]);
          const double sx = (X0 + dv * Pm[9]) * Z, sy = (X1 + dv * Pm[10]) * Z;
          double s = 0;
          for (int c = 0; c < a.C; ++c) {
            const double v = interp2_linear(img + (size_t)c * Npx, a.H, a.W, sx, sy, -1000.0);
            const double r = fmin(fmax(floor(a.images[(size_t)c * Npx + (size_t)cc * a.H + rr] + 0.5), 0.0), 255.0);  // uint8(images{1}): half away from zero
            const double f = v - r;
            s = s + f * f;
          }
          y = log(2.0) - log(exp(s * ct) + 1.0);
        }
        Yt[t] = y;
      }
      __syncthreads();
      for (int t = threadIdx.x; t < kTile * T; t += kTile * kTile) {   // columns first: t(r, c) = sum_i Y(r + i, c) / wn
        const int r = t % kTile, c = t / kTile;
        double acc = 0;
        for (int i = 0; i < wn; ++i) acc = acc + inv * Yt[c * T + r + i];
        tt[c * kTile + r] = acc;
      }
      __syncthreads();
      double acc = 0;
      for (int j = 0; j < wn; ++j) acc = acc + inv * tt[(tc + j) * kTile + tr];
      o += acc;
    }
    const double score = (x1 - o) / x1;
    if (score > best) { best = score; bestd = a.disps[b]; }   // max(., [], 3): the first maximum
  }
  const int r = r0 + tr, c = c0 + tc;
  if (r < Hv && c < Wv) a.out[(size_t)c * Hv + r] = best < a.min_corr ? 0.0 : bestd;
}

// padarray(., [w w], 'symmetric') (:113)
__global__ void segpln_pad_kernel(const double *in, int Hv, int Wv, int w, double *out) {
  const int H = Hv + 2 * w, W = Wv + 2 * w;
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (int64_t)H * W) return;
  int r = (int)(i % H) - w, c = (int)(i / H) - w;
  r = r < 0 ? -r - 1 : r >= Hv ? 2 * Hv - 1 - r : r;
  c = c < 0 ? -c - 1 : c >= Wv ? 2 * Wv - 1 - c : c;
  out[i] = in[(size_t)c * Hv + r];
}

// ---- plane fits ---------------------------------------------------------------------------------------------
__device__ __forceinline__ uint64_t splitmix64(uint64_t x) {
  x += 0x9E3779B97F4A7C15ull;
  uint64_t z = x;
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  return z ^ (z >> 31);
}

// Cramer's rule in the association of oracle/terms.py:_solve3
__device__ __forceinline__ void solve3(const double a[9] /*row major*/, const double b[3], double out[3]) {
  const double a11 = a[0], a12 = a[1], a13 = a[2], a21 = a[3], a22 = a[4], a23 = a[5], a31 = a[6], a32 = a[7], a33 = a[8];
  const double c11 = a22 * a33 - a23 * a32, c12 = a21 * a33 - a23 * a31, c13 = a21 * a32 - a22 * a31;
  const double det = (a11 * c11 - a12 * c12) + a13 * c13;
  const double b1 = b[0], b2 = b[1], b3 = b[2];
  const double d1 = (b1 * c11 - a12 * (b2 * a33 - a23 * b3)) + a13 * (b2 * a32 - a22 * b3);
  const double d2 = (a11 * (b2 * a33 - a23 * b3) - b1 * c12) + a13 * (a21 * b3 - b2 * a31);
  const double d3 = (a11 * (a22 * b3 - b2 * a32) - a12 * (a21 * b3 - b2 * a31)) + b1 * c13;
  out[0] = d1 / det; out[1] = d2 / det; out[2] = d3 / det;
}

struct FitArgs {
  const double *wta;        // H x W column major
  const int32_t *seg_ptr;   // S + 1
  const int32_t *seg_idx;   // pixel ids grouped by segment, ascending inside a segment
  int H, W, S, max_samples;
  double rt;
  uint64_t seed;
  double *px, *py, *pz;     // scratch: world coordinates of a segment's points, at the segment's offset in seg_idx
  uint8_t *cur, *tmp, *inl; // scratch: inlier flags of a batch of trials (bit b = trial b), of a re-estimated plane, of the best one
  double *proposal;         // 4 x N
  double *planes;           // 3 x S
  int32_t *ninl;            // S
  unsigned long long *prof; // development (STEREO_HIP_SEGPLN_TIMING): 100 MHz ticks per phase of the launch's first workgroup
};

struct FitWork { int32_t map, segment; };   // a workgroup's job: entry of the argument table, segment of that map

constexpr int kFitBatch = 8;   // RANSAC trials judged per pass over a segment's points
constexpr int kFitStage = 4096;   // points per LDS stage of the large segments' least-squares sums (3 doubles each)
constexpr size_t kFitStageBytes = (size_t)kFitStage * 3 * sizeof(double);

// One least-squares sum of the flagged points, as oracle/terms.py:_lstsq3 forms it: 64 strided partial sums (lane l adds
// points l, l + 64, ... in order), then a tree.  K: x x, x y, x z, y y, y z, z z, -x, -y, -z.  Returns the sum in lane 0.
template <int K>
__device__ __forceinline__ double fit_sum(const double *X, const double *Y, const double *Z, const uint8_t *flags, int bit, int n, int lane) {
  // (sixteen points' loads go out together -- they do not depend on each other --, the additions then follow in order:
  //  one point per round trip to memory made a 150 000-point sum a third of a millisecond)
  constexpr int U = 16;
  double acc = 0;
  for (int base = lane; base < n; base += 64 * U) {
    double xv[U], yv[U], zv[U];
    bool fv[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int i = base + 64 * u, ic = i < n ? i : n - 1;
      xv[u] = (K == 0 || K == 1 || K == 2 || K == 6) ? X[ic] : 0.0;
      yv[u] = (K == 1 || K == 3 || K == 4 || K == 7) ? Y[ic] : 0.0;
      zv[u] = (K == 2 || K == 4 || K == 5 || K == 8) ? Z[ic] : 0.0;
      fv[u] = i < n && (flags == nullptr || ((flags[ic] >> bit) & 1));
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const double x = xv[u], y = yv[u], z = zv[u];
      const double t = K == 0 ? x * x : K == 1 ? x * y : K == 2 ? x * z : K == 3 ? y * y : K == 4 ? y * z : K == 5 ? z * z : K == 6 ? -x : K == 7 ? -y : -z;
      const double next = acc + t;
      acc = fv[u] ? next : acc;
    }
  }
  for (int o = 32; o > 0; o >>= 1) {
    const double other = __shfl_down(acc, o, 64);
    acc = acc + other;   // (lanes >= o hold values nobody reads afterwards)
  }
  return acc;
}

// One workgroup of T threads per segment (dispmap_globalstereo.m:164-191 + rplane :417-450).  Control flow is uniform:
// every decision is made on a count or on values every thread holds.  What a segment costs and who does it:
//   * world coordinates of its points, compacted in order: all waves, a chunk of T points per step (counts by ballot);
//   * RANSAC trials: the plane through a trial's three points and its inlier count are pure functions of (seed,
//     segment, trial) -- kFitBatch consecutive trials are drawn by as many lanes side by side and judged in ONE pass
//     over the points (a bit per trial in `cur`, a count per trial), then looked at in order by the reference's
//     bookkeeping (:426-447: which trial improves on the best, how many trials are still needed); trials drawn beyond
//     the point where that bookkeeping stops are simply not looked at;
//   * least squares of an inlier set: nine sums whose order of addition IS the definition (oracle/terms.py:_lstsq3) --
//     one wave per sum where the workgroup has nine (a coarse map's 150 000-pixel segment on ONE wave used to take a
//     third of a millisecond per fit, eleven fits);
// same operations on the same operands in the same order as before, so the planes keep their bits (tests/test_segpln_gpu.py).
template <int T>
__global__ __launch_bounds__(T) void segpln_fit_kernel(const FitArgs *maps, const FitWork *work) {
  constexpr int NW = T / 64;
  constexpr int LW = NW < 9 ? NW : 9;   // waves that sum
  __shared__ int s_cnt[NW * kFitBatch];
  __shared__ int s_tot[kFitBatch];
  __shared__ double s_N[3 * kFitBatch];
  __shared__ double s_acc[9];
  // (one launch holds segments of several maps: the workgroup's map and segment come from the work table, the map's
  //  arguments -- uniform values -- from the argument table)
  const FitWork job = work[blockIdx.x];
  const FitArgs a = maps[job.map];
  const int s = job.segment, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int p0 = a.seg_ptr[s], p1 = a.seg_ptr[s + 1];
  if (p1 <= p0) { if (tid == 0) { a.ninl[s] = 0; a.planes[3 * s] = a.planes[3 * s + 1] = a.planes[3 * s + 2] = 0; } return; }
  double *X = a.px + p0, *Y = a.py + p0, *Z = a.pz + p0;
  uint8_t *cur = a.cur + p0, *tmp = a.tmp + p0, *inl = a.inl + p0;
  unsigned long long tmark = (unsigned long long)wall_clock64();   // (100 MHz)
  auto stamp = [&](int k) {
    if (a.prof && blockIdx.x == 0 && tid == 0) { const unsigned long long now = (unsigned long long)wall_clock64(); a.prof[k] += now - tmark; a.prof[8 + k] += 1; tmark = now; }
  };
  // world coordinates [x y 1] / d of the segment's pixels (:141-145), those with WC(:,3) ~= 0 kept (:168), in order
  int n = 0;
  for (int base = p0; base < p1; base += 4 * T) {   // (four chunks of T pixels per step: their loads go out together)
    double zq[4];
    int pxq[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int i = base + u * T + tid;
      pxq[u] = a.seg_idx[i < p1 ? i : p1 - 1];
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) zq[u] = a.wta[pxq[u]];
    unsigned long long mq[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      zq[u] = 1.0 / zq[u];
      mq[u] = __builtin_amdgcn_ballot_w64(base + u * T + tid < p1 && zq[u] != 0);
      if (NW > 1 && lane == 0) s_cnt[wave * 4 + u] = __builtin_popcountll(mq[u]);
    }
    if (NW > 1) __syncthreads();
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      int before = 0, total = __builtin_popcountll(mq[u]);
      if (NW > 1) {
        total = 0;
        for (int w = 0; w < NW; ++w) { const int c = s_cnt[w * 4 + u]; before += w < wave ? c : 0; total += c; }
      }
      const int at = n + before + __builtin_popcountll(mq[u] & ((1ull << lane) - 1));
      if ((mq[u] >> lane) & 1) {
        const double z = zq[u];
        X[at] = z * (double)(pxq[u] / a.H + 1); Y[at] = z * (double)(pxq[u] % a.H + 1); Z[at] = z;
      }
      n += total;
    }
    if (NW > 1) __syncthreads();
  }
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
  __syncthreads();
  stamp(0);
  // distances of all points to the plane N . p = -1, flags into `dst`; returns the number of inliers
  auto classify = [&](const double N[3], uint8_t *dst) {
    int cnt = 0;
    for (int base = tid; base < n; base += 8 * T) {
      double xv[8], yv[8], zv[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) { const int i = base + u * T, ic = i < n ? i : n - 1; xv[u] = X[ic]; yv[u] = Y[ic]; zv[u] = Z[ic]; }
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int i = base + u * T;
        const double dist = fabs(((xv[u] * N[0] + yv[u] * N[1]) + zv[u] * N[2]) + 1.0);
        const bool v = i < n && dist < a.rt;
        if (i < n) dst[i] = v ? 1 : 0;
        cnt += v ? 1 : 0;
      }
    }
    for (int o = 32; o > 0; o >>= 1) cnt += __shfl_xor(cnt, o, 64);
    if (NW > 1) {
      __syncthreads();   // (the last reader of s_cnt is done)
      if (lane == 0) s_cnt[wave] = cnt;
      __syncthreads();
      cnt = 0;
      for (int w = 0; w < NW; ++w) cnt += s_cnt[w];
    }
    __syncthreads();
    return cnt;
  };
  // least squares of the flagged points (bit `bit` of flags[i]; nullptr: all points) by the normal equations
  auto lstsq = [&](const uint8_t *flags, int bit, double N[3]) {
    __syncthreads();   // (flags written by other waves; the last readers of s_acc are done)
    if (NW == 1) {
      // one wave: the nine sums side by side in one pass (segments of a few hundred points at most)
      double acc[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
      for (int i = lane; i < n; i += 64) {
        if (flags == nullptr || ((flags[i] >> bit) & 1)) {
          const double x = X[i], y = Y[i], z = Z[i];
          acc[0] += x * x; acc[1] += x * y; acc[2] += x * z; acc[3] += y * y; acc[4] += y * z; acc[5] += z * z;
          acc[6] += -x; acc[7] += -y; acc[8] += -z;
        }
      }
      for (int o = 32; o > 0; o >>= 1)
        for (int k = 0; k < 9; ++k) {
          const double other = __shfl_down(acc[k], o, 64);
          acc[k] = acc[k] + other;   // (lanes >= o hold values nobody reads afterwards)
        }
      if (lane == 0) for (int k = 0; k < 9; ++k) s_acc[k] = acc[k];
    } else if (NW >= 8) {
      // large segments: ALL waves stream the points through LDS, kFitStage at a time (the next stage's loads are in flight
      // while this one is summed), and three waves add the sums' terms from there in order (below) --
      // a wave that pulls a 150 000-point segment through its own registers waits a memory round trip per 16 points.
      // A point that is not flagged (or lies behind the segment's end) is staged as (0, 0, 0): its terms are +0 or -0,
      // and adding those leaves a partial sum as it is (a partial sum is never -0: it starts at +0, and +0 + -0 = +0) --
      // the same bits as skipping the point, without a flag to read, an index to clamp or a select per addition.
      extern __shared__ __attribute__((aligned(16))) double fit_stage[];
      constexpr int PER = kFitStage / T;
      double *sX = fit_stage, *sY = sX + kFitStage, *sZ = sY + kFitStage;
      double rx[PER], ry[PER], rz[PER];
      unsigned rfb[PER];   // (the flag BYTES: a test of their bit here would wait for the loads it is meant to leave in flight --
      auto request = [&](int c0) {   //  2.3 of a stage's 5.7 us)
#pragma unroll
        for (int u = 0; u < PER; ++u) {
          const int i = c0 + u * T + tid, ic = i < n ? i : n - 1;
          rx[u] = X[ic]; ry[u] = Y[ic]; rz[u] = Z[ic];
          rfb[u] = flags ? (unsigned)flags[ic] : 0xffu;
        }
      };
      request(0);
      // three waves sum: wave 0 x x, x y, x z | wave 1 y y, y z, -y | wave 2 z z, -z, -x -- seven LDS reads per point
      // instead of one or two per sum (the stage's reads were what bounded it), three independent chains per lane
      double acc0 = 0, acc1 = 0, acc2 = 0;
      for (int c0 = 0; c0 < n; c0 += kFitStage) {
#pragma unroll
        for (int u = 0; u < PER; ++u) {
          const int j = u * T + tid;
          const bool f = c0 + j < n && ((rfb[u] >> bit) & 1u);
          sX[j] = f ? rx[u] : 0.0; sY[j] = f ? ry[u] : 0.0; sZ[j] = f ? rz[u] : 0.0;
        }
        __syncthreads();
        if (c0 + kFitStage < n) request(c0 + kFitStage);
        const int m = n - c0 < kFitStage ? n - c0 : kFitStage;
        const int rounds = (m + 64 * 8 - 1) / (64 * 8);   // (whole rounds of 8 x 64 entries: zeros behind the end)
        // (eight points' LDS reads go out together, the additions follow in order)
        if (wave == 0) {
          for (int r = 0; r < rounds; ++r) {
            double xv[8], yv[8], zv[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) { const int j = (r * 8 + u) * 64 + lane; xv[u] = sX[j]; yv[u] = sY[j]; zv[u] = sZ[j]; }
#pragma unroll
            for (int u = 0; u < 8; ++u) { acc0 = acc0 + xv[u] * xv[u]; acc1 = acc1 + xv[u] * yv[u]; acc2 = acc2 + xv[u] * zv[u]; }
          }
        } else if (wave == 1) {
          for (int r = 0; r < rounds; ++r) {
            double yv[8], zv[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) { const int j = (r * 8 + u) * 64 + lane; yv[u] = sY[j]; zv[u] = sZ[j]; }
#pragma unroll
            for (int u = 0; u < 8; ++u) { acc0 = acc0 + yv[u] * yv[u]; acc1 = acc1 + yv[u] * zv[u]; acc2 = acc2 + (-yv[u]); }
          }
        } else if (wave == 2) {
          for (int r = 0; r < rounds; ++r) {
            double zv[8], xv[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) { const int j = (r * 8 + u) * 64 + lane; zv[u] = sZ[j]; xv[u] = sX[j]; }
#pragma unroll
            for (int u = 0; u < 8; ++u) { acc0 = acc0 + zv[u] * zv[u]; acc1 = acc1 + (-zv[u]); acc2 = acc2 + (-xv[u]); }
          }
        }
        __syncthreads();
      }
      for (int o = 32; o > 0; o >>= 1) {
        const double o0 = __shfl_down(acc0, o, 64), o1 = __shfl_down(acc1, o, 64), o2 = __shfl_down(acc2, o, 64);
        acc0 = acc0 + o0; acc1 = acc1 + o1; acc2 = acc2 + o2;
      }
      if (lane == 0 && wave == 0) { s_acc[0] = acc0; s_acc[1] = acc1; s_acc[2] = acc2; }
      if (lane == 0 && wave == 1) { s_acc[3] = acc0; s_acc[4] = acc1; s_acc[7] = acc2; }
      if (lane == 0 && wave == 2) { s_acc[5] = acc0; s_acc[8] = acc1; s_acc[6] = acc2; }
    } else {
      for (int k = wave; k < 9; k += LW) {
        double v = 0;
        switch (k) {
          case 0: v = fit_sum<0>(X, Y, Z, flags, bit, n, lane); break;
          case 1: v = fit_sum<1>(X, Y, Z, flags, bit, n, lane); break;
          case 2: v = fit_sum<2>(X, Y, Z, flags, bit, n, lane); break;
          case 3: v = fit_sum<3>(X, Y, Z, flags, bit, n, lane); break;
          case 4: v = fit_sum<4>(X, Y, Z, flags, bit, n, lane); break;
          case 5: v = fit_sum<5>(X, Y, Z, flags, bit, n, lane); break;
          case 6: v = fit_sum<6>(X, Y, Z, flags, bit, n, lane); break;
          case 7: v = fit_sum<7>(X, Y, Z, flags, bit, n, lane); break;
          default: v = fit_sum<8>(X, Y, Z, flags, bit, n, lane); break;
        }
        if (lane == 0) s_acc[k] = v;
      }
    }
    __syncthreads();
    double m[9], b[3];
    m[0] = s_acc[0]; m[1] = s_acc[1]; m[2] = s_acc[2]; m[3] = s_acc[1]; m[4] = s_acc[3]; m[5] = s_acc[4]; m[6] = s_acc[2]; m[7] = s_acc[4]; m[8] = s_acc[5];
    b[0] = s_acc[6]; b[1] = s_acc[7]; b[2] = s_acc[8];
    solve3(m, b, N);   // (every thread, from the same nine sums)
  };
  int n_in = n;          // local_WC_points = N when there are too few points for RANSAC (:170)
  bool use_flags = false;
  if (n > 3) {
    int max_i = 3, no_sam = 0, best = 0;
    double max_sam = (double)a.max_samples;
    for (int i = tid; i < n; i += T) inl[i] = 0;
    __syncthreads();
    while ((double)no_sam < max_sam) {
      // trials no_sam + 1 .. no_sam + kFitBatch: lane b draws trial b's three points and solves for its plane
      if (tid < kFitBatch) {
        const int trial = no_sam + 1 + tid;
        int sam[3], got = 0, attempt = 0;
        while (got < 3) {   // oracle/terms.py:segpln_sample
          const uint64_t key = a.seed * 0x100000001B3ull + (uint64_t)(s + 1) * 0x1000193ull + (uint64_t)trial * 64ull + (uint64_t)attempt;
          const int v = (int)(splitmix64(key) % (uint64_t)n);
          ++attempt;
          bool dup = false;
          for (int k = 0; k < got; ++k) dup = dup || sam[k] == v;
          if (!dup) sam[got++] = v;
        }
        double m[9], N[3];
        const double div[3] = {-1.0, -1.0, -1.0};
        for (int k = 0; k < 3; ++k) { m[3 * k] = X[sam[k]]; m[3 * k + 1] = Y[sam[k]]; m[3 * k + 2] = Z[sam[k]]; }
        solve3(m, div, N);
        s_N[3 * tid] = N[0]; s_N[3 * tid + 1] = N[1]; s_N[3 * tid + 2] = N[2];
      }
      __syncthreads();
      stamp(1);
      {
        double Nb[3 * kFitBatch];
#pragma unroll
        for (int k = 0; k < 3 * kFitBatch; ++k) Nb[k] = s_N[k];
        int cnt[kFitBatch];
#pragma unroll
        for (int b = 0; b < kFitBatch; ++b) cnt[b] = 0;
        // (the next four points are asked for before these four are judged: a pass was half memory latency; eight and eight
        //  did not fit the registers)
        constexpr int U = 4;
        double xn[U], yn[U], zn[U];
#pragma unroll
        for (int u = 0; u < U; ++u) { const int i = tid + u * T, ic = i < n ? i : n - 1; xn[u] = X[ic]; yn[u] = Y[ic]; zn[u] = Z[ic]; }
        for (int base = tid; base < n; base += U * T) {
          double xv[U], yv[U], zv[U];
#pragma unroll
          for (int u = 0; u < U; ++u) { xv[u] = xn[u]; yv[u] = yn[u]; zv[u] = zn[u]; }
          if (base + U * T < n) {
#pragma unroll
            for (int u = 0; u < U; ++u) { const int i = base + U * T + u * T, ic = i < n ? i : n - 1; xn[u] = X[ic]; yn[u] = Y[ic]; zn[u] = Z[ic]; }
          }
#pragma unroll
          for (int u = 0; u < U; ++u) {
            const int i = base + u * T;
            const double x = xv[u], y = yv[u], z = zv[u];
            unsigned mask = 0;
#pragma unroll
            for (int b = 0; b < kFitBatch; ++b) {
              const double dist = fabs(((x * Nb[3 * b] + y * Nb[3 * b + 1]) + z * Nb[3 * b + 2]) + 1.0);
              const bool v = i < n && dist < a.rt;
              mask |= v ? (1u << b) : 0u;
              cnt[b] += v ? 1 : 0;
            }
            if (i < n) cur[i] = (uint8_t)mask;
          }
        }
#pragma unroll
        for (int b = 0; b < kFitBatch; ++b) {
          int c = cnt[b];
          for (int o = 32; o > 0; o >>= 1) c += __shfl_xor(c, o, 64);
          if (lane == 0) s_cnt[wave * kFitBatch + b] = c;
        }
        __syncthreads();
        if (tid < kFitBatch) {
          int c = 0;
          for (int w = 0; w < NW; ++w) c += s_cnt[w * kFitBatch + tid];
          s_tot[tid] = c;
        }
        __syncthreads();
      }
      stamp(2);
      // the reference's bookkeeping over the batch, trial by trial (:426-447)
      for (int b = 0; b < kFitBatch && (double)no_sam < max_sam; ++b) {
        ++no_sam;
        const int no_i = s_tot[b];
        if (max_i < no_i) {
          double N[3];
          stamp(3);
          lstsq(cur, b, N);   // re-estimate plane and inliers (:437-440)
          stamp(4);
          const int cnt = classify(N, tmp);
          stamp(5);
          if (cnt > best) {
            { uint8_t *keep = inl; inl = tmp; tmp = keep; }   // (the best flags so far: the arrays swap roles, every thread alike)
            best = cnt;
            max_i = no_i;
            // nsamples(sum(inls), len, 3, conf) (:451-463)
            double q = 1.0;
            for (int k = 0; k < 3; ++k) q = q * ((double)(best - 3 + 1 + k) / (double)(n - 3 + 1 + k));
            double c = 1.0;
            if (!((1.0 - q) < 2.220446049250313e-16)) c = log(1.0 - 0.95) / log(1.0 - q);
            if (c < 1.0) c = 1.0;
            max_sam = c < max_sam ? c : max_sam;
          }
        }
      }
      __syncthreads();   // (s_N, s_tot and cur are rewritten by the next batch)
      stamp(3);
    }
    n_in = best;
    use_flags = true;
  }
  double N_[3] = {0, 0, 0};
  const bool fitted = n_in > 2;
  if (fitted) lstsq(use_flags ? inl : nullptr, 0, N_);
  stamp(6);
  if (tid == 0) {
    a.ninl[s] = n_in;
    for (int k = 0; k < 3; ++k) a.planes[3 * s + k] = fitted ? N_[k] : 0.0;
  }
  if (fitted) {   // proposals{b}(:, M) = [N1 N2 1 N3]' for ALL pixels of the segment (:183-186), NaN / Inf -> 1e-100 (:193-196)
    double v[4] = {N_[0], N_[1], 1.0, N_[2]};
    for (int k = 0; k < 4; ++k) v[k] = (v[k] == v[k] && fabs(v[k]) != __builtin_huge_val()) ? v[k] : 1e-100;
    for (int i = p0 + tid; i < p1; i += T) {
      double *c = a.proposal + 4 * (size_t)a.seg_idx[i];
      c[0] = v[0]; c[1] = v[1]; c[2] = v[2]; c[3] = v[3];
    }
  }
  stamp(7);
}

__global__ void segpln_init_kernel(int64_t N, double *proposal) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < N) { proposal[4 * i] = 0; proposal[4 * i + 1] = 0; proposal[4 * i + 2] = 1; proposal[4 * i + 3] = 0; }
}

template <class F>
int guarded(const char *what, char *err, size_t errcap, F &&f) {
  try {
    if (stereo_hip_device_count() < 1) return fail(std::string(what) + ": no HIP device available", err, errcap);
    f();
    return 0;
  } catch (const HipError &e) {
    return fail(e.msg, err, errcap);
  } catch (const std::exception &e) {
    return fail(std::string(what) + ": " + e.what(), err, errcap);
  }
}

}  // namespace
}  // namespace stereo

using namespace stereo;

extern "C" {

int stereo_segpln_wta(const double *images, int n_images, int H, int W, int C, const double *P, const double *disps,
                      int nd, double col_thresh, int window, double min_corr, double *wta, char *err, size_t errcap) {
  if (!images || !P || !disps || !wta || n_images < 1 || C < 1 || nd < 1 || window < 0 || window > 8)
    return fail("stereo_segpln_wta: bad argument", err, errcap);
  if (H <= 2 * window || W <= 2 * window) return fail("stereo_segpln_wta: image smaller than the matching window", err, errcap);
  return guarded("stereo_segpln_wta", err, errcap, [&] {
    const size_t npx = (size_t)H * W;
    DevBuf<double> dI, dP, dd, valid, full;
    dI.upload(images, npx * C * n_images); dP.upload(P, (size_t)12 * n_images); dd.upload(disps, nd);
    const int Hv = H - 2 * window, Wv = W - 2 * window;
    valid.alloc((size_t)Hv * Wv); full.alloc(npx);
    WtaArgs a{dI.p, dP.p, dd.p, n_images, H, W, C, nd, window, col_thresh, min_corr, valid.p};
    const int T = kTile + 2 * window;
    const size_t lds = sizeof(double) * ((size_t)T * T + (size_t)kTile * T);
    hipLaunchKernelGGL(segpln_wta_kernel, dim3((Hv + kTile - 1) / kTile, (Wv + kTile - 1) / kTile), dim3(kTile * kTile), lds, 0, a);
    STEREO_HIP_CHECK(hipGetLastError());
    hipLaunchKernelGGL(segpln_pad_kernel, dim3((unsigned)((npx + 255) / 256)), dim3(256), 0, 0, valid.p, Hv, Wv, window, full.p);
    STEREO_HIP_CHECK(hipGetLastError());
    STEREO_HIP_CHECK(hipMemcpy(wta, full.p, sizeof(double) * npx, hipMemcpyDeviceToHost));
  });
}

int stereo_segpln_planes_batch(const double *wta, const int32_t *const *segments, int M, int H, int W, double rt, const uint64_t *seeds,
                               int max_samples, double *const *proposals, const int *S, double *const *planes, int32_t *const *inliers,
                               char *err, size_t errcap) {
  if (!wta || !segments || !seeds || !S || M < 1 || H < 1 || W < 1 || max_samples < 1) return fail("stereo_segpln_planes: bad argument", err, errcap);
  const int64_t N = (int64_t)H * W;
  for (int m = 0; m < M; ++m)
    if (!segments[m] || S[m] < 0 || (!(proposals && proposals[m]) && !(planes && planes[m])))
      return fail("stereo_segpln_planes: bad argument", err, errcap);
  return guarded("stereo_segpln_planes", err, errcap, [&] {
    // The maps are independent of each other (a proposal per map, dispmap_globalstereo.m:140-197), and a map's segments fill
    // a fraction of the device only (one workgroup per segment; a coarse map has a handful of segments, one of them most of
    // the image).  So every map has scratch of its own, the host groups the maps' pixels by segment side by side (helper
    // threads), and the segments of ALL maps go into one launch per workgroup size, the three launches on three streams.
    // Everything is kept from call to call on this thread (fourteen maps per object, a dozen allocations each otherwise).
    constexpr int kStreams = 4;
    struct MapScratch {
      DevBuf<double> px, py, pz, dprop, dpl;
      DevBuf<int32_t> dptr, didx, dn;
      DevBuf<uint8_t> cur, tmp, inl;
      PinnedBuf<double> hpl;       // planes and inlier counts come back through pinned memory; the grouped pixel ids, segment
      PinnedBuf<int32_t> hn, hidx, hptr, hlist;   // bounds and launch lists go up from pinned memory: copies that do not hold the host
      size_t N = 0, S = 0;
      size_t count[3] = {0, 0, 0};
      PinnedBuf<double> hprop;     // the 4 x N proposal on its way to the caller's array
      hipEvent_t copied = nullptr;
    };
    struct Pool {
      DevBuf<double> dw;
      DevBuf<FitArgs> dargs;       // a map's arguments; (map, segment) of every workgroup of the three launches
      DevBuf<FitWork> dwork;
      PinnedBuf<FitArgs> hargs;
      PinnedBuf<FitWork> hwork;
      size_t Nw = 0, cap_args = 0, cap_work = 0;
      std::vector<std::unique_ptr<MapScratch>> maps;
      hipStream_t stream[kStreams] = {nullptr, nullptr, nullptr, nullptr};
      hipEvent_t uploaded = nullptr, done[kStreams] = {nullptr, nullptr, nullptr, nullptr};
      int device = -1;
    };
    static thread_local Pool this_threads_pool;
    Pool &pool = this_threads_pool;   // (a plain reference: the helper threads below must see THIS thread's pool, not theirs)
    int device = 0;
    STEREO_HIP_CHECK(hipGetDevice(&device));
    if (pool.device != device) {   // (buffers of another device are released; its streams and events stay with it)
      pool.maps.clear(); pool.Nw = 0; pool.cap_args = 0; pool.cap_work = 0; pool.device = device;
      for (int k = 0; k < kStreams; ++k) {
        STEREO_HIP_CHECK(hipStreamCreateWithFlags(&pool.stream[k], hipStreamNonBlocking));
        STEREO_HIP_CHECK(hipEventCreateWithFlags(&pool.done[k], hipEventDisableTiming));
      }
      STEREO_HIP_CHECK(hipEventCreateWithFlags(&pool.uploaded, hipEventDisableTiming));
    }
    if (pool.Nw < (size_t)N) { pool.dw.alloc(N); pool.Nw = (size_t)N; }
    while (pool.maps.size() < (size_t)M) pool.maps.emplace_back(new MapScratch);
    static const bool attr_set = [] {
      return hipFuncSetAttribute((const void *)segpln_fit_kernel<512>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)kFitStageBytes) == hipSuccess;
    }();
    if (!attr_set) throw HipError{"stereo_segpln_planes: hipFuncSetAttribute(MaxDynamicSharedMemorySize) failed"};
    static const bool timing = std::getenv("STEREO_HIP_SEGPLN_TIMING") != nullptr;   // (development: the three launches' device time, one map at a time)
    struct Drain {   // whatever way this call ends, nothing of it is still running when the scratch is used again
      Pool &pool;
      ~Drain() { for (int k = 0; k < kStreams; ++k) (void)hipStreamSynchronize(pool.stream[k]); }
    } drain{pool};
    // (the disparity map: pageable memory, so this copy returns when the source has been read)
    STEREO_HIP_CHECK(hipMemcpyAsync(pool.dw.p, wta, sizeof(double) * N, hipMemcpyHostToDevice, pool.stream[0]));
    STEREO_HIP_CHECK(hipEventRecord(pool.uploaded, pool.stream[0]));
    std::vector<int> order((size_t)M);
    for (int m = 0; m < M; ++m) order[m] = m;
    std::stable_sort(order.begin(), order.end(), [&](int x, int y) { return S[x] < S[y]; });
    for (int r = 0; r < M; ++r) {   // scratch first: the helper threads below only write into it
      const int Sm = S[order[r]];
      MapScratch &sc = *pool.maps[r];
      if (sc.N < (size_t)N) {
        sc.px.alloc(N); sc.py.alloc(N); sc.pz.alloc(N); sc.cur.alloc(N); sc.tmp.alloc(N); sc.inl.alloc(N); sc.dprop.alloc(4 * N);
        sc.didx.alloc(N); sc.hidx.alloc(N);
        sc.N = (size_t)N;
      }
      if (sc.S < (size_t)Sm + 2) {
        const size_t cap = (size_t)Sm + 2;
        sc.dptr.alloc(cap); sc.dpl.alloc(3 * cap); sc.dn.alloc(cap);
        sc.hpl.alloc(3 * cap); sc.hn.alloc(cap); sc.hptr.alloc(cap); sc.hlist.alloc(cap);
        sc.S = cap;
      }
    }
    // pixels grouped by segment, ascending pixel id inside a segment (= MATLAB's logical indexing order); label 0 = none.
    // Host work, a map at a time in two passes over its labels -- and the maps do not depend on each other: every map
    // but the first is grouped by a helper thread, the first (and a single map) here; false: a label beyond [0, S].
    auto group = [&](int r) -> bool {
      const int m = order[r], Sm = S[m];
      MapScratch &sc = *pool.maps[r];
      const int32_t *seg = segments[m];
      int32_t *ptr = sc.hptr.p, *idx = sc.hidx.p;   // ptr[l] .. ptr[l + 1]: the pixels of label l
      std::fill(ptr, ptr + Sm + 2, 0);
      uint32_t beyond = 0;   // (labels are checked where they are counted)
      for (int64_t i = 0; i < N; ++i) {
        const uint32_t l = (uint32_t)seg[i];
        if (l <= (uint32_t)Sm) ++ptr[(size_t)l + 1]; else beyond = 1;
      }
      if (beyond) return false;
      for (int l = 0; l <= Sm; ++l) ptr[l + 1] += ptr[l];
      {
        std::vector<int32_t> at(ptr, ptr + Sm + 1);
        for (int64_t i = 0; i < N; ++i) idx[(size_t)at[seg[i]]++] = (int32_t)i;
      }
      // segments by size: the workgroup grows with the passes over the points it has to make
      constexpr int kLargeSegment = 4096, kMediumSegment = 384;
      size_t *count = sc.count;
      count[0] = count[1] = count[2] = 0;
      for (int sg = 0; sg < Sm; ++sg) {
        const int len = ptr[sg + 2] - ptr[sg + 1];
        ++count[len > kLargeSegment ? 0 : len > kMediumSegment ? 1 : 2];
      }
      size_t at[3] = {0, count[0], count[0] + count[1]};
      for (int sg = 0; sg < Sm; ++sg) {
        const int len = ptr[sg + 2] - ptr[sg + 1];
        sc.hlist.p[at[len > kLargeSegment ? 0 : len > kMediumSegment ? 1 : 2]++] = sg;
      }
      return true;
    };
    std::vector<std::future<bool>> grouped((size_t)M);
    struct Join {   // (no helper outlives the call, whichever way it ends)
      std::vector<std::future<bool>> &f;
      ~Join() { for (auto &x : f) if (x.valid()) x.wait(); }
    } join{grouped};
    for (int r = 1; r < M; ++r) grouped[r] = std::async(std::launch::async, group, r);
    // uploads and the default proposal of every map on the first stream, as the maps' groupings arrive
    const hipStream_t st0 = pool.stream[0];
    std::vector<FitArgs> args((size_t)M + 1);
    size_t total[3] = {0, 0, 0};
    for (int r = 0; r < M; ++r) {
      const int m = order[r], Sm = S[m];
      MapScratch &sc = *pool.maps[r];
      if (!(r == 0 ? group(0) : grouped[r].get())) throw std::runtime_error("segment label out of range [0, S]");
      // (the kernels index segments 0 .. S - 1 = labels 1 .. S: the bounds go up from ptr + 1)
      STEREO_HIP_CHECK(hipMemcpyAsync(sc.dptr.p, sc.hptr.p + 1, sizeof(int32_t) * ((size_t)Sm + 1), hipMemcpyHostToDevice, st0));
      STEREO_HIP_CHECK(hipMemcpyAsync(sc.didx.p, sc.hidx.p, sizeof(int32_t) * N, hipMemcpyHostToDevice, st0));
      hipLaunchKernelGGL(segpln_init_kernel, dim3((unsigned)((N + 255) / 256)), dim3(256), 0, st0, N, sc.dprop.p);
      args[r] = FitArgs{pool.dw.p, sc.dptr.p, sc.didx.p, H, W, Sm, max_samples, rt, seeds[m], sc.px.p, sc.py.p, sc.pz.p, sc.cur.p, sc.tmp.p, sc.inl.p,
                        sc.dprop.p, sc.dpl.p, sc.dn.p, nullptr};
      for (int c = 0; c < 3; ++c) total[c] += sc.count[c];
    }
    STEREO_HIP_CHECK(hipGetLastError());
    // ONE launch per workgroup size for the segments of ALL maps (the device runs a few queues side by side, not fourteen:
    // a stream per map left most maps waiting behind each other): work tables of (map, segment), the large segments by
    // falling size -- the longest workgroup, which bounds the call, starts first
    const size_t jobs = total[0] + total[1] + total[2];
    if (pool.cap_work < jobs) { pool.dwork.alloc(jobs); pool.hwork.alloc(jobs); pool.cap_work = jobs; }
    if (pool.cap_args < (size_t)M + 1) { pool.dargs.alloc((size_t)M + 1); pool.hargs.alloc((size_t)M + 1); pool.cap_args = (size_t)M + 1; }
    {
      std::vector<std::pair<int32_t, FitWork>> large;
      large.reserve(total[0]);
      size_t at[3] = {0, total[0], total[0] + total[1]};
      for (int r = 0; r < M; ++r) {
        const MapScratch &sc = *pool.maps[r];
        const int32_t *l = sc.hlist.p, *ptr = sc.hptr.p;
        for (size_t k = 0; k < sc.count[0]; ++k) large.push_back({ptr[l[k] + 2] - ptr[l[k] + 1], FitWork{r, l[k]}});
        for (size_t k = 0; k < sc.count[1]; ++k) pool.hwork.p[at[1]++] = FitWork{r, l[sc.count[0] + k]};
        for (size_t k = 0; k < sc.count[2]; ++k) pool.hwork.p[at[2]++] = FitWork{r, l[sc.count[0] + sc.count[1] + k]};
      }
      std::stable_sort(large.begin(), large.end(), [](const std::pair<int32_t, FitWork> &x, const std::pair<int32_t, FitWork> &y) { return x.first > y.first; });
      for (const auto &j : large) pool.hwork.p[at[0]++] = j.second;
    }
    DevBuf<unsigned long long> dprof;
    if (timing && total[0]) {   // (development: the launch's first workgroup -- the largest segment -- keeps time per phase, through an entry of its own)
      dprof.alloc(16);
      STEREO_HIP_CHECK(hipMemset(dprof.p, 0, 16 * sizeof(unsigned long long)));
      args[M] = args[pool.hwork.p[0].map];
      args[M].prof = dprof.p;
      pool.hwork.p[0].map = M;
    }
    std::copy(args.begin(), args.end(), pool.hargs.p);
    STEREO_HIP_CHECK(hipMemcpyAsync(pool.dargs.p, pool.hargs.p, sizeof(FitArgs) * ((size_t)M + 1), hipMemcpyHostToDevice, st0));
    if (jobs) STEREO_HIP_CHECK(hipMemcpyAsync(pool.dwork.p, pool.hwork.p, sizeof(FitWork) * jobs, hipMemcpyHostToDevice, st0));
    STEREO_HIP_CHECK(hipEventRecord(pool.uploaded, st0));
    const FitWork *w0 = pool.dwork.p, *w1 = w0 + total[0], *w2 = w1 + total[1];
    if (!timing) {
      for (int k = 1; k < 3; ++k) STEREO_HIP_CHECK(hipStreamWaitEvent(pool.stream[k], pool.uploaded, 0));
      if (total[0]) hipLaunchKernelGGL(segpln_fit_kernel<512>, dim3((unsigned)total[0]), dim3(512), kFitStageBytes, st0, pool.dargs.p, w0);
      if (total[1]) hipLaunchKernelGGL(segpln_fit_kernel<256>, dim3((unsigned)total[1]), dim3(256), 0, pool.stream[1], pool.dargs.p, w1);
      if (total[2]) hipLaunchKernelGGL(segpln_fit_kernel<64>, dim3((unsigned)total[2]), dim3(64), 0, pool.stream[2], pool.dargs.p, w2);
      STEREO_HIP_CHECK(hipGetLastError());
      for (int k = 1; k < 3; ++k) {
        STEREO_HIP_CHECK(hipEventRecord(pool.done[k], pool.stream[k]));
        STEREO_HIP_CHECK(hipStreamWaitEvent(st0, pool.done[k], 0));
      }
    } else {   // one after the other, timed
      hipEvent_t ev[4] = {nullptr, nullptr, nullptr, nullptr};
      for (auto &e : ev) STEREO_HIP_CHECK(hipEventCreate(&e));
      STEREO_HIP_CHECK(hipEventRecord(ev[0], st0));
      if (total[0]) hipLaunchKernelGGL(segpln_fit_kernel<512>, dim3((unsigned)total[0]), dim3(512), kFitStageBytes, st0, pool.dargs.p, w0);
      STEREO_HIP_CHECK(hipEventRecord(ev[1], st0));
      if (total[1]) hipLaunchKernelGGL(segpln_fit_kernel<256>, dim3((unsigned)total[1]), dim3(256), 0, st0, pool.dargs.p, w1);
      STEREO_HIP_CHECK(hipEventRecord(ev[2], st0));
      if (total[2]) hipLaunchKernelGGL(segpln_fit_kernel<64>, dim3((unsigned)total[2]), dim3(64), 0, st0, pool.dargs.p, w2);
      STEREO_HIP_CHECK(hipGetLastError());
      STEREO_HIP_CHECK(hipEventRecord(ev[3], st0));
      STEREO_HIP_CHECK(hipEventSynchronize(ev[3]));
      float t[3];
      for (int k = 0; k < 3; ++k) STEREO_HIP_CHECK(hipEventElapsedTime(&t[k], ev[k], ev[k + 1]));
      std::fprintf(stderr, "[stereo_hip segpln] %d map(s): %zu large segments %.3f ms, %zu medium %.3f ms, %zu small %.3f ms\n", M, total[0], t[0], total[1], t[1],
                   total[2], t[2]);
      if (total[0]) {
        unsigned long long pr[16];
        STEREO_HIP_CHECK(hipMemcpy(pr, dprof.p, sizeof(pr), hipMemcpyDeviceToHost));
        std::fprintf(stderr, "[stereo_hip segpln]   largest segment, us (count): coordinates %.0f | draws %.0f (%llu) | batch passes %.0f (%llu) | bookkeeping %.0f | "
                     "least squares %.0f (%llu) | re-classify %.0f (%llu) | final fit %.0f | proposal %.0f\n", pr[0] / 100.0, pr[1] / 100.0, pr[9],
                     pr[2] / 100.0, pr[10], pr[3] / 100.0, pr[4] / 100.0, pr[12], pr[5] / 100.0, pr[13], pr[6] / 100.0, pr[7] / 100.0);
      }
      for (auto &e : ev) (void)hipEventDestroy(e);
    }
    for (int r = 0; r < M; ++r) {
      const int Sm = S[order[r]];
      MapScratch &sc = *pool.maps[r];
      if (Sm > 0) {
        STEREO_HIP_CHECK(hipMemcpyAsync(sc.hpl.p, sc.dpl.p, sizeof(double) * 3 * (size_t)Sm, hipMemcpyDeviceToHost, st0));
        STEREO_HIP_CHECK(hipMemcpyAsync(sc.hn.p, sc.dn.p, sizeof(int32_t) * (size_t)Sm, hipMemcpyDeviceToHost, st0));
      }
    }
    // The proposals (4 x N doubles per map, 75 MB for fourteen Teddy-sized maps) come back through pinned memory, a copy
    // and an event per map, and go to the caller's arrays on helper threads: a copy straight into pageable memory holds
    // the host for its length, and most of that length is first-touch page faults when the arrays are fresh -- which
    // fourteen threads take side by side (14 - 40 ms for the copies alone before).
    std::vector<std::future<bool>> delivered((size_t)M);
    struct JoinCopies {
      std::vector<std::future<bool>> &f;
      ~JoinCopies() { for (auto &x : f) if (x.valid()) x.wait(); }
    } join_copies{delivered};
    for (int r = 0; r < M; ++r) {
      const int m = order[r];
      if (!(proposals && proposals[m])) continue;
      MapScratch &sc = *pool.maps[r];
      if (sc.hprop.n < (size_t)4 * N) sc.hprop.alloc((size_t)4 * N);
      if (!sc.copied) STEREO_HIP_CHECK(hipEventCreateWithFlags(&sc.copied, hipEventDisableTiming));
      STEREO_HIP_CHECK(hipMemcpyAsync(sc.hprop.p, sc.dprop.p, sizeof(double) * 4 * N, hipMemcpyDeviceToHost, st0));
      STEREO_HIP_CHECK(hipEventRecord(sc.copied, st0));
      double *dst = proposals[m];
      const double *src = sc.hprop.p;
      const hipEvent_t ev = sc.copied;
      const size_t bytes = sizeof(double) * 4 * (size_t)N;
      delivered[r] = std::async(std::launch::async, [dst, src, ev, bytes, device]() -> bool {
        if (hipSetDevice(device) != hipSuccess || hipEventSynchronize(ev) != hipSuccess) return false;
        std::memcpy(dst, src, bytes);
        return true;
      });
    }
    STEREO_HIP_CHECK(hipEventRecord(pool.done[0], st0));
    STEREO_HIP_CHECK(hipEventSynchronize(pool.done[0]));
    for (auto &x : delivered)
      if (x.valid() && !x.get()) throw HipError{"stereo_segpln_planes: a proposal's copy back failed"};
    for (int r = 0; r < M; ++r) {
      const int m = order[r];
      const MapScratch &sc = *pool.maps[r];
      if (planes && planes[m] && S[m] > 0) std::memcpy(planes[m], sc.hpl.p, sizeof(double) * 3 * (size_t)S[m]);
      if (inliers && inliers[m] && S[m] > 0) std::memcpy(inliers[m], sc.hn.p, sizeof(int32_t) * (size_t)S[m]);
    }
  });
}

int stereo_segpln_planes(const double *wta, const int32_t *segments, int H, int W, double rt, uint64_t seed, int max_samples,
                         double *proposal, int S, double *planes, int32_t *inliers, char *err, size_t errcap) {
  if (!wta || !segments || H < 1 || W < 1 || S < 0 || max_samples < 1 || (!proposal && !planes))
    return fail("stereo_segpln_planes: bad argument", err, errcap);
  return stereo_segpln_planes_batch(wta, &segments, 1, H, W, rt, &seed, max_samples, &proposal, &S, &planes, &inliers, err, errcap);
}

}  // extern "C"
