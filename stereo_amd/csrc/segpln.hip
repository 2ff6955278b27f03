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
  uint8_t *cur, *inl;       // scratch: inlier flags of the current trial / of the best one
  double *proposal;         // 4 x N
  double *planes;           // 3 x S
  int32_t *ninl;            // S
  const int32_t *list;      // the segments of this launch
};

// One workgroup of T threads per segment (dispmap_globalstereo.m:164-191 + rplane :417-450).  Control flow is uniform:
// every decision is made on a count or on values every thread holds.  The cost of a segment is its RANSAC trials, each
// a pass over all of its points (`classify`: flags and an integer count, independent of who looks at which point) --
// that pass is spread over all T threads (T = 1024 for the large segments of a coarse segmentation map: one of them on
// ONE wave used to take 58 ms of the Teddy example's 119 ms of plane fitting); the least-squares sums, whose order of
// addition is part of the definition (oracle/terms.py:_lstsq3: 64 strided partial sums, then a tree), stay with the
// first wave, which hands the plane to the others through LDS.  They run a few times per segment, not once per trial.
template <int T>
__global__ __launch_bounds__(T) void segpln_fit_kernel(FitArgs a) {
  __shared__ int s_cnt[T / 64 > 0 ? T / 64 : 1];
  __shared__ double s_plane[3];
  const int s = a.list[blockIdx.x], tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int p0 = a.seg_ptr[s], p1 = a.seg_ptr[s + 1];
  if (p1 <= p0) { if (tid == 0) { a.ninl[s] = 0; a.planes[3 * s] = a.planes[3 * s + 1] = a.planes[3 * s + 2] = 0; } return; }
  double *X = a.px + p0, *Y = a.py + p0, *Z = a.pz + p0;
  uint8_t *cur = a.cur + p0, *inl = a.inl + p0;
  // world coordinates [x y 1] / d of the segment's pixels (:141-145), those with WC(:,3) ~= 0 kept (:168), in order
  int n = 0;
  if (wave == 0)
  for (int base = p0; base < p1; base += 64) {
    const int i = base + lane;
    double z = 0, x = 0, y = 0;
    bool keep = false;
    if (i < p1) {
      const int px = a.seg_idx[i];
      z = 1.0 / a.wta[px];
      x = z * (double)(px / a.H + 1); y = z * (double)(px % a.H + 1);
      keep = z != 0;
    }
    const unsigned long long m = __builtin_amdgcn_ballot_w64(keep);
    const int at = n + __builtin_popcountll(m & ((1ull << lane) - 1));
    if (keep) { X[at] = x; Y[at] = y; Z[at] = z; }
    n += __builtin_popcountll(m);
  }
  if (T > 64) {
    if (tid == 0) s_cnt[0] = n;
    __threadfence_block();
    __syncthreads();
    n = s_cnt[0];
  }
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
  __syncthreads();
  // distances of all points to the plane N . p = -1, flags into `dst`; returns the number of inliers
  auto classify = [&](const double N[3], uint8_t *dst) {
    int cnt = 0;
    for (int i = tid; i < n; i += T) {
      const double dist = fabs(((X[i] * N[0] + Y[i] * N[1]) + Z[i] * N[2]) + 1.0);
      const bool v = dist < a.rt;
      dst[i] = v ? 1 : 0;
      cnt += v ? 1 : 0;
    }
    for (int o = 32; o > 0; o >>= 1) cnt += __shfl_xor(cnt, o, 64);
    if (T > 64) {
      __syncthreads();   // (the last reader of s_cnt is done)
      if (lane == 0) s_cnt[wave] = cnt;
      __syncthreads();
      cnt = 0;
      for (int w = 0; w < T / 64; ++w) cnt += s_cnt[w];
    }
    __syncthreads();
    return cnt;
  };
  // least squares of the flagged points by the normal equations, summed as oracle/terms.py:_lstsq3 sums them
  auto lstsq = [&](const uint8_t *flags, double N[3]) {
    if (T > 64 && wave != 0) {   // (the first wave sums, in the definition's order; the others take the plane from LDS)
      __syncthreads();
      N[0] = s_plane[0]; N[1] = s_plane[1]; N[2] = s_plane[2];
      __syncthreads();
      return;
    }
    double acc[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
    for (int i = lane; i < n; i += 64) {
      if (flags == nullptr || flags[i]) {
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
    double m[9], b[3];
    for (int k = 0; k < 9; ++k) acc[k] = __shfl(acc[k], 0, 64);
    m[0] = acc[0]; m[1] = acc[1]; m[2] = acc[2]; m[3] = acc[1]; m[4] = acc[3]; m[5] = acc[4]; m[6] = acc[2]; m[7] = acc[4]; m[8] = acc[5];
    b[0] = acc[6]; b[1] = acc[7]; b[2] = acc[8];
    solve3(m, b, N);
    if (T > 64) {
      if (lane == 0) { s_plane[0] = N[0]; s_plane[1] = N[1]; s_plane[2] = N[2]; }
      __syncthreads();
      __syncthreads();
    }
  };
  int n_in = n;          // local_WC_points = N when there are too few points for RANSAC (:170)
  bool use_flags = false;
  if (n > 3) {
    int max_i = 3, no_sam = 0, best = 0;
    double max_sam = (double)a.max_samples;
    for (int i = tid; i < n; i += T) inl[i] = 0;
    __syncthreads();
    while ((double)no_sam < max_sam) {
      ++no_sam;
      int sam[3], got = 0, attempt = 0;
      while (got < 3) {   // oracle/terms.py:segpln_sample
        const uint64_t key = a.seed * 0x100000001B3ull + (uint64_t)(s + 1) * 0x1000193ull + (uint64_t)no_sam * 64ull + (uint64_t)attempt;
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
      const int no_i = classify(N, cur);
      if (max_i < no_i) {
        lstsq(cur, N);   // re-estimate plane and inliers (:437-440)
        __syncthreads();
        const int cnt = classify(N, cur);
        if (cnt > best) {
          for (int i = tid; i < n; i += T) inl[i] = cur[i];
          __syncthreads();
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
    n_in = best;
    use_flags = true;
  }
  double N_[3] = {0, 0, 0};
  const bool fitted = n_in > 2;
  if (fitted) lstsq(use_flags ? inl : nullptr, N_);
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

int stereo_segpln_planes(const double *wta, const int32_t *segments, int H, int W, double rt, uint64_t seed, int max_samples,
                         double *proposal, int S, double *planes, int32_t *inliers, char *err, size_t errcap) {
  if (!wta || !segments || !proposal || H < 1 || W < 1 || S < 0 || max_samples < 1)
    return fail("stereo_segpln_planes: bad argument", err, errcap);
  const int64_t N = (int64_t)H * W;
  for (int64_t i = 0; i < N; ++i)
    if (segments[i] < 0 || segments[i] > S) return fail("stereo_segpln_planes: segment label out of range [0, S]", err, errcap);
  return guarded("stereo_segpln_planes", err, errcap, [&] {
    // pixels grouped by segment, ascending pixel id inside a segment (= MATLAB's logical indexing order); label 0 = none
    std::vector<int32_t> ptr((size_t)S + 2, 0), idx((size_t)N);
    for (int64_t i = 0; i < N; ++i) ++ptr[(size_t)segments[i] + 1];
    for (int s = 0; s <= S; ++s) ptr[s + 1] += ptr[s];
    std::vector<int32_t> at(ptr.begin(), ptr.end() - 1);
    for (int64_t i = 0; i < N; ++i) idx[(size_t)at[segments[i]]++] = (int32_t)i;
    // (ptr[1 .. S + 1] delimit segments 1 .. S; the pixels of label 0 sit in front)
    DevBuf<double> dw, px, py, pz, dprop, dpl;
    DevBuf<int32_t> dptr, didx, dn;
    DevBuf<uint8_t> cur, inl;
    dw.upload(wta, N); dptr.upload(ptr.data() + 1, (size_t)S + 1); didx.upload(idx.data(), N);
    px.alloc(N); py.alloc(N); pz.alloc(N); cur.alloc(N); inl.alloc(N); dprop.alloc(4 * N);
    dpl.alloc((size_t)3 * std::max(S, 1)); dn.alloc((size_t)std::max(S, 1));
    hipLaunchKernelGGL(segpln_init_kernel, dim3((unsigned)((N + 255) / 256)), dim3(256), 0, 0, N, dprop.p);
    if (S > 0) {
      // segments by size: a large one gets a workgroup of 1024 threads for its passes over the points, the others a wave
      constexpr int kLargeSegment = 4096;
      std::vector<int32_t> list[2];
      for (int sg = 0; sg < S; ++sg) list[ptr[sg + 2] - ptr[sg + 1] > kLargeSegment ? 1 : 0].push_back(sg);
      std::vector<int32_t> both(list[1]);
      both.insert(both.end(), list[0].begin(), list[0].end());
      DevBuf<int32_t> dlist;
      dlist.upload(both.data(), both.size());
      FitArgs a{dw.p, dptr.p, didx.p, H, W, S, max_samples, rt, seed, px.p, py.p, pz.p, cur.p, inl.p, dprop.p, dpl.p, dn.p, dlist.p};
      if (!list[1].empty()) hipLaunchKernelGGL(segpln_fit_kernel<1024>, dim3((unsigned)list[1].size()), dim3(1024), 0, 0, a);
      a.list = dlist.p + list[1].size();
      if (!list[0].empty()) hipLaunchKernelGGL(segpln_fit_kernel<64>, dim3((unsigned)list[0].size()), dim3(64), 0, 0, a);
      STEREO_HIP_CHECK(hipGetLastError());
      STEREO_HIP_CHECK(hipDeviceSynchronize());   // (dlist lives until here)
    }
    STEREO_HIP_CHECK(hipGetLastError());
    STEREO_HIP_CHECK(hipMemcpy(proposal, dprop.p, sizeof(double) * 4 * N, hipMemcpyDeviceToHost));
    if (planes && S > 0) STEREO_HIP_CHECK(hipMemcpy(planes, dpl.p, sizeof(double) * 3 * S, hipMemcpyDeviceToHost));
    if (inliers && S > 0) STEREO_HIP_CHECK(hipMemcpy(inliers, dn.p, sizeof(int32_t) * S, hipMemcpyDeviceToHost));
  });
}

}  // extern "C"
