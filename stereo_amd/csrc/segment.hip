// Image segmentation behind the globalstereo edge weights and the SegPln proposals: device stages
// and the C ABI.  Part of libstereo_hip.so.
//
//   stereo_segment_ms = vgg_segment_ms(A, h_s, h_r, min_sz)            imrender/vgg/vgg_segment_ms.cxx:18-87
//   stereo_segment_gb = vgg_segment_gb(A, sigma, k, min_sz, compress)  imrender/vgg/vgg_segment_gb.cxx:21-87
//
// Mean shift: the filter (msImageProcessor.cpp:3803-4303, what Filter(.., HIGH_SPEEDUP) runs) is the segmenter's cost --
// every pixel walks its own mean-shift trajectory, up to 100 window means over a lattice of buckets -- and it is one
// thread per pixel here.  The reference's "high speed-up" shortcuts make pixels depend on each other in scan order; they
// hang on msImageProcessor::speedThreshold, which neither the constructor (:64-108) nor the gateway sets -- the gateway
// reads an uninitialised float, in effect "the colours are EQUAL" (seg::kSpeedThreshold, segment_host.h).  So the
// shortcuts fire in flat image regions only.  The kernel walks every pixel's whole trajectory and reports, next to the
// pixel's own mode, whether another pixel's colour ever came within that threshold of it; the host stage
// (seg::ms_filter_finish) walks the flagged pixels again in scan order with the reference's bookkeeping (a few per cent
// of an image) and leaves everybody else their own mode.
// Every sum is made in the reference's order (27 neighbour buckets in its offset order, a bucket's points last inserted
// first; double precision, no contraction), so the filtered image equals the reference's bit for bit and the host
// stages behind it (segment_host.cpp) see the reference's input.
//
// Graph based: separable Gaussian smoothing and the four edge weights per pixel in single precision, in the
// reference's order of operations (convolve.h:29-45, segment-image.h:41-46); sorting and the union-find are the host's.
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdint>
#include <vector>

#include "../../include/stereo_hip.h"
#include "common.h"
#include "segment_host.h"

namespace stereo {
namespace {

struct MsArgs {
  const float *sdata;
  const int32_t *bucket_ptr, *bucket_items;
  int L, W, nb1, nb2;
  int neigh[27];
  float smin, sigmaS, sigmaR;
  double thr;        // seg::kSpeedThreshold (a single-precision denormal) as a double: no comparison hangs on the denormal mode
  float *out;        // L x 3: every pixel's own mode
  uint8_t *events;   // L: another pixel's colour came within thr of the trajectory
};

// One window mean (msImageProcessor.cpp:3955-4040 = :4130-4215): Mh = mean of the points within one bandwidth - yk.
__device__ __forceinline__ void ms_window(const MsArgs &a, const double (&yk)[5], double (&Mh)[5], double hiLTr, int self, int &event) {
  double wsum = 0;
#pragma unroll
  for (int j = 0; j < 5; ++j) Mh[j] = 0;
  const int c1 = (int)yk[0] + 1, c2 = (int)yk[1] + 1, c3 = (int)(yk[2] - a.smin) + 1;
  const int cb = c1 + a.nb1 * (c2 + a.nb2 * c3);
  for (int j = 0; j < 27; ++j) {
    const int b = cb + a.neigh[j];
    const int e0 = a.bucket_ptr[b], e1 = a.bucket_ptr[b + 1];
    for (int e = e0; e < e1; ++e) {
      const int d = a.bucket_items[e];
      const float *s = a.sdata + 5 * (size_t)d;
      const float s0 = s[0], s1 = s[1];
      double el = s0 - yk[0];
      double diff = el * el;
      el = s1 - yk[1];
      diff += el * el;
      if (diff < 1.0) {
        const float s2 = s[2], s3 = s[3], s4 = s[4];
        el = s2 - yk[2];
        diff = yk[2] > hiLTr ? 4 * el * el : el * el;
        el = s3 - yk[3];
        diff += el * el;
        el = s4 - yk[4];
        diff += el * el;
        if (diff < 1.0) {   // (weight 1 - weightMap = 1: no weight map without the gateway's fifth argument)
          Mh[0] += 1.0 * s0; Mh[1] += 1.0 * s1; Mh[2] += 1.0 * s2; Mh[3] += 1.0 * s3; Mh[4] += 1.0 * s4;
          wsum += 1.0;
          event |= (diff < a.thr && d != self) ? 1 : 0;   // (:4015-4022: such a point would join this pixel's basin)
        }
      }
    }
  }
  if (wsum > 0) {
#pragma unroll
    for (int j = 0; j < 5; ++j) Mh[j] = Mh[j] / wsum - yk[j];
  } else {
#pragma unroll
    for (int j = 0; j < 5; ++j) Mh[j] = 0;
  }
}

__global__ __launch_bounds__(64) void ms_filter_kernel(MsArgs a) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= a.L) return;
  const double hiLTr = 80.0 / a.sigmaR;
  double yk[5], Mh[5];
#pragma unroll
  for (int j = 0; j < 5; ++j) yk[j] = a.sdata[5 * (size_t)i + j];
  int event = 0;
  ms_window(a, yk, Mh, hiLTr, i, event);
  double mv = (Mh[0] * Mh[0] + Mh[1] * Mh[1]) * a.sigmaS * a.sigmaS;
  mv += (Mh[2] * Mh[2] + Mh[3] * Mh[3] + Mh[4] * Mh[4]) * a.sigmaR * a.sigmaR;
  int iter = 1;
  while (mv >= 0.01 && iter < 100) {   // EPSILON, LIMIT (ms.h:106,111)
#pragma unroll
    for (int j = 0; j < 5; ++j) yk[j] += Mh[j];
    {
      // :4085-4100: the pixel under the trajectory's rounded position
      const int cx = (int)(a.sigmaS * yk[0] + 0.5), cy = (int)(a.sigmaS * yk[1] + 0.5);
      const int ci = cy * a.W + cx;
      if (ci != i) {
        const float *s = a.sdata + 5 * (size_t)ci;
        double diff = 0;
#pragma unroll
        for (int k = 2; k < 5; ++k) { const double el = s[k] - yk[k]; diff += el * el; }
        event |= diff < a.thr ? 1 : 0;
      }
    }
    ms_window(a, yk, Mh, hiLTr, i, event);
    mv = (Mh[0] * Mh[0] + Mh[1] * Mh[1]) * a.sigmaS * a.sigmaS;
    mv += (Mh[2] * Mh[2] + Mh[3] * Mh[3] + Mh[4] * Mh[4]) * a.sigmaR * a.sigmaR;
    ++iter;
  }
#pragma unroll
  for (int j = 0; j < 5; ++j) yk[j] += Mh[j];
#pragma unroll
  for (int k = 0; k < 3; ++k) a.out[3 * (size_t)i + k] = (float)(yk[k + 2] * a.sigmaR);
  a.events[i] = (uint8_t)event;
}

// ---- graph based ---------------------------------------------------------------------------------------------------------
// One pass of convolve_even (convolve.h:29-45) along x for the three channels; the second pass runs along y on the
// result (the reference transposes instead).  src / dst: 3 planes of H x W floats, row-major.
__global__ void gb_convolve_kernel(const float *src, float *dst, int H, int W, const float *mask, int len, int along_y) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= H * W) return;
  const int y = i / W, x = i % W;
  for (int c = 0; c < 3; ++c) {
    const float *p = src + (size_t)c * H * W;
    float sum = mask[0] * p[i];
    for (int k = 1; k < len; ++k) {
      float lo, hi;
      if (along_y) { lo = p[(size_t)max(y - k, 0) * W + x]; hi = p[(size_t)min(y + k, H - 1) * W + x]; }
      else { lo = p[(size_t)y * W + max(x - k, 0)]; hi = p[(size_t)y * W + min(x + k, W - 1)]; }
      sum += mask[k] * (lo + hi);
    }
    dst[(size_t)c * H * W + i] = sum;
  }
}

// diff() of segment-image.h:41-46 for the four edges of a pixel (right, down, down-right, up-right).
__global__ void gb_weights_kernel(const float *sm, int H, int W, float *weights) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= H * W) return;
  const int y = i / W, x = i % W;
  const float *r = sm, *g = sm + (size_t)H * W, *b = sm + 2 * (size_t)H * W;
  const int dx[4] = {1, 0, 1, 1}, dy[4] = {0, 1, 1, -1};
  for (int t = 0; t < 4; ++t) {
    const int x2 = x + dx[t], y2 = y + dy[t];
    float w = 0;
    if (x2 < W && y2 >= 0 && y2 < H) {
      const int j = y2 * W + x2;
      const float dr = r[i] - r[j], dg = g[i] - g[j], db = b[i] - b[j];
      const float s = dr * dr + dg * dg + db * db;
      w = (float)__dsqrt_rn((double)s);   // (the reference's sqrt is the double one, its result stored as float)
    }
    weights[4 * (size_t)i + t] = w;
  }
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

void ms_own_device(const seg::MsLattice &lat, float *own, uint8_t *events) {
  const int L = lat.H * lat.W;
  DevBuf<float> ds, dout;
  DevBuf<int32_t> dptr, ditems;
  DevBuf<uint8_t> dev;
  dev.alloc((size_t)L);
  ds.upload(lat.sdata.data(), lat.sdata.size());
  dptr.upload(lat.bucket_ptr.data(), lat.bucket_ptr.size());
  ditems.upload(lat.bucket_items.data(), lat.bucket_items.size());
  dout.alloc((size_t)3 * L);
  MsArgs a{};
  a.sdata = ds.p; a.bucket_ptr = dptr.p; a.bucket_items = ditems.p;
  a.L = L; a.W = lat.W; a.nb1 = lat.nb1; a.nb2 = lat.nb2;
  for (int j = 0; j < 27; ++j) a.neigh[j] = lat.neigh[j];
  a.smin = lat.smin; a.sigmaS = lat.sigmaS; a.sigmaR = lat.sigmaR; a.thr = (double)seg::kSpeedThreshold;
  a.out = dout.p; a.events = dev.p;
  hipLaunchKernelGGL(ms_filter_kernel, dim3((unsigned)((L + 63) / 64)), dim3(64), 0, 0, a);
  STEREO_HIP_CHECK(hipGetLastError());
  STEREO_HIP_CHECK(hipMemcpy(own, dout.p, sizeof(float) * 3 * (size_t)L, hipMemcpyDeviceToHost));
  STEREO_HIP_CHECK(hipMemcpy(events, dev.p, (size_t)L, hipMemcpyDeviceToHost));
}

void ms_filter_device(const seg::MsLattice &lat, float *filtered) {
  const size_t L = (size_t)lat.H * lat.W;
  std::vector<float> own(3 * L);
  std::vector<uint8_t> events(L);
  ms_own_device(lat, own.data(), events.data());
  seg::ms_filter_finish(lat, own.data(), events.data(), seg::kSpeedThreshold, filtered);
}

}  // namespace
}  // namespace stereo

using namespace stereo;

extern "C" {

int stereo_segment_ms(const uint8_t *A, int H, int W, double h_s, double h_r, double min_sz, uint32_t *out, char *err,
                      size_t errcap) {
  if (!A || !out || H < 1 || W < 1 || (int64_t)H * W >= (int64_t)1 << 28) return fail("stereo_segment_ms: bad argument", err, errcap);
  // vgg_segment_ms.cxx:34-36: the scalars as the gateway casts them
  const int sigmaS = (int)h_s;
  const float sigmaR = (float)h_r;
  const int min_region = (int)min_sz;
  if (sigmaS <= 0 || !(sigmaR > 0))   // msImageProcessor.cpp:3820-3824
    return fail("stereo_segment_ms: sigmaS and/or sigmaR is zero or negative.", err, errcap);
  return guarded("stereo_segment_ms", err, errcap, [&] {
    const size_t L = (size_t)H * W;
    std::vector<float> luv(3 * L), filtered(3 * L);
    seg::rgb_to_luv(A, H, W, luv.data());
    seg::MsLattice lat;
    seg::ms_lattice(luv.data(), H, W, sigmaS, sigmaR, lat);
    ms_filter_device(lat, filtered.data());
    std::vector<int32_t> labels(L);
    seg::ms_regions(filtered.data(), H, W, sigmaR, min_region, labels.data());
    for (int y = 0; y < H; ++y)   // vgg_segment_ms.cxx:79-84: row-major labels + 1 -> H x W column-major
      for (int x = 0; x < W; ++x) out[(size_t)x * H + y] = (uint32_t)labels[(size_t)y * W + x] + 1;
  });
}

int stereo_segment_ms_filter(const uint8_t *A, int H, int W, double h_s, double h_r, float *filtered, char *err, size_t errcap) {
  if (!A || !filtered || H < 1 || W < 1 || (int64_t)H * W >= (int64_t)1 << 28) return fail("stereo_segment_ms_filter: bad argument", err, errcap);
  const int sigmaS = (int)h_s;
  const float sigmaR = (float)h_r;
  if (sigmaS <= 0 || !(sigmaR > 0)) return fail("stereo_segment_ms_filter: sigmaS and/or sigmaR is zero or negative.", err, errcap);
  return guarded("stereo_segment_ms_filter", err, errcap, [&] {
    std::vector<float> luv(3 * (size_t)H * W);
    seg::rgb_to_luv(A, H, W, luv.data());
    seg::MsLattice lat;
    seg::ms_lattice(luv.data(), H, W, sigmaS, sigmaR, lat);
    ms_filter_device(lat, filtered);
  });
}

int stereo_segment_ms_own(const uint8_t *A, int H, int W, double h_s, double h_r, float *own, uint8_t *events, char *err,
                          size_t errcap) {
  if (!A || !own || !events || H < 1 || W < 1 || (int64_t)H * W >= (int64_t)1 << 28) return fail("stereo_segment_ms_own: bad argument", err, errcap);
  const int sigmaS = (int)h_s;
  const float sigmaR = (float)h_r;
  if (sigmaS <= 0 || !(sigmaR > 0)) return fail("stereo_segment_ms_own: sigmaS and/or sigmaR is zero or negative.", err, errcap);
  return guarded("stereo_segment_ms_own", err, errcap, [&] {
    std::vector<float> luv(3 * (size_t)H * W);
    seg::rgb_to_luv(A, H, W, luv.data());
    seg::MsLattice lat;
    seg::ms_lattice(luv.data(), H, W, sigmaS, sigmaR, lat);
    ms_own_device(lat, own, events);
  });
}

int stereo_segment_ms_finish(const uint8_t *A, int H, int W, double h_s, double h_r, const float *own, const uint8_t *events,
                             float *filtered, int64_t *walked, char *err, size_t errcap) {
  if (!A || !own || !events || !filtered || H < 1 || W < 1) return fail("stereo_segment_ms_finish: bad argument", err, errcap);
  const int sigmaS = (int)h_s;
  const float sigmaR = (float)h_r;
  if (sigmaS <= 0 || !(sigmaR > 0)) return fail("stereo_segment_ms_finish: sigmaS and/or sigmaR is zero or negative.", err, errcap);
  try {
    std::vector<float> luv(3 * (size_t)H * W);
    seg::rgb_to_luv(A, H, W, luv.data());
    seg::MsLattice lat;
    seg::ms_lattice(luv.data(), H, W, sigmaS, sigmaR, lat);
    const int64_t n = seg::ms_filter_finish(lat, own, events, seg::kSpeedThreshold, filtered);
    if (walked) *walked = n;
    return 0;
  } catch (const std::exception &e) {
    return fail(std::string("stereo_segment_ms_finish: ") + e.what(), err, errcap);
  }
}

int stereo_segment_ms_regions(const float *filtered, int H, int W, double h_r, double min_sz, uint32_t *out, char *err,
                              size_t errcap) {
  if (!filtered || !out || H < 1 || W < 1) return fail("stereo_segment_ms_regions: bad argument", err, errcap);
  try {
    std::vector<int32_t> labels((size_t)H * W);
    seg::ms_regions(filtered, H, W, (float)h_r, (int)min_sz, labels.data());
    for (int y = 0; y < H; ++y)
      for (int x = 0; x < W; ++x) out[(size_t)x * H + y] = (uint32_t)labels[(size_t)y * W + x] + 1;
    return 0;
  } catch (const std::exception &e) {
    return fail(std::string("stereo_segment_ms_regions: ") + e.what(), err, errcap);
  }
}

int stereo_segment_ms_luv(const uint8_t *A, int H, int W, float *luv, char *err, size_t errcap) {
  if (!A || !luv || H < 1 || W < 1) return fail("stereo_segment_ms_luv: bad argument", err, errcap);
  seg::rgb_to_luv(A, H, W, luv);
  return 0;
}

int stereo_segment_gb_weights(const uint8_t *A, int H, int W, double sigma, float *weights, char *err, size_t errcap) {
  if (!A || !weights || H < 1 || W < 1 || (int64_t)H * W >= (int64_t)1 << 28) return fail("stereo_segment_gb_weights: bad argument", err, errcap);
  return guarded("stereo_segment_gb_weights", err, errcap, [&] {
    const size_t L = (size_t)H * W;
    // vgg_segment_gb.cxx:41-50 + segment-image.h:190-196: the three channels as float planes, row-major
    std::vector<float> planes(3 * L);
    for (int c = 0; c < 3; ++c)
      for (int y = 0; y < H; ++y)
        for (int x = 0; x < W; ++x) planes[c * L + (size_t)y * W + x] = (float)A[c * L + (size_t)x * H + y];
    const std::vector<float> mask = seg::gb_mask((float)sigma);
    DevBuf<float> d0, d1, dm, dw;
    d0.upload(planes.data(), planes.size()); d1.alloc(3 * L); dm.upload(mask.data(), mask.size()); dw.alloc(4 * L);
    const dim3 grid((unsigned)((L + 255) / 256)), block(256);
    hipLaunchKernelGGL(gb_convolve_kernel, grid, block, 0, 0, d0.p, d1.p, H, W, dm.p, (int)mask.size(), 0);
    hipLaunchKernelGGL(gb_convolve_kernel, grid, block, 0, 0, d1.p, d0.p, H, W, dm.p, (int)mask.size(), 1);
    hipLaunchKernelGGL(gb_weights_kernel, grid, block, 0, 0, d0.p, H, W, dw.p);
    STEREO_HIP_CHECK(hipGetLastError());
    STEREO_HIP_CHECK(hipMemcpy(weights, dw.p, sizeof(float) * 4 * L, hipMemcpyDeviceToHost));
  });
}

int stereo_segment_gb_regions(const float *weights, int H, int W, double k, double min_sz, int compress, uint32_t *out,
                              char *err, size_t errcap) {
  if (!weights || !out || H < 1 || W < 1) return fail("stereo_segment_gb_regions: bad argument", err, errcap);
  try {
    seg::gb_regions(weights, H, W, (float)k, (int)min_sz, compress, out);
    return 0;
  } catch (const std::exception &e) {
    return fail(std::string("stereo_segment_gb_regions: ") + e.what(), err, errcap);
  }
}

int stereo_segment_gb(const uint8_t *A, int H, int W, double sigma, double k, double min_sz, int compress, uint32_t *out,
                      char *err, size_t errcap) {
  if (!A || !out || H < 1 || W < 1) return fail("stereo_segment_gb: bad argument", err, errcap);
  std::vector<float> weights(4 * (size_t)H * W);
  if (int rc = stereo_segment_gb_weights(A, H, W, sigma, weights.data(), err, errcap)) return rc;
  return stereo_segment_gb_regions(weights.data(), H, W, k, min_sz, compress, out, err, errcap);
}

}  // extern "C"
