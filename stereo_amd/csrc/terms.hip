// Cost-volume and term-builder kernels (MI355X / gfx950) + their C ABI.
//
// Device versions of the MATLAB array math that feeds the two solvers:
//   dispmap_super.m:318-328   plane -> disparity
//   dispmap_super.m:226-262   truncated |p-q|^k pairwise terms (E00, E01, E10, E11)
//   dispmap_super.m:177-183   q / qprim of K proposals for TRW-S
//   dispmap_ncc.m:116-198     5x5xRGB normalised cross-correlation volume
//   dispmap_ncc.m:107-115,222-276  parabola sampling of the volume -> NCC unary
//   dispmap_ncc.m:208-221     winner-takes-all initial disparity
//   dispmap_globalstereo.m:336-345,355-375,405 + imrender/vgg/vgg_interp2.cxx:245-322
//                             photo-consistency unary (warp, bilinear gather, ephoto)
// All are independent per pixel / per edge: one thread each, coalesced along the
// column-major pixel index (MATLAB layout), images served from L2.  HBM-bound
// streaming kernels; no contraction (-ffp-contract=off) so that + - * / match numpy.
#include <hip/hip_runtime.h>

#include <memory>

#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#include <string>
#include <vector>

#include "../../include/stereo_hip.h"
#include "common.h"

namespace stereo {
namespace {

constexpr int kTB = 256;

__device__ __forceinline__ double plane_disp(const double *pl, double x, double y) {
  // -(sum(assignment(1:2,:).*points) + assignment(4,:)) ./ assignment(3,:)
  const double s = pl[0] * x + pl[1] * y;
  return -(s + pl[3]) / pl[2];
}

__device__ __forceinline__ double pair_term(int kernel, double w, double p, double q, double tol) {
  const double d = p - q;
  const double v = kernel == 1 ? fabs(d) : d * d;
  return w * (v < tol ? v : tol);
}

__global__ __launch_bounds__(kTB) void pairwise_terms_kernel(int kernel, int64_t E, const uint32_t *conn,
                                                            const double *points, const double *cur,
                                                            const double *prop, const double *w, double tol,
                                                            double d_min, double d_step, double *E00,
                                                            double *E01, double *E10, double *E11) {
  const int64_t e = (int64_t)blockIdx.x * kTB + threadIdx.x;
  if (e >= E) return;
  const uint32_t i1 = conn[2 * e], i2 = conn[2 * e + 1];
  const double x = points[2 * (size_t)i2], y = points[2 * (size_t)i2 + 1];
  double q = plane_disp(cur + 4 * (size_t)i2, x, y), qp = plane_disp(cur + 4 * (size_t)i1, x, y);
  if (d_step != 0) { q = (q - d_min) / d_step; qp = (qp - d_min) / d_step; }
  const double we = w[e];
  E00[e] = pair_term(kernel, we, q, qp, tol);
  if (prop) {
    double nq = plane_disp(prop + 4 * (size_t)i2, x, y), nqp = plane_disp(prop + 4 * (size_t)i1, x, y);
    if (d_step != 0) { nq = (nq - d_min) / d_step; nqp = (nqp - d_min) / d_step; }
    E11[e] = pair_term(kernel, we, nq, nqp, tol);
    E10[e] = pair_term(kernel, we, q, nqp, tol);   // dispmap_super.m:256
    E01[e] = pair_term(kernel, we, nq, qp, tol);   // dispmap_super.m:259
  }
}

__global__ __launch_bounds__(kTB) void trws_positions_kernel(int64_t E, int K, int64_t N, const uint32_t *conn,
                                                            const double *points, const double *props,
                                                            double d_min, double d_step, double *q,
                                                            double *qprim) {
  const int64_t t = (int64_t)blockIdx.x * kTB + threadIdx.x;
  if (t >= E * K) return;
  const int64_t e = t / K;
  const int k = (int)(t % K);
  const uint32_t i1 = conn[2 * e], i2 = conn[2 * e + 1];
  const double x = points[2 * (size_t)i2], y = points[2 * (size_t)i2 + 1];
  const double *P = props + (size_t)k * 4 * N;
  double a = plane_disp(P + 4 * (size_t)i2, x, y), b = plane_disp(P + 4 * (size_t)i1, x, y);
  if (d_step != 0) { a = (a - d_min) / d_step; b = (b - d_min) / d_step; }
  q[t] = a;       // K x E column major: label fastest
  qprim[t] = b;
}

// ---- NCC volume -----------------------------------------------------------

struct NccParams {
  int H, W, D, r;
  const double *im0, *im1;  // H x W x 3 column major
  const double *disp;       // D
  double *out;
  int layout;               // 0: H x W x D (MATLAB), 1: D x N label fastest
};

// right image shifted by d (dispmap_ncc.m:145-154), zero outside the filled columns
__device__ __forceinline__ double shifted(const NccParams &p, int row, int col, int ch, double d, int c0, int n,
                                          double step) {
  if (row < 0 || row >= p.H || col < 0 || col >= p.W) return 0.0;  // conv2 'same' zero padding
  if (col + 1 < c0) return 0.0;
  const int i = col + 1 - c0;
  const double X = n > 1 ? 1.0 + i * step : 1.0;  // linspace(1, W-d, n)
  int x0 = (int)floor(X);
  if (x0 < 1) x0 = 1;
  if (p.W > 1 && x0 > p.W - 1) x0 = p.W - 1;
  const double t = X - x0;
  const double *plane = p.im1 + (size_t)ch * p.H * p.W;
  const double left = plane[(size_t)(x0 - 1) * p.H + row];
  if (t == 0) return left;
  const int xr = x0 < p.W - 1 ? x0 : p.W - 1;
  const double right = plane[(size_t)xr * p.H + row];
  return left * (1 - t) + right * t;
}

__device__ __forceinline__ double ref_px(const NccParams &p, int row, int col, int ch) {
  if (row < 0 || row >= p.H || col < 0 || col >= p.W) return 0.0;
  return p.im0[(size_t)ch * p.H * p.W + (size_t)col * p.H + row];
}

__global__ __launch_bounds__(kTB) void ncc_volume_kernel(NccParams p) {
  const int64_t t = (int64_t)blockIdx.x * kTB + threadIdx.x;
  const int64_t Npx = (int64_t)p.H * p.W;
  if (t >= Npx * p.D) return;
  const int di = (int)(t / Npx);
  const int64_t px = t % Npx;
  const int row = (int)(px % p.H), col = (int)(px / p.H);
  const double d = p.disp[di];
  const int c0 = (int)ceil(d + 1);
  const int n = p.W - c0 + 1;
  const double step = n > 1 ? ((p.W - d) - 1.0) / (n - 1) : 0.0;
  const int r = p.r;
  const double npatch = (double)((2 * r + 1) * (2 * r + 1));
  const double mscale = 1.0 / npatch / 3.0;
  double bL[3] = {0, 0, 0}, bT[3] = {0, 0, 0}, bLL[3] = {0, 0, 0}, bTT[3] = {0, 0, 0}, bLT[3] = {0, 0, 0};
  for (int dx = -r; dx <= r; ++dx)
    for (int dy = -r; dy <= r; ++dy)
      for (int ch = 0; ch < 3; ++ch) {
        const double L = ref_px(p, row + dy, col + dx, ch);
        const double T = shifted(p, row + dy, col + dx, ch, d, c0, n, step);
        bL[ch] += L; bT[ch] += T; bLL[ch] += L * L; bTT[ch] += T * T; bLT[ch] += L * T;
      }
  const double meanL = (bL[0] * mscale + bL[1] * mscale) + bL[2] * mscale;
  const double meanT = (bT[0] * mscale + bT[1] * mscale) + bT[2] * mscale;
  const double t1L = (bLL[0] + bLL[1]) + bLL[2];
  const double t2L = (meanL * bL[0] + meanL * bL[1]) + meanL * bL[2];
  const double varL = t1L - 2 * t2L + npatch * 3 * meanL * meanL;
  const double u1 = (bTT[0] + bTT[1]) + bTT[2];
  const double u2 = (meanT * bT[0] + meanT * bT[1]) + meanT * bT[2];
  const double varT = u1 - 2 * u2 + npatch * 3 * meanT * meanT;
  const double c1 = (bLT[0] + bLT[1]) + bLT[2];
  const double c2 = (meanL * bT[0] + meanL * bT[1]) + meanL * bT[2];
  const double c3 = (meanT * bL[0] + meanT * bL[1]) + meanT * bL[2];
  const double num = c1 - c2 - c3 + npatch * 3 * meanT * meanL;
  // sqrt of a negative variance is imaginary in MATLAB; real(num/normL/normT) follows
  // (dispmap_ncc.m:141,171,188-192): one imaginary norm -> 0, both -> -num/(sL*sT)
  double val;
  const double sL = sqrt(fabs(varL)), sT = sqrt(fabs(varT));
  if (varL >= 0 && varT >= 0) val = num / sL / sT;
  else if (varL < 0 && varT < 0) val = -(num / sL / sT);
  else val = 0.0;
  if (!isfinite(val)) val = 0.0;
  const int cmask = (int)floor(d + 1 + 0.5);  // bnd_im(:, round(d+1):end) = 1, MATLAB round()
  if (col + 1 < cmask) val = 0.0;
  if (p.layout == 0) p.out[(size_t)di * Npx + px] = val;
  else p.out[(size_t)px * p.D + di] = val;
}

// ---- NCC volume, integer disparities: staged statistics + separable cross term ---------------
// With an integer disparity d the shifted image of dispmap_ncc.m:145-154 is a pure shift,
// imtr(row, col) = im1(row, col - d) (zero for col < d), so the d-independent box sums of im1 and
// im1^2 serve every slice: box5(imtr)(row, col) = box5(im1)(row, col - d) wherever the 5 x 5 window
// stays inside the image horizontally (the last `r` columns differ: conv2's zero padding cuts the
// window of imtr at column W, the one of im1 only at column W + d; they are summed directly).  One
// pre-pass per image stores per pixel the three per-channel box sums, the mean and the variance term
// (ncc_stats_kernel: 25 taps x 3 channels in the order of ncc_volume_kernel, d-independent).  What is
// left per (pixel, d) is the cross term sum_window sum_ch im0 * imtr -- a sliding-window sum with no
// contraction index shared between outputs (no MFMA) -- computed separably: lane = image row, a wave
// walks along the columns for one disparity with the five column products of the window in
// registers, the vertical five-tap sum comes from the neighbouring lanes (DPP wave shifts).  All
// image and statistics reads are coalesced 512-byte rows served by L2; the volume is written once,
// coalesced in both layouts (the label-fastest layout through an 8-disparity LDS transpose: 64-byte runs).
// Association: products (c0 + c1) + c2, then columns left to right, then rows top to bottom; the
// per-channel 25-tap order of the reference's conv2 is NOT kept for the cross term (agreement with
// the NumPy restatement to 1e-12 is the bar, tests/test_terms_gpu.py).
constexpr int kNccRows = 60;    // output rows per wave (64 lanes = 60 + 2 x 2 halo rows)
constexpr int kNccWaves = 8;    // disparities per workgroup (a 1024-thread workgroup would cap the wave at 128 VGPRs: spills)
constexpr int kNccCols = 32;    // columns per workgroup

struct NccFast {
  int H, W, D;
  const double *im0, *im1;
  const double *st0, *st1;   // [5][N]: box sums of the three channels, mean, signed reciprocal norm
  const double *disp;
  double *out;
  int layout;
};

__global__ __launch_bounds__(kTB) void ncc_stats_kernel(const double *im, int H, int W, int r, double *st, double *stT = nullptr) {
  const int64_t px = (int64_t)blockIdx.x * kTB + threadIdx.x;
  const int64_t Npx = (int64_t)H * W;
  if (px >= Npx) return;
  const int row = (int)(px % H), col = (int)(px / H);
  const double npatch = (double)((2 * r + 1) * (2 * r + 1));
  const double mscale = 1.0 / npatch / 3.0;
  double b[3] = {0, 0, 0}, bb[3] = {0, 0, 0};
  for (int dx = -r; dx <= r; ++dx)
    for (int dy = -r; dy <= r; ++dy)
      for (int ch = 0; ch < 3; ++ch) {
        const int rr = row + dy, cc = col + dx;
        const double v = (rr < 0 || rr >= H || cc < 0 || cc >= W) ? 0.0 : im[(size_t)ch * Npx + (size_t)cc * H + rr];
        b[ch] += v; bb[ch] += v * v;
      }
  const double mean = (b[0] * mscale + b[1] * mscale) + b[2] * mscale;
  const double t1 = (bb[0] + bb[1]) + bb[2];
  const double t2 = (mean * b[0] + mean * b[1]) + mean * b[2];
  const double var = t1 - 2 * t2 + npatch * 3 * mean * mean;
  // the norm with the sign of the variance term: sqrt of a negative one is imaginary in MATLAB
  // (dispmap_ncc.m:141,171), which the cross kernel resolves from the two signs
  // (stored as the RECIPROCAL: the volume kernels multiply by it instead of dividing twice per output --
  //  a third of their instructions; the quotient moves by an ulp or two, the 1e-12 agreement with the
  //  NumPy restatement is unaffected.  A zero norm gives +-inf and, as with the division, a non-finite
  //  value that becomes 0, dispmap_ncc.m:190)
  const double rnorm = 1.0 / sqrt(fabs(var));
  if (st) {    // column-major planes (ncc_cross_kernel); the label-fastest path only wants the row-major ones
    st[px] = b[0]; st[Npx + px] = b[1]; st[2 * Npx + px] = b[2]; st[3 * Npx + px] = mean;
    st[4 * Npx + px] = var < 0 ? -rnorm : rnorm;
  }
  if (stT) {   // the same five planes row-major, for ncc_lane_kernel
    const int64_t t = (int64_t)row * W + col;
    stT[t] = b[0]; stT[Npx + t] = b[1]; stT[2 * Npx + t] = b[2]; stT[3 * Npx + t] = mean;
    stT[4 * Npx + t] = var < 0 ? -rnorm : rnorm;
  }
}

struct NccCol {  // what one step of the column walk reads: loaded one step ahead
  double l0, l1, l2, r0, r1, r2;          // the two pixels of the product two columns ahead
  double bL0, bL1, bL2, meanL, nL;        // statistics of im0 at (row, c)
  double bT0, bT1, bT2, meanT, nT;        // statistics of im1 at (row, c - d)
};

__device__ __forceinline__ void ncc_load(const NccFast &p, int64_t Npx, int row_c, int c, int d, NccCol &o) {
  // unconditional, clamped addresses (masked by the caller): every load of a step is in flight together
  const int H = p.H, W = p.W;
  int cp = c + 2; cp = cp < 0 ? 0 : cp > W - 1 ? W - 1 : cp;
  int cr = c + 2 - d; cr = cr < 0 ? 0 : cr > W - 1 ? W - 1 : cr;
  const size_t a = (size_t)cp * H + row_c, b = (size_t)cr * H + row_c;
  o.l0 = p.im0[a]; o.l1 = p.im0[Npx + a]; o.l2 = p.im0[2 * Npx + a];
  o.r0 = p.im1[b]; o.r1 = p.im1[Npx + b]; o.r2 = p.im1[2 * Npx + b];
  int cs = c < 0 ? 0 : c > W - 1 ? W - 1 : c;
  int ct = c - d; ct = ct < 0 ? 0 : ct > W - 1 ? W - 1 : ct;
  const size_t sa = (size_t)cs * H + row_c, sb = (size_t)ct * H + row_c;
  o.bL0 = p.st0[sa]; o.bL1 = p.st0[Npx + sa]; o.bL2 = p.st0[2 * Npx + sa]; o.meanL = p.st0[3 * Npx + sa]; o.nL = p.st0[4 * Npx + sa];
  o.bT0 = p.st1[sb]; o.bT1 = p.st1[Npx + sb]; o.bT2 = p.st1[2 * Npx + sb]; o.meanT = p.st1[3 * Npx + sb]; o.nT = p.st1[4 * Npx + sb];
}

// Statistics of the shifted image where the 5 x 5 window leaves the image on the right: its columns
// >= W are zero for imtr but not for im1, so the pre-pass planes do not apply (last two columns only).
// Out of line (arguments and result by value, in registers): inlined, its 75 unrolled taps cost the
// column walk a hundred VGPRs and half of its occupancy.
struct NccStat { double b0, b1, b2, mean, nrm; };
__device__ __attribute__((noinline)) NccStat ncc_right_border(const double *im1, int H, int W, int64_t Npx, int row, int c, int d) {
  double b0 = 0, b1 = 0, b2 = 0, q0 = 0, q1 = 0, q2 = 0;
#pragma unroll 1
  for (int dx = -2; dx <= 2; ++dx)
#pragma unroll 1
    for (int dy = -2; dy <= 2; ++dy) {
      const int rr = row + dy, cc = c + dx;
      const bool in = !(rr < 0 || rr >= H || cc < 0 || cc >= W || cc - d < 0);
      const size_t a = in ? (size_t)(cc - d) * H + rr : 0;
      const double v0 = in ? im1[a] : 0.0, v1 = in ? im1[Npx + a] : 0.0, v2 = in ? im1[2 * Npx + a] : 0.0;
      b0 += v0; q0 += v0 * v0; b1 += v1; q1 += v1 * v1; b2 += v2; q2 += v2 * v2;
    }
  const double mscale = 1.0 / 25.0 / 3.0;
  NccStat o;
  o.b0 = b0; o.b1 = b1; o.b2 = b2;
  o.mean = (b0 * mscale + b1 * mscale) + b2 * mscale;
  const double u1 = (q0 + q1) + q2;
  const double u2 = (o.mean * b0 + o.mean * b1) + o.mean * b2;
  const double varT = u1 - 2 * u2 + 75.0 * o.mean * o.mean;
  const double nn = 1.0 / sqrt(fabs(varT));   // signed reciprocal, as in the statistics planes
  o.nrm = varT < 0 ? -nn : nn;
  return o;
}

// value of the lane above (UP) / below in the wave of 64; the lane that falls off the end reads 0
// (DPP wave_shr:1 / wave_shl:1, bound_ctrl: one VALU move per half instead of a trip through LDS)
template <bool UP>
__device__ __forceinline__ double lane_neighbour(double v) {
  int lo = __double2loint(v), hi = __double2hiint(v);
  lo = __builtin_amdgcn_update_dpp(0, lo, UP ? 0x138 : 0x130, 0xF, 0xF, true);
  hi = __builtin_amdgcn_update_dpp(0, hi, UP ? 0x138 : 0x130, 0xF, 0xF, true);
  return __hiloint2double(hi, lo);
}

__global__ __launch_bounds__(kNccWaves * 64, 2) void ncc_cross_kernel(NccFast p) {
  __shared__ double tr[2][kNccRows][kNccWaves + 1];  // label-fastest transpose, double buffered
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int H = p.H, W = p.W;
  const int64_t Npx = (int64_t)H * W;
  const int row0 = blockIdx.x * kNccRows, row = row0 - 2 + lane;
  const int di = blockIdx.y * kNccWaves + wave;
  const bool dvalid = di < p.D;
  const int d = dvalid ? (int)p.disp[di] : 0;
  const int c_begin = blockIdx.z * kNccCols, c_end = c_begin + kNccCols < W ? c_begin + kNccCols : W;
  const bool rvalid = row >= 0 && row < H;
  const int row_c = row < 0 ? 0 : row > H - 1 ? H - 1 : row;
  const bool outlane = lane >= 2 && lane < 2 + kNccRows && row < H;
  const double npatch3 = 75.0;
  auto product = [&](int c, const NccCol &v) -> double {  // sum over the channels of im0(row, c) * imtr(row, c)
    const double pr = (v.l0 * v.r0 + v.l1 * v.r1) + v.l2 * v.r2;
    return (rvalid && dvalid && c >= 0 && c < W && c - d >= 0) ? pr : 0.0;
  };
  NccCol cur, nxt;
  double P0, P1, P2, P3;
  ncc_load(p, Npx, row_c, c_begin - 4, d, cur); P0 = product(c_begin - 2, cur);
  ncc_load(p, Npx, row_c, c_begin - 3, d, cur); P1 = product(c_begin - 1, cur);
  ncc_load(p, Npx, row_c, c_begin - 2, d, cur); P2 = product(c_begin, cur);
  ncc_load(p, Npx, row_c, c_begin - 1, d, cur); P3 = product(c_begin + 1, cur);
  ncc_load(p, Npx, row_c, c_begin, d, cur);
#pragma unroll 1
  for (int c = c_begin; c < c_end; ++c) {
    ncc_load(p, Npx, row_c, c + 1, d, nxt);  // the next step's loads travel while this one computes
    const double P4 = product(c + 2, cur);
    const double h = (((P0 + P1) + P2) + P3) + P4;
    const double hm1 = lane_neighbour<true>(h), hp1 = lane_neighbour<false>(h);      // rows - 1, + 1
    const double hm2 = lane_neighbour<true>(hm1), hp2 = lane_neighbour<false>(hp1);  // rows - 2, + 2
    const double c1 = (((hm2 + hm1) + h) + hp1) + hp2;
    P0 = P1; P1 = P2; P2 = P3; P3 = P4;
    double val = 0.0;
    const bool live = outlane && dvalid && c >= d;  // columns left of round(d + 1) are masked (dispmap_ncc.m:190-191)
    double bT0 = cur.bT0, bT1 = cur.bT1, bT2 = cur.bT2, meanT = cur.meanT, nT = cur.nT;
    if (c + 2 >= W && live) {
      const NccStat o = ncc_right_border(p.im1, H, W, Npx, row, c, d);
      bT0 = o.b0; bT1 = o.b1; bT2 = o.b2; meanT = o.mean; nT = o.nrm;
    }
    {
      const double c2 = (cur.meanL * bT0 + cur.meanL * bT1) + cur.meanL * bT2;
      const double c3 = (meanT * cur.bL0 + meanT * cur.bL1) + meanT * cur.bL2;
      const double num = c1 - c2 - c3 + npatch3 * meanT * cur.meanL;
      const double q = num * fabs(cur.nL) * fabs(nT);
      // real(num / normL / normT) with imaginary norms for negative variance terms (dispmap_ncc.m:188-192)
      const bool negL = cur.nL < 0, negT = nT < 0;
      val = negL == negT ? (negL ? -q : q) : 0.0;
      if (!isfinite(val) || !live) val = 0.0;
    }
    if (p.layout == 0) {
      if (outlane && dvalid) p.out[((size_t)di * W + c) * H + row] = val;
    } else {
      double (*buf)[kNccWaves + 1] = tr[(c - c_begin) & 1];
      if (lane >= 2 && lane < 2 + kNccRows) buf[lane - 2][wave] = val;
      __syncthreads();
      const int t = threadIdx.x;
      if (t < kNccRows * kNccWaves) {
        const int rr = t / kNccWaves, ww = t % kNccWaves;
        const int orow = row0 + rr, odi = blockIdx.y * kNccWaves + ww;
        if (orow < H && odi < p.D) p.out[((size_t)c * H + orow) * p.D + odi] = buf[rr][ww];
      }
    }
    cur = nxt;
  }
}

// ---- NCC volume, label-fastest layout: lane = disparity (round 3) ------------------------------
// ncc_cross_kernel above has lane = image row, which writes the MATLAB layout (H x W x D) in full rows
// but reaches the label-fastest layout TRW-S reads only through an LDS transpose with one workgroup
// barrier per column and 64-byte runs, re-fetching both images for every group of eight disparities
// (round 2: 151 us for 89 MB, 6 % of the write roofline, 3.8 x the algorithmic traffic).  Here a lane
// is a disparity: pixel (r, c) of the volume is ONE coalesced store of D doubles, the right image and
// its statistics are read as im1(r, c - d) from row-major copies (consecutive lanes, consecutive
// addresses), the left image and its statistics are the same for every lane (scalar loads), and both
// images are read once per 12-row tile for ALL disparities.  A workgroup is 16 waves = 16 image rows
// (12 output rows + the 2 + 2 rows the 5 x 5 window needs); a wave walks along its row keeping the
// window's five column products in registers, posts the horizontal sum in LDS and the 12 output waves
// add the five rows -- the same association as ncc_cross_kernel (products (c0 + c1) + c2, columns
// left to right, rows top to bottom), so both kernels give the same bits.
constexpr int kNlWaves = 16;
constexpr int kNlRows = kNlWaves - 4;

struct NccLane {
  int H, W, D;
  const double *im1;          // MATLAB layout (right-border statistics)
  const double *i0T, *i1T;    // [3][H][W]: row-major copies of the two images
  const double *s0T, *s1T;    // [5][H][W]: row-major statistics (ncc_stats_kernel)
  const double *disp;
  double *out;                // [W][H][D]
  int cols;                   // columns per workgroup
};

// row-major copy of an H x W x 3 image (MATLAB layout: [ch][col][row])
__global__ __launch_bounds__(kTB) void ncc_transpose_kernel(const double *im, int H, int W, double *out) {
  __shared__ double tile[16][17];
  const int64_t Npx = (int64_t)H * W;
  const int tx = threadIdx.x & 15, ty = threadIdx.x >> 4;
  const int r0 = blockIdx.x * 16, c0 = blockIdx.y * 16, ch = blockIdx.z;
  if (r0 + tx < H && c0 + ty < W) tile[ty][tx] = im[(size_t)ch * Npx + (size_t)(c0 + ty) * H + r0 + tx];
  __syncthreads();
  if (r0 + ty < H && c0 + tx < W) out[(size_t)ch * Npx + (size_t)(r0 + ty) * W + c0 + tx] = tile[tx][ty];
}

// LDS of ncc_lane_kernel: per wave (= image row) the row segments it reads, staged once per workgroup --
// every load of a workgroup is in flight together, and the column walk itself touches LDS only:
//   seg1 [8][kNlSeg1]: im1 (3 channels) and its statistics (5 planes) at columns seg_c0 + x
//   seg0 [8][kNlSeg0]: im0 and its statistics at columns c_begin - 2 + x
constexpr int kNlSeg1 = 96, kNlSeg0 = 36;
constexpr int kNlImDoubles = 3 * (kNlSeg1 + kNlSeg0);   // per wave: the two images' row segments
constexpr int kNlStDoubles = 5 * (kNlSeg1 + kNlSeg0);   // per OUTPUT wave: their statistics
constexpr int kNlLdsDoubles = kNlWaves * kNlImDoubles + kNlRows * kNlStDoubles + 4 * kNlWaves * 64;   // + two pairs of horizontal sums

__global__ __launch_bounds__(kNlWaves * 64) void ncc_lane_kernel(NccLane p) {
  extern __shared__ __attribute__((aligned(16))) double nl_lds[];
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  double *seg1 = nl_lds + wave * kNlImDoubles, *seg0 = seg1 + 3 * kNlSeg1;                 // images: [3][kNlSeg1], [3][kNlSeg0]
  const int ow = wave >= 2 && wave < 2 + kNlRows ? wave - 2 : 0;
  double *sts1 = nl_lds + kNlWaves * kNlImDoubles + ow * kNlStDoubles, *sts0 = sts1 + 5 * kNlSeg1;  // statistics: [5][..]
  double (*hbuf)[kNlWaves][64] = (double (*)[kNlWaves][64])(nl_lds + kNlWaves * kNlImDoubles + kNlRows * kNlStDoubles);   // [4][16][64]
  const int H = p.H, W = p.W;
  const int64_t Npx = (int64_t)H * W;
  const int row = blockIdx.x * kNlRows - 2 + wave;
  const int di = blockIdx.y * 64 + lane;
  const bool dvalid = di < p.D;
  const int d = dvalid ? (int)p.disp[di] : (int)p.disp[blockIdx.y * 64];
  // smallest / largest disparity of this 64-label chunk (the host checked that the segment holds the span)
  int dmin = d, dmax = d;
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    const int a = __shfl_xor(dmin, o, 64), b2 = __shfl_xor(dmax, o, 64);
    dmin = a < dmin ? a : dmin; dmax = b2 > dmax ? b2 : dmax;
  }
  const int c_begin = blockIdx.z * p.cols, c_end = c_begin + p.cols < W ? c_begin + p.cols : W;
  const bool rvalid = row >= 0 && row < H;
  const int row_c = row < 0 ? 0 : row > H - 1 ? H - 1 : row;
  const size_t rowoff = (size_t)row_c * W;
  const bool outw = wave >= 2 && wave < 2 + kNlRows && row < H;   // (uniform) this wave writes a row of the volume
  const int seg_c0 = c_begin - 2 - dmax, seg0_c0 = c_begin - 2;
  // ---- stage the row segments (clamped columns: what lies outside the image is masked below)
  {
    double v1[8][2], v0[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      const double *src1 = (k < 3 ? p.i1T + (size_t)k * Npx : p.s1T + (size_t)(k - 3) * Npx) + rowoff;
      const double *src0 = (k < 3 ? p.i0T + (size_t)k * Npx : p.s0T + (size_t)(k - 3) * Npx) + rowoff;
      // (image planes: zero outside the image and in rows outside it, so that the products need no
      //  mask in the loop -- the shifted image is zero-filled, conv2 pads with zeros)
      const bool need = k < 3 || outw;   // (uniform) statistics only where a row of the volume is written
#pragma unroll
      for (int hlf = 0; hlf < 2; ++hlf) {
        const int gg = seg_c0 + lane + 64 * hlf, g = gg < 0 ? 0 : gg > W - 1 ? W - 1 : gg;
        const double v = need ? src1[g] : 0.0;
        v1[k][hlf] = (k < 3 && (gg < 0 || gg > W - 1 || !rvalid)) ? 0.0 : v;
      }
      const int gg0 = seg0_c0 + lane, g0 = gg0 < 0 ? 0 : gg0 > W - 1 ? W - 1 : gg0;
      const double v = need ? src0[g0] : 0.0;
      v0[k] = (k < 3 && (gg0 < 0 || gg0 > W - 1 || !rvalid)) ? 0.0 : v;
    }
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      if (k >= 3 && !outw) continue;
      double *d1 = k < 3 ? seg1 + k * kNlSeg1 : sts1 + (k - 3) * kNlSeg1;
      double *d0 = k < 3 ? seg0 + k * kNlSeg0 : sts0 + (k - 3) * kNlSeg0;
      d1[lane] = v1[k][0];
      if (lane + 64 < kNlSeg1) d1[lane + 64] = v1[k][1];
      if (lane < kNlSeg0) d0[lane] = v0[k];
    }
  }
  __syncthreads();
  const double npatch3 = 75.0;
  // sum over the channels of im0(row, c) * imtr(row, c), c = column of the product (indices are inside the
  // segments by construction; columns outside the image hold zeros)
  const double *s1d = seg1 - d - seg_c0;   // this lane's view of seg1: s1d[c] = plane 0 at column c - d
  const double *s0c = seg0 - seg0_c0;
  const double *t1d = sts1 - d - seg_c0, *t0c = sts0 - seg0_c0;   // the same views of the statistics
  auto product = [&](int c) -> double {
    const double l0 = s0c[c], l1 = s0c[kNlSeg0 + c], l2 = s0c[2 * kNlSeg0 + c];
    const double r0 = s1d[c], r1 = s1d[kNlSeg1 + c], r2 = s1d[2 * kNlSeg1 + c];
    return (l0 * r0 + l1 * r1) + l2 * r2;
  };
  double P0 = product(c_begin - 2), P1 = product(c_begin - 1), P2 = product(c_begin), P3 = product(c_begin + 1);
  // two columns per workgroup barrier (the horizontal sums of both are posted first)
  double *outp = p.out + ((size_t)c_begin * H + (row_c)) * p.D + di;   // this lane's element of pixel (row, c_begin)
  const size_t ostep = (size_t)H * p.D;
  auto emit = [&](int c, double h, const double (*hb)[64], double *dst) {
    const double hm2 = hb[wave - 2][lane], hm1 = hb[wave - 1][lane], hp1 = hb[wave + 1][lane], hp2 = hb[wave + 2][lane];
    const double bL0 = t0c[c], bL1 = t0c[kNlSeg0 + c], bL2 = t0c[2 * kNlSeg0 + c];
    const double meanL = t0c[3 * kNlSeg0 + c], nL = t0c[4 * kNlSeg0 + c];
    double bT0 = t1d[c], bT1 = t1d[kNlSeg1 + c], bT2 = t1d[2 * kNlSeg1 + c];
    double meanT = t1d[3 * kNlSeg1 + c], nT = t1d[4 * kNlSeg1 + c];
    const double c1 = (((hm2 + hm1) + h) + hp1) + hp2;
    const bool live = dvalid && c >= d;  // columns left of round(d + 1) are masked (dispmap_ncc.m:190-191)
    if (c + 2 >= W && live) {
      const NccStat o = ncc_right_border(p.im1, H, W, Npx, row, c, d);
      bT0 = o.b0; bT1 = o.b1; bT2 = o.b2; meanT = o.mean; nT = o.nrm;
    }
    const double c2 = (meanL * bT0 + meanL * bT1) + meanL * bT2;
    const double c3 = (meanT * bL0 + meanT * bL1) + meanT * bL2;
    const double num = c1 - c2 - c3 + npatch3 * meanT * meanL;
    const double q = num * fabs(nL) * fabs(nT);
    // real(num / normL / normT) with imaginary norms for negative variance terms (dispmap_ncc.m:188-192)
    const bool negL = nL < 0, negT = nT < 0;
    double val = negL == negT ? (negL ? -q : q) : 0.0;
    if (!isfinite(val) || !live) val = 0.0;
    if (dvalid) *dst = val;
  };
  int parity = 0;
#pragma unroll 1
  for (int c = c_begin; c < c_end; c += 2, parity ^= 1) {
    const double Pa = product(c + 2), Pb = product(c + 3);
    const double ha = (((P0 + P1) + P2) + P3) + Pa;
    const double hb2 = (((P1 + P2) + P3) + Pa) + Pb;
    P0 = P2; P1 = P3; P2 = Pa; P3 = Pb;
    double (*hA)[64] = hbuf[2 * parity], (*hB)[64] = hbuf[2 * parity + 1];
    hA[wave][lane] = ha; hB[wave][lane] = hb2;
    __syncthreads();   // (double buffered: the other pair is written by the next step, after every wave has read this one)
    if (outw) {
      emit(c, ha, hA, outp);
      if (c + 1 < c_end) emit(c + 1, hb2, hB, outp + ostep);
    }
    outp += 2 * ostep;
  }
}

// ---- NCC sampling (dispmap_ncc.m:222-276) -----------------------------------

__device__ __forceinline__ double vol(const double *ncc, int64_t Npx, int D, int layout, int64_t px, int k) {
  return layout == 0 ? ncc[(size_t)k * Npx + px] : ncc[(size_t)px * D + k];
}

__device__ __forceinline__ void lagrange(const double *ncc, int64_t Npx, int D, int layout, int64_t px,
                                         const double *dv, int t2 /*1-based*/, double y2, bool ok, double &r,
                                         double &pp, double &q, double &d2) {
  const int t1 = ok ? t2 - 1 : t2, t3 = ok ? t2 + 1 : t2;
  const double d1 = dv[t1 - 1], d3 = dv[t3 - 1];
  d2 = dv[t2 - 1];
  const double y1 = vol(ncc, Npx, D, layout, px, t1 - 1), y3 = vol(ncc, Npx, D, layout, px, t3 - 1);
  const double a = y1 / (d1 - d2) / (d1 - d3);
  const double b = y2 / (d2 - d1) / (d2 - d3);
  const double c = y3 / (d3 - d1) / (d3 - d2);
  r = a + b + c;
  pp = -(a * (d2 + d3) + b * (d1 + d3) + c * (d1 + d2));
  q = a * d2 * d3 + b * d1 * d3 + c * d1 * d2;
}

__global__ __launch_bounds__(kTB) void ncc_unary_kernel(const double *ncc, int H, int W, int D, int layout,
                                                       const double *dv, double dmin, double dmax,
                                                       double unary_weight, const double *assign, double *U,
                                                       int ascending) {
  const int64_t px = (int64_t)blockIdx.x * kTB + threadIdx.x;
  const int64_t Npx = (int64_t)H * W;
  if (px >= Npx) return;
  const double x = (double)(px / H + 1), y = (double)(px % H + 1);
  const double disp = plane_disp(assign + 4 * (size_t)px, x, y);
  // nearest sample, ties to the LARGER index (dispmap_ncc.m:230-236: a scan over all slices with
  // "<="; y2 keeps its initial 1 if no comparison succeeds, i.e. for a NaN disparity)
  int t2 = 1;
  double smallest = fabs(disp - dv[0]), y2 = 1.0;
  if (ascending) {
    // strictly ascending samples: |disp - dv[i]| falls, then rises (rounding is monotone), so the
    // scan's answer is the later of the two samples around disp -- found by bisection -- unless
    // further samples tie with it after rounding, which the short forward walk covers
    if (disp == disp) {
      int lo = 0, hi = D;  // first index with dv[i] > disp
      while (lo < hi) { const int mid = (lo + hi) >> 1; if (dv[mid] > disp) hi = mid; else lo = mid + 1; }
      int j = lo > 0 ? lo - 1 : 0;
      double best = fabs(disp - dv[j]);
      // (an earlier sample can tie with j only after rounding; the scan would still end on j or later)
      while (j + 1 < D && fabs(disp - dv[j + 1]) <= best) { ++j; best = fabs(disp - dv[j]); }
      t2 = j + 1; smallest = best; y2 = vol(ncc, Npx, D, layout, px, j);
    }
  } else {
    for (int i = 0; i < D; ++i) {
      const double nd = fabs(disp - dv[i]);
      if (nd <= smallest) { t2 = i + 1; y2 = vol(ncc, Npx, D, layout, px, i); smallest = nd; }
    }
  }
  const bool ok = t2 < D && t2 > 1;
  double r, pp, q, d2;
  lagrange(ncc, Npx, D, layout, px, dv, t2, y2, ok, r, pp, q, d2);
  double v = r * (disp * disp) + pp * disp + q;
  if (t2 == 1) v = vol(ncc, Npx, D, layout, px, 0);
  if (t2 == D) v = vol(ncc, Npx, D, layout, px, D - 1);
  if (!(disp <= dmax && disp >= dmin)) v = -1e6;
  U[px] = unary_weight * (1 - v);
}

__global__ __launch_bounds__(kTB) void ncc_best_disp_kernel(const double *ncc, int H, int W, int D, int layout,
                                                           const double *dv, double *best) {
  const int64_t px = (int64_t)blockIdx.x * kTB + threadIdx.x;
  const int64_t Npx = (int64_t)H * W;
  if (px >= Npx) return;
  int t2 = 1;
  double y2 = vol(ncc, Npx, D, layout, px, 0);
  for (int i = 1; i < D; ++i) {
    const double v = vol(ncc, Npx, D, layout, px, i);
    if (v > y2) { y2 = v; t2 = i + 1; }  // max(...,[],3): first maximum
  }
  const bool ok = t2 < D && t2 > 1;
  double r, pp, q, d2;
  lagrange(ncc, Npx, D, layout, px, dv, t2, y2, ok, r, pp, q, d2);
  best[px] = ok ? -pp / r / 2 : d2;
}

// ---- globalstereo unary -------------------------------------------------------

__global__ __launch_bounds__(kTB) void globalstereo_unary_kernel(const double *im0, const double *im1, int H,
                                                                int W, int C, const double *P2 /*4x3 col major*/,
                                                                double d_min, double d_step, double col_thresh,
                                                                const double *assign, double *U) {
  const int64_t px = (int64_t)blockIdx.x * kTB + threadIdx.x;
  const int64_t Npx = (int64_t)H * W;
  if (px >= Npx) return;
  const double x = (double)(px / H + 1), y = (double)(px % H + 1);
  const double dhat = (plane_disp(assign + 4 * (size_t)px, x, y) - d_min) / d_step;
  const double disp = d_step * (dhat + d_min);  // dispmap_globalstereo.m:356 (sic)
  double T[3];
  for (int k = 0; k < 3; ++k)
    T[k] = (x * P2[4 * k] + y * P2[4 * k + 1]) + (1.0 * P2[4 * k + 2] + disp * P2[4 * k + 3]);
  const double Nn = 1.0 / T[2];
  const double X = T[0] * Nn, Y = T[1] * Nn;
  const double oobv = -1000.0, dw = (double)W, dh = (double)H;
  double ssd = 0;
  for (int c = 0; c < C; ++c) {
    const double *A = im1 + (size_t)c * Npx;
    double o = oobv;
    if (X >= 1 && Y >= 1) {  // vgg_interp2.cxx:260-317
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
    const double m = o - im0[(size_t)c * Npx + px];
    ssd = ssd + m * m;
  }
  U[px] = log(2.0) - log(exp(ssd * (-1.0 / (col_thresh * C))) + 1.0);
}

// ---- device-resident fusion moves (stereo_fusion_*) -----------------------------------------
// dispmap_super.m:78-82: assignment(:, labelling == 1) = proposal(:, labelling == 1); the cached
// unary of the current assignment follows (the unary is a per-pixel function of the pixel's plane).
__global__ __launch_bounds__(kTB) void fusion_scatter_kernel(int64_t N, const int8_t *label, const double *prop,
                                                            const double *U1, double *cur, double *Ucur) {
  const int64_t i = (int64_t)blockIdx.x * kTB + threadIdx.x;
  if (i >= N || label[i] != 1) return;
  cur[4 * i] = prop[4 * i]; cur[4 * i + 1] = prop[4 * i + 1];
  cur[4 * i + 2] = prop[4 * i + 2]; cur[4 * i + 3] = prop[4 * i + 3];
  Ucur[i] = U1[i];
}

// unary of proposal k into the K x N (label fastest) layout TRW-S consumes
__global__ __launch_bounds__(kTB) void fusion_interleave_kernel(int64_t N, int K, int k, const double *Uk,
                                                               double *unary) {
  const int64_t i = (int64_t)blockIdx.x * kTB + threadIdx.x;
  if (i < N) unary[i * K + k] = Uk[i];
}

// dispmap_super.m:189-195: every pixel takes the plane of the proposal TRW-S chose for it;
// its unary follows from the K x N table
__global__ __launch_bounds__(kTB) void fusion_select_kernel(int64_t N, int K, const int32_t *label,
                                                           const double *props, const double *unary,
                                                           double *cur, double *Ucur) {
  const int64_t i = (int64_t)blockIdx.x * kTB + threadIdx.x;
  if (i >= N) return;
  const int k = label[i];
  const double *P = props + (size_t)k * 4 * N + 4 * i;
  cur[4 * i] = P[0]; cur[4 * i + 1] = P[1]; cur[4 * i + 2] = P[2]; cur[4 * i + 3] = P[3];
  Ucur[i] = unary[i * K + k];
}

// fixed-shape reduction (same tree for every run): partial[b] = sum of a 2048-element chunk
__global__ __launch_bounds__(kTB) void fusion_sum_kernel(const double *x, int64_t n, double *partial) {
  __shared__ double sh[kTB];
  const int64_t base = (int64_t)blockIdx.x * (kTB * 8);
  double acc = 0;
  for (int k = 0; k < 8; ++k) {
    const int64_t i = base + (int64_t)k * kTB + threadIdx.x;
    if (i < n) acc += x[i];
  }
  sh[threadIdx.x] = acc;
  __syncthreads();
  for (int s2 = kTB / 2; s2 > 0; s2 >>= 1) {
    if ((int)threadIdx.x < s2) sh[threadIdx.x] += sh[threadIdx.x + s2];
    __syncthreads();
  }
  if (threadIdx.x == 0) partial[blockIdx.x] = sh[0];
}

// ---- proposals built on the device --------------------------------------------
// A proposal of the reference's examples is never an arbitrary 4 x N array: it is one plane for
// every pixel (example_ncc.m:24-41: plane fits and fronto-parallel planes, repmat'ed,
// dispmap_ncc.m:65) or one plane per image segment (dispmap_globalstereo.m:154-192 SegPln).  Built
// from a 4 x S table (+ an N-vector of segment ids) in HBM, so a move uploads 32 S bytes, not 32 N.
__global__ __launch_bounds__(kTB) void proposal_from_planes_kernel(int64_t N, const double *planes, int S,
                                                                  const int32_t *segments, double *prop) {
  const int64_t i = (int64_t)blockIdx.x * kTB + threadIdx.x;
  if (i >= N) return;
  int sgm = segments ? segments[i] : 0;
  sgm = sgm < 0 ? 0 : sgm >= S ? S - 1 : sgm;
  prop[4 * i] = planes[4 * sgm]; prop[4 * i + 1] = planes[4 * sgm + 1];
  prop[4 * i + 2] = planes[4 * sgm + 2]; prop[4 * i + 3] = planes[4 * sgm + 3];
}

// dispmap_ncc.m:48-92 generate_new_plane_RANSAC + fit_plane_to_points: the plane through the
// winner-takes-all disparities of the pixels within radius r of (x, y) (1-based, x = column): total
// least squares (kernel 2) or 20 rounds of iteratively reweighted least squares (kernel 1).  The
// reference takes V(:, end) of svd(w .* cost): the eigenvector of the smallest eigenvalue of
// sum_i w_i^2 c_i c_i' -- found here by cyclic Jacobi rotations of that 3 x 3 matrix (MATLAB's svd
// is outside the reference tree: agreement with a LAPACK SVD to ~1e-9, not bit for bit; the sign
// of the vector cancels in p / p(3)).  At most (2r+1)^2 points: one thread does the arithmetic in
// the reference's order (pixels in column-major order), the rest of the wave idles.
constexpr int kFitMax = 1024;
// One workgroup per fit: blockIdx.x picks the centre (xy = [x_0 y_0 x_1 y_1 ...], 1-based pixel
// coordinates) and the output slot -- the whole lattice of example_ncc.m:24-32 is one launch.
__global__ __launch_bounds__(64) void fit_plane_kernel(const double *best, int H, int W, const double *xy, double r,
                                                      int kernel, double *out_all /* (4 + count) per fit */) {
  __shared__ double px[kFitMax], py[kFitMax], pz[kFitMax], wt[kFitMax];
  if (threadIdx.x != 0) return;
  const double x = xy[2 * blockIdx.x], y = xy[2 * blockIdx.x + 1];
  double *out = out_all + 5 * (size_t)blockIdx.x;
  int n = 0;
  const int c0 = (int)fmax(1.0, floor(x - r)), c1 = (int)fmin((double)W, ceil(x + r));
  const int r0 = (int)fmax(1.0, floor(y - r)), r1 = (int)fmin((double)H, ceil(y + r));
  for (int c = c0; c <= c1; ++c)
    for (int rr = r0; rr <= r1; ++rr) {
      const double dx = (double)c - x, dy = (double)rr - y;
      if (sqrt(dx * dx + dy * dy) < r && n < kFitMax) {
        px[n] = c; py[n] = rr; pz[n] = best[(size_t)(c - 1) * H + (rr - 1)]; wt[n] = 1.0; ++n;
      }
    }
  out[4] = n;
  if (n < 3) { out[0] = 0; out[1] = 0; out[2] = 1; out[3] = 0; return; }
  double cx = 0, cy = 0, cz = 0;
  for (int i = 0; i < n; ++i) { cx += px[i]; cy += py[i]; cz += pz[i]; }
  cx /= n; cy /= n; cz /= n;
  double v[3] = {0, 0, 1};
  const int rounds = kernel == 1 ? 20 : 1;
  for (int it = 0; it < rounds; ++it) {
    double A[3][3] = {{0, 0, 0}, {0, 0, 0}, {0, 0, 0}};
    for (int i = 0; i < n; ++i) {
      const double w = wt[i], a = -(px[i] - cx) * w, b = -(py[i] - cy) * w, c = -(pz[i] - cz) * w;
      A[0][0] += a * a; A[0][1] += a * b; A[0][2] += a * c; A[1][1] += b * b; A[1][2] += b * c; A[2][2] += c * c;
    }
    A[1][0] = A[0][1]; A[2][0] = A[0][2]; A[2][1] = A[1][2];
    double V[3][3] = {{1, 0, 0}, {0, 1, 0}, {0, 0, 1}};
    for (int sweep = 0; sweep < 30; ++sweep) {
      const double off = fabs(A[0][1]) + fabs(A[0][2]) + fabs(A[1][2]);
      if (off < 1e-300) break;
      for (int pp = 0; pp < 2; ++pp)
        for (int q = pp + 1; q < 3; ++q) {
          if (A[pp][q] == 0) continue;
          const double theta = (A[q][q] - A[pp][pp]) / (2 * A[pp][q]);
          const double t = (theta >= 0 ? 1.0 : -1.0) / (fabs(theta) + sqrt(theta * theta + 1));
          const double cs = 1 / sqrt(t * t + 1), sn = t * cs;
          for (int k = 0; k < 3; ++k) {  // A <- A J
            const double akp = A[k][pp], akq = A[k][q];
            A[k][pp] = cs * akp - sn * akq; A[k][q] = sn * akp + cs * akq;
          }
          for (int k = 0; k < 3; ++k) {  // A <- J' A
            const double apk = A[pp][k], aqk = A[q][k];
            A[pp][k] = cs * apk - sn * aqk; A[q][k] = sn * apk + cs * aqk;
          }
          for (int k = 0; k < 3; ++k) {
            const double vkp = V[k][pp], vkq = V[k][q];
            V[k][pp] = cs * vkp - sn * vkq; V[k][q] = sn * vkp + cs * vkq;
          }
        }
    }
    int m = 0;
    if (A[1][1] < A[m][m]) m = 1;
    if (A[2][2] < A[m][m]) m = 2;
    v[0] = V[0][m]; v[1] = V[1][m]; v[2] = V[2][m];
    if (kernel == 1)
      for (int i = 0; i < n; ++i)  // w = sqrt(abs(cost * V(:, end)))
        wt[i] = sqrt(fabs(-(px[i] - cx) * v[0] + -(py[i] - cy) * v[1] + -(pz[i] - cz) * v[2]));
  }
  const double p4 = -(v[0] * cx + v[1] * cy + v[2] * cz);
  out[0] = v[0] / v[2]; out[1] = v[1] / v[2]; out[2] = v[2] / v[2]; out[3] = p4 / v[2];
}

template <class F>
int guarded(const char *what, char *err, size_t errcap, F f) {
  if (stereo_hip_device_count() < 1)
    return fail(std::string(what) + ": no HIP device available (the HIP path has no CPU fallback)", err, errcap);
  try {
    f();
    STEREO_HIP_CHECK(hipGetLastError());
    STEREO_HIP_CHECK(hipDeviceSynchronize());
    return 0;
  } catch (const HipError &e) {
    return fail(e.msg, err, errcap);
  } catch (const std::exception &e) {
    return fail(std::string(what) + ": " + e.what(), err, errcap);
  }
}

inline unsigned blocks(int64_t n) { return (unsigned)((n + kTB - 1) / kTB); }

void check_planes(const double *pl, int64_t M) {
  for (int64_t i = 0; i < M; ++i)
    if (pl[4 * i + 2] == 0) throw std::runtime_error("Infinite disparity");  // dispmap_super.m:323-325
}

template <class T>
void download(T *dst, const DevBuf<T> &b, size_t n) {
  STEREO_HIP_CHECK(hipMemcpy(dst, b.p, n * sizeof(T), hipMemcpyDeviceToHost));
}

bool strictly_ascending(const double *d, int D) {
  for (int i = 0; i < D; ++i)
    if (!(d[i] == d[i]) || (i > 0 && !(d[i] > d[i - 1]))) return false;
  return true;
}

// NCC volume on the device: the staged kernels when every disparity is an integer in [0, W) and the
// patch is 5 x 5 (what dispmap_ncc.m:24 always asks for), the per-tap kernel otherwise.
void launch_ncc_volume(const double *d_im0, const double *d_im1, int H, int W, const double *h_disp,
                       const double *d_disp, int D, int patchsize, int layout, double *d_out) {
  const size_t npx = (size_t)H * W;
  bool staged = patchsize == 2 && !std::getenv("STEREO_HIP_NCC_NAIVE");
  for (int i = 0; i < D && staged; ++i) staged = h_disp[i] >= 0 && h_disp[i] < W && h_disp[i] == std::floor(h_disp[i]);
  if (!staged) {
    NccParams p{H, W, D, patchsize, d_im0, d_im1, d_disp, d_out, layout};
    hipLaunchKernelGGL(ncc_volume_kernel, dim3(blocks((int64_t)npx * D)), dim3(kTB), 0, 0, p);
    return;
  }
  // (ncc_lane_kernel stages 96 columns of the right image per row: the disparities of every 64-label
  //  chunk must span less than that minus the workgroup's own columns)
  bool lanes_ok = layout == 1 && !std::getenv("STEREO_HIP_NCC_ROWLANES");
  int span = 0;
  for (int c0 = 0; c0 < D && lanes_ok; c0 += 64) {
    double lo = h_disp[c0], hi = h_disp[c0];
    for (int i = c0; i < std::min(D, c0 + 64); ++i) { lo = std::min(lo, h_disp[i]); hi = std::max(hi, h_disp[i]); }
    span = std::max(span, (int)(hi - lo));
  }
  lanes_ok = lanes_ok && span + 5 + 8 <= kNlSeg1;
  if (lanes_ok) {
    // label-fastest volume: lane = disparity (ncc_lane_kernel) over row-major copies of images and statistics
    // (row-major staging copies only, in stream-ordered scratch: released behind the kernel without a
    //  device-wide synchronisation; the column-major statistics are the other path's)
    StreamBuf<double> s0T, s1T, i0T, i1T;
    s0T.alloc(5 * npx); s1T.alloc(5 * npx); i0T.alloc(3 * npx); i1T.alloc(3 * npx);
    const dim3 tg((H + 15) / 16, (W + 15) / 16, 3);
    hipLaunchKernelGGL(ncc_transpose_kernel, tg, dim3(kTB), 0, 0, d_im0, H, W, i0T.p);
    hipLaunchKernelGGL(ncc_transpose_kernel, tg, dim3(kTB), 0, 0, d_im1, H, W, i1T.p);
    hipLaunchKernelGGL(ncc_stats_kernel, dim3(blocks(npx)), dim3(kTB), 0, 0, d_im0, H, W, patchsize, (double *)nullptr, s0T.p);
    hipLaunchKernelGGL(ncc_stats_kernel, dim3(blocks(npx)), dim3(kTB), 0, 0, d_im1, H, W, patchsize, (double *)nullptr, s1T.p);
    // enough workgroups for every CU: the columns are cut so that row tiles x column chunks x 64-label
    // chunks reach ~2 workgroups per CU (each chunk re-computes four columns of products)
    int cus = 256, dev = 0;
    if (hipGetDevice(&dev) == hipSuccess) (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
    const int rt = (H + kNlRows - 1) / kNlRows, dc = (D + 63) / 64;
    int chunks = std::max(1, (2 * cus + rt * dc - 1) / (rt * dc));
    int cols = std::max(8, (W + chunks - 1) / chunks);
    cols = std::min({cols, kNlSeg0 - 5, kNlSeg1 - 5 - span});   // what the staged segments hold (two columns per step: one more product column)
    chunks = (W + cols - 1) / cols;
    NccLane f{H, W, D, d_im1, i0T.p, i1T.p, s0T.p, s1T.p, d_disp, d_out, cols};
    const size_t lds = sizeof(double) * kNlLdsDoubles;
    static thread_local int lds_set_on = -1;   // the attribute is per device: set once per device and thread
    if (lds_set_on != dev) {
      STEREO_HIP_CHECK(hipFuncSetAttribute((const void *)ncc_lane_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
      lds_set_on = dev;
    }
    hipLaunchKernelGGL(ncc_lane_kernel, dim3(rt, dc, chunks), dim3(kNlWaves * 64), lds, 0, f);
    STEREO_HIP_CHECK(hipGetLastError());
    return;   // (the staging buffers are freed in stream order behind the kernel)
  }
  DevBuf<double> st0, st1;
  st0.alloc(5 * npx); st1.alloc(5 * npx);
  hipLaunchKernelGGL(ncc_stats_kernel, dim3(blocks(npx)), dim3(kTB), 0, 0, d_im0, H, W, patchsize, st0.p, (double *)nullptr);
  hipLaunchKernelGGL(ncc_stats_kernel, dim3(blocks(npx)), dim3(kTB), 0, 0, d_im1, H, W, patchsize, st1.p, (double *)nullptr);
  NccFast f{H, W, D, d_im0, d_im1, st0.p, st1.p, d_disp, d_out, layout};
  const dim3 grid((H + kNccRows - 1) / kNccRows, (D + kNccWaves - 1) / kNccWaves, (W + kNccCols - 1) / kNccCols);
  hipLaunchKernelGGL(ncc_cross_kernel, grid, dim3(kNccWaves * 64), 0, 0, f);
  STEREO_HIP_CHECK(hipGetLastError());
  STEREO_HIP_CHECK(hipDeviceSynchronize());  // st0 / st1 are released on return
}

}  // namespace
}  // namespace stereo

using namespace stereo;

extern "C" {

int stereo_pairwise_terms(int kernel, int64_t N, int64_t E, const uint32_t *conn, const double *points,
                          const double *assignment, const double *proposal, const double *weights, double tol,
                          double d_min, double d_step, double *E00, double *E01, double *E10, double *E11,
                          char *err, size_t errcap) {
  if (kernel != 1 && kernel != 2) return fail("Unkown kernel type", err, errcap);  // dispmap_super.m:233
  if (!conn || !points || !assignment || !weights || !E00 || (proposal && (!E01 || !E10 || !E11)))
    return fail("stereo_pairwise_terms: NULL argument", err, errcap);
  return guarded("stereo_pairwise_terms", err, errcap, [&] {
    check_planes(assignment, N);
    if (proposal) check_planes(proposal, N);
    DevBuf<uint32_t> dc; DevBuf<double> dp, da, dpr, dw, o0, o1, o2, o3;
    dc.upload(conn, 2 * E); dp.upload(points, 2 * N); da.upload(assignment, 4 * N); dw.upload(weights, E);
    if (proposal) dpr.upload(proposal, 4 * N);
    o0.alloc(E); o1.alloc(E); o2.alloc(E); o3.alloc(E);
    hipLaunchKernelGGL(pairwise_terms_kernel, dim3(blocks(E)), dim3(kTB), 0, 0, kernel, E, dc.p, dp.p, da.p,
                       proposal ? dpr.p : nullptr, dw.p, tol, d_min, d_step, o0.p, o1.p, o2.p, o3.p);
    download(E00, o0, E);
    if (proposal) { download(E01, o1, E); download(E10, o2, E); download(E11, o3, E); }
  });
}

int stereo_trws_positions(int64_t N, int64_t E, int K, const uint32_t *conn, const double *points,
                          const double *proposals, double d_min, double d_step, double *q, double *qprim,
                          char *err, size_t errcap) {
  if (!conn || !points || !proposals || !q || !qprim || K < 1)
    return fail("stereo_trws_positions: bad argument", err, errcap);
  return guarded("stereo_trws_positions", err, errcap, [&] {
    check_planes(proposals, N * K);
    DevBuf<uint32_t> dc; DevBuf<double> dp, dpr, oq, oqp;
    dc.upload(conn, 2 * E); dp.upload(points, 2 * N); dpr.upload(proposals, (size_t)4 * N * K);
    oq.alloc((size_t)E * K); oqp.alloc((size_t)E * K);
    hipLaunchKernelGGL(trws_positions_kernel, dim3(blocks(E * K)), dim3(kTB), 0, 0, E, K, N, dc.p, dp.p, dpr.p,
                       d_min, d_step, oq.p, oqp.p);
    download(q, oq, (size_t)E * K); download(qprim, oqp, (size_t)E * K);
  });
}

int stereo_ncc_volume(const double *im0, const double *im1, int H, int W, const double *disparities, int D,
                      int patchsize, int layout, double *ncc, char *err, size_t errcap) {
  if (!im0 || !im1 || !disparities || !ncc || H < 1 || W < 1 || D < 1 || patchsize < 0 || (layout != 0 && layout != 1))
    return fail("stereo_ncc_volume: bad argument", err, errcap);
  return guarded("stereo_ncc_volume", err, errcap, [&] {
    const size_t npx = (size_t)H * W;
    DevBuf<double> d0, d1, dd, out;
    d0.upload(im0, npx * 3); d1.upload(im1, npx * 3); dd.upload(disparities, D); out.alloc(npx * D);
    launch_ncc_volume(d0.p, d1.p, H, W, disparities, dd.p, D, patchsize, layout, out.p);
    download(ncc, out, npx * D);
  });
}

int stereo_ncc_unary(const double *ncc, int H, int W, int D, int layout, const double *disparities,
                     double unary_weight, const double *assignment, double *U, char *err, size_t errcap) {
  if (!ncc || !disparities || !assignment || !U || D < 1) return fail("stereo_ncc_unary: bad argument", err, errcap);
  return guarded("stereo_ncc_unary", err, errcap, [&] {
    const size_t npx = (size_t)H * W;
    check_planes(assignment, npx);
    double dmin = disparities[0], dmax = disparities[0];
    for (int i = 1; i < D; ++i) { dmin = std::min(dmin, disparities[i]); dmax = std::max(dmax, disparities[i]); }
    DevBuf<double> dn, dd, da, out;
    dn.upload(ncc, npx * D); dd.upload(disparities, D); da.upload(assignment, 4 * npx); out.alloc(npx);
    hipLaunchKernelGGL(ncc_unary_kernel, dim3(blocks(npx)), dim3(kTB), 0, 0, dn.p, H, W, D, layout, dd.p, dmin,
                       dmax, unary_weight, da.p, out.p, strictly_ascending(disparities, D) ? 1 : 0);
    download(U, out, npx);
  });
}

int stereo_ncc_best_disp(const double *ncc, int H, int W, int D, int layout, const double *disparities,
                         double *best, char *err, size_t errcap) {
  if (!ncc || !disparities || !best || D < 1) return fail("stereo_ncc_best_disp: bad argument", err, errcap);
  return guarded("stereo_ncc_best_disp", err, errcap, [&] {
    const size_t npx = (size_t)H * W;
    DevBuf<double> dn, dd, out;
    dn.upload(ncc, npx * D); dd.upload(disparities, D); out.alloc(npx);
    hipLaunchKernelGGL(ncc_best_disp_kernel, dim3(blocks(npx)), dim3(kTB), 0, 0, dn.p, H, W, D, layout, dd.p, out.p);
    download(best, out, npx);
  });
}

int stereo_globalstereo_unary(const double *im0, const double *im1, int H, int W, int C, const double *P2,
                              double d_min, double d_step, double col_thresh, const double *assignment,
                              double *U, char *err, size_t errcap) {
  if (!im0 || !im1 || !P2 || !assignment || !U || C < 1) return fail("stereo_globalstereo_unary: bad argument", err, errcap);
  return guarded("stereo_globalstereo_unary", err, errcap, [&] {
    const size_t npx = (size_t)H * W;
    check_planes(assignment, npx);
    DevBuf<double> d0, d1, dP, da, out;
    d0.upload(im0, npx * C); d1.upload(im1, npx * C); dP.upload(P2, 12); da.upload(assignment, 4 * npx);
    out.alloc(npx);
    hipLaunchKernelGGL(globalstereo_unary_kernel, dim3(blocks(npx)), dim3(kTB), 0, 0, d0.p, d1.p, H, W, C, dP.p,
                       d_min, d_step, col_thresh, da.p, out.p);
    download(U, out, npx);
  });
}

}  // extern "C"

// ---------------------------------------------------------------- fusion context

struct stereo_fusion {
  int H = 0, W = 0, kernel = 1;
  int64_t N = 0, E = 0;
  double tol = 0, d_min = 0, d_step = 0;
  DevBuf<uint32_t> conn;
  DevBuf<double> points, weights, cur, prop, Ucur, U1, E00, E01, E10, E11, partial;
  DevBuf<uint8_t> take;
  stereo_rd_plan *rd = nullptr;
  stereo_trws_plan *trws = nullptr;  // simultaneous fusion: plan of the last label count
  int trws_K = 0;
  DevBuf<double> props, unaryK, qK, qpK;
  DevBuf<int32_t> labels;
  std::vector<uint32_t> h_conn;
  std::vector<int32_t> h_label;
  // unary source: 1 = NCC volume (dispmap_ncc.m), 2 = photo-consistency (dispmap_globalstereo.m)
  int unary_kind = 0;
  DevBuf<double> ncc, disparities, im0, im1, P2;
  int D = 0, C = 0;
  double unary_weight = 0, dmin = 0, dmax = 0, col_thresh = 0;
  bool ascending = false;  // strictly ascending disparities: the sampler bisects instead of scanning
  DevBuf<double> best, plane_tab, fit_out;   // proposals built on the device: WTA disparities, 4 x S planes
  DevBuf<int32_t> segments;
  bool have_best = false, have_segments = false;
  bool have_assignment = false;
  double energy = 0;
  std::vector<double> h_lab;
  std::vector<uint8_t> h_take;
  ~stereo_fusion() {
    if (rd) stereo_rd_plan_destroy(rd);
    if (trws) stereo_trws_plan_destroy(trws);
  }
};

namespace {

double fusion_sum(stereo_fusion *F, const double *x, int64_t n) {
  if (n <= 0) return 0;
  const int64_t nb = (n + kTB * 8 - 1) / (kTB * 8);
  if ((int64_t)F->partial.n < nb) F->partial.alloc(nb);
  hipLaunchKernelGGL(fusion_sum_kernel, dim3((unsigned)nb), dim3(kTB), 0, 0, x, n, F->partial.p);
  std::vector<double> h(nb);
  STEREO_HIP_CHECK(hipMemcpy(h.data(), F->partial.p, sizeof(double) * nb, hipMemcpyDeviceToHost));
  double t = 0;
  for (int64_t i = 0; i < nb; ++i) t += h[i];
  return t;
}

void fusion_unary(stereo_fusion *F, const double *d_planes, double *d_out) {
  if (F->unary_kind == 1) {
    hipLaunchKernelGGL(ncc_unary_kernel, dim3(blocks(F->N)), dim3(kTB), 0, 0, F->ncc.p, F->H, F->W, F->D, 0,
                       F->disparities.p, F->dmin, F->dmax, F->unary_weight, d_planes, d_out, F->ascending ? 1 : 0);
  } else if (F->unary_kind == 2) {
    hipLaunchKernelGGL(globalstereo_unary_kernel, dim3(blocks(F->N)), dim3(kTB), 0, 0, F->im0.p, F->im1.p, F->H,
                       F->W, F->C, F->P2.p, F->d_min, F->d_step, F->col_thresh, d_planes, d_out);
  } else {
    throw std::runtime_error("Overload unary_cost");  // dispmap_super.m:200-203
  }
}

// dispmap_super.m:263-274 update_energy on the resident assignment
void fusion_update_energy(stereo_fusion *F) {
  hipLaunchKernelGGL(pairwise_terms_kernel, dim3(blocks(F->E)), dim3(kTB), 0, 0, F->kernel, F->E, F->conn.p,
                     F->points.p, F->cur.p, (const double *)nullptr, F->weights.p, F->tol, F->d_min, F->d_step,
                     F->E00.p, F->E01.p, F->E10.p, F->E11.p);
  // both sums with one copy back (same block partials, same order of addition as two fusion_sum calls)
  const int64_t N = F->N, E = F->E;
  const int64_t nbN = N > 0 ? (N + kTB * 8 - 1) / (kTB * 8) : 0, nbE = E > 0 ? (E + kTB * 8 - 1) / (kTB * 8) : 0;
  if ((int64_t)F->partial.n < nbN + nbE) F->partial.alloc((size_t)(nbN + nbE));
  if (nbN) hipLaunchKernelGGL(fusion_sum_kernel, dim3((unsigned)nbN), dim3(kTB), 0, 0, F->Ucur.p, N, F->partial.p);
  if (nbE) hipLaunchKernelGGL(fusion_sum_kernel, dim3((unsigned)nbE), dim3(kTB), 0, 0, F->E00.p, E, F->partial.p + nbN);
  std::vector<double> h((size_t)(nbN + nbE));
  if (!h.empty()) STEREO_HIP_CHECK(hipMemcpy(h.data(), F->partial.p, sizeof(double) * h.size(), hipMemcpyDeviceToHost));
  double tu = 0, te = 0;
  for (int64_t i = 0; i < nbN; ++i) tu += h[i];
  for (int64_t i = 0; i < nbE; ++i) te += h[nbN + i];
  F->energy = tu + te;
}

}  // namespace

extern "C" {

int stereo_fusion_create(int H, int W, int kernel, double tol, int64_t E, const uint32_t *conn,
                         const double *weights, double d_min, double d_step, stereo_fusion **ctx, char *err,
                         size_t errcap) {
  if (!ctx) return fail("stereo_fusion_create: ctx is NULL", err, errcap);
  *ctx = nullptr;
  if (kernel != 1 && kernel != 2) return fail("Unkown kernel type", err, errcap);  // dispmap_super.m:233
  if (H < 1 || W < 1 || E < 0 || (E > 0 && (!conn || !weights))) return fail("stereo_fusion_create: bad argument", err, errcap);
  std::unique_ptr<stereo_fusion> F(new stereo_fusion);
  const int rc = guarded("stereo_fusion_create", err, errcap, [&] {
    F->H = H; F->W = W; F->kernel = kernel; F->tol = tol; F->E = E; F->N = (int64_t)H * W;
    F->d_min = d_min; F->d_step = d_step;
    const int64_t N = F->N;
    std::vector<double> pts(2 * N);
    for (int64_t i = 0; i < N; ++i) { pts[2 * i] = (double)(i / H + 1); pts[2 * i + 1] = (double)(i % H + 1); }  // :275-278
    F->conn.upload(conn, 2 * E); F->weights.upload(weights, E); F->points.upload(pts.data(), 2 * N);
    F->h_conn.assign(conn, conn + 2 * E);
    F->cur.alloc(4 * N); F->prop.alloc(4 * N); F->Ucur.alloc(N); F->U1.alloc(N); F->take.alloc(N);
    F->E00.alloc(std::max<int64_t>(E, 1)); F->E01.alloc(std::max<int64_t>(E, 1));
    F->E10.alloc(std::max<int64_t>(E, 1)); F->E11.alloc(std::max<int64_t>(E, 1));
    F->h_lab.resize(N); F->h_take.resize(N);
    char e2[256] = {0};
    if (stereo_rd_plan_create(N, E, conn, &F->rd, e2, sizeof(e2)) != 0) throw std::runtime_error(e2);
    if (stereo_rd_plan_set_grid(F->rd, H, W, e2, sizeof(e2)) != 0) throw std::runtime_error(e2);
  });
  if (rc == 0) *ctx = F.release();
  return rc;
}

void stereo_fusion_destroy(stereo_fusion *ctx) { delete ctx; }

int stereo_fusion_unary_ncc(stereo_fusion *F, const double *ncc, int D, const double *disparities,
                            double unary_weight, char *err, size_t errcap) {
  if (!F || !ncc || !disparities || D < 1) return fail("stereo_fusion_unary_ncc: bad argument", err, errcap);
  return guarded("stereo_fusion_unary_ncc", err, errcap, [&] {
    F->ncc.upload(ncc, (size_t)F->N * D); F->disparities.upload(disparities, D);
    F->D = D; F->unary_weight = unary_weight; F->unary_kind = 1;
    F->ascending = strictly_ascending(disparities, D);
    F->have_best = false;
    F->dmin = F->dmax = disparities[0];
    for (int i = 1; i < D; ++i) { F->dmin = std::min(F->dmin, disparities[i]); F->dmax = std::max(F->dmax, disparities[i]); }
    F->have_assignment = false;
  });
}

int stereo_fusion_unary_globalstereo(stereo_fusion *F, const double *im0, const double *im1, int C,
                                     const double *P2, double col_thresh, char *err, size_t errcap) {
  if (!F || !im0 || !im1 || !P2 || C < 1) return fail("stereo_fusion_unary_globalstereo: bad argument", err, errcap);
  return guarded("stereo_fusion_unary_globalstereo", err, errcap, [&] {
    F->im0.upload(im0, (size_t)F->N * C); F->im1.upload(im1, (size_t)F->N * C); F->P2.upload(P2, 12);
    F->C = C; F->col_thresh = col_thresh; F->unary_kind = 2;
    F->have_assignment = false;
  });
}

int stereo_fusion_set_assignment(stereo_fusion *F, const double *assignment, double *energy, char *err,
                                 size_t errcap) {
  if (!F || !assignment) return fail("stereo_fusion_set_assignment: bad argument", err, errcap);
  return guarded("stereo_fusion_set_assignment", err, errcap, [&] {
    check_planes(assignment, F->N);
    F->cur.upload(assignment, 4 * F->N);
    fusion_unary(F, F->cur.p, F->Ucur.p);
    fusion_update_energy(F);
    F->have_assignment = true;
    if (energy) *energy = F->energy;
  });
}

int stereo_fusion_get_assignment(stereo_fusion *F, double *assignment, double *energy, char *err, size_t errcap) {
  if (!F) return fail("stereo_fusion_get_assignment: bad argument", err, errcap);
  if (!F->have_assignment) return fail("stereo_fusion: no assignment set", err, errcap);
  return guarded("stereo_fusion_get_assignment", err, errcap, [&] {
    if (assignment) download(assignment, F->cur, 4 * F->N);
    if (energy) *energy = F->energy;
  });
}

// the 4 x S plane table (+ segment ids) of a proposal -> 4 x N on the device
static void fusion_build_proposal(stereo_fusion *F, const double *planes, int S, const int32_t *segments, double *d_prop) {
  check_planes(planes, S);
  if (S > 1 && !segments && !F->have_segments) throw std::runtime_error("stereo_fusion: more than one plane needs segment ids");
  if (segments) { F->segments.upload(segments, F->N); F->have_segments = true; }
  F->plane_tab.upload(planes, (size_t)4 * S);
  hipLaunchKernelGGL(proposal_from_planes_kernel, dim3(blocks(F->N)), dim3(kTB), 0, 0, F->N, F->plane_tab.p, S,
                     S > 1 ? F->segments.p : (const int32_t *)nullptr, d_prop);
}

// one binary fusion move with the proposal already in F->prop
static void fusion_binary_core(stereo_fusion *F, int improve, double *energy, double *rd_energy, double *lower_bound,
                               double *num_unlabelled, double t_start) {
  {
    const int64_t N = F->N, E = F->E;
    const bool verbose = std::getenv("STEREO_HIP_FUSION_VERBOSE") != nullptr;
    auto now = [] { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
    double t[6];
    t[0] = t_start;
    hipLaunchKernelGGL(pairwise_terms_kernel, dim3(blocks(E)), dim3(kTB), 0, 0, F->kernel, E, F->conn.p, F->points.p,
                       F->cur.p, F->prop.p, F->weights.p, F->tol, F->d_min, F->d_step, F->E00.p, F->E01.p,
                       F->E10.p, F->E11.p);
    fusion_unary(F, F->prop.p, F->U1.p);
    if (verbose) STEREO_HIP_CHECK(hipDeviceSynchronize());
    t[1] = now();
    double e = 0, lb = 0, unl = 0;
    char e2[256] = {0};
    if (stereo_rd_plan_solve_device(F->rd, F->Ucur.p, F->U1.p, F->E00.p, F->E01.p, F->E10.p, F->E11.p, improve,
                                    nullptr, &e, &lb, &unl, e2, sizeof(e2)) != 0)  // labels stay on the device
      throw std::runtime_error(e2);
    t[2] = now();
    hipLaunchKernelGGL(fusion_scatter_kernel, dim3(blocks(N)), dim3(kTB), 0, 0, N, stereo_rd_plan_device_labels(F->rd),
                       F->prop.p, F->U1.p, F->cur.p, F->Ucur.p);
    if (verbose) STEREO_HIP_CHECK(hipDeviceSynchronize());
    t[3] = now();
    fusion_update_energy(F);
    t[4] = now();
    if (verbose)
      std::fprintf(stderr, "[stereo_hip fusion] upload+terms %.2f ms, qpbo %.2f ms, scatter %.2f ms, energy %.2f ms\n",
                   t[1] - t[0], t[2] - t[1], t[3] - t[2], t[4] - t[3]);
    if (energy) *energy = F->energy;
    if (rd_energy) *rd_energy = e;
    if (lower_bound) *lower_bound = lb;
    if (num_unlabelled) *num_unlabelled = unl;
  }
}

static double wall_ms() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

int stereo_fusion_binary(stereo_fusion *F, const double *proposal, int improve, double *energy,
                         double *rd_energy, double *lower_bound, double *num_unlabelled, char *err,
                         size_t errcap) {
  if (!F || !proposal) return fail("stereo_fusion_binary: bad argument", err, errcap);
  if (!F->have_assignment) return fail("stereo_fusion: no assignment set", err, errcap);
  return guarded("stereo_fusion_binary", err, errcap, [&] {
    const double t0 = wall_ms();
    check_planes(proposal, F->N);
    F->prop.upload(proposal, 4 * F->N);
    fusion_binary_core(F, improve, energy, rd_energy, lower_bound, num_unlabelled, t0);
  });
}

int stereo_fusion_binary_planes(stereo_fusion *F, const double *planes, int S, const int32_t *segments, int improve,
                                double *energy, double *rd_energy, double *lower_bound, double *num_unlabelled,
                                char *err, size_t errcap) {
  if (!F || !planes || S < 1) return fail("stereo_fusion_binary_planes: bad argument", err, errcap);
  if (!F->have_assignment) return fail("stereo_fusion: no assignment set", err, errcap);
  return guarded("stereo_fusion_binary_planes", err, errcap, [&] {
    const double t0 = wall_ms();
    fusion_build_proposal(F, planes, S, segments, F->prop.p);
    fusion_binary_core(F, improve, energy, rd_energy, lower_bound, num_unlabelled, t0);
  });
}

// binary_fuse_until_convergence (dispmap_super.m:85-152) as ONE call: the n proposals go to the device
// once (4 x N x n arrays, or a 4 x n table of single planes that are expanded there), the revisit
// schedule `ids` (1-based, what the reference builds from 1:n and its randi draws, with its
// `ids([diff(ids) == 0]) = 0` filter already applied by the caller) is walked here with the reference's
// loop -- the loop variable bumped inside the body, the exact `~=` on consecutive energies, the
// `visited` bookkeeping -- and every move runs on the resident state; nothing of size N crosses PCIe
// between moves and no interpreter sits between them.  energies: E(1) = energy before the first move,
// then one entry per move (at most cap; n_energies = length(E) = what the reference returns).
int stereo_fusion_fuse_until_convergence(stereo_fusion *F, const double *proposals, const double *planes, int n,
                                         const int64_t *ids, int64_t n_ids, int64_t maxiter, int improve,
                                         double *n_energies, double *energies, int64_t cap, char *err, size_t errcap) {
  if (!F || (!proposals && !planes) || n < 1 || !ids || n_ids < 0 || maxiter < 0)
    return fail("stereo_fusion_fuse_until_convergence: bad argument", err, errcap);
  if (!F->have_assignment) return fail("stereo_fusion: no assignment set", err, errcap);
  for (int64_t i = 0; i < n_ids; ++i)
    if (ids[i] < 1 || ids[i] > n) return fail("stereo_fusion_fuse_until_convergence: proposal id out of range", err, errcap);
  return guarded("stereo_fusion_fuse_until_convergence", err, errcap, [&] {
    const int64_t N = F->N;
    // Where a move's proposal comes from: a single plane is expanded on the device when its move comes
    // (32 bytes of input, no storage); 4 x N arrays stay resident for the whole schedule only while all
    // n of them fit a budget (STEREO_HIP_FUSION_RESIDENT_MB, default 4096 MB and never more than a quarter
    // of the free device memory) -- beyond that each move uploads its own proposal from the caller's
    // buffer, as the per-move entry point does (a 3000 x 2000 image is 192 MB per proposal: a schedule
    // over a few dozen proposals must not need tens of GB up front).
    DevBuf<double> all;
    bool resident = false;
    if (proposals) {
      check_planes(proposals, N * n);
      size_t budget = (size_t)4096 << 20, free_b = 0, total_b = 0;
      if (const char *mb = std::getenv("STEREO_HIP_FUSION_RESIDENT_MB")) budget = (size_t)std::max(0L, std::atol(mb)) << 20;
      if (hipMemGetInfo(&free_b, &total_b) == hipSuccess) budget = std::min(budget, free_b / 4);
      const size_t need = sizeof(double) * 4 * (size_t)N * (size_t)n;
      resident = need <= budget;
      if (resident) {
        all.alloc((size_t)4 * N * n);
        STEREO_HIP_CHECK(hipMemcpy(all.p, proposals, need, hipMemcpyHostToDevice));
      }
    } else {
      check_planes(planes, n);
    }
    std::vector<double> E(1, F->energy);
    std::vector<char> visited((size_t)n, 0);
    for (int64_t it = 1; it <= maxiter; ++it) {
      const int64_t it1 = it + 1;                    // iter = iter + 1 inside the for body (dispmap_super.m:108)
      if (it1 > n_ids) break;
      const int64_t pid = ids[it1 - 1];
      if (visited[pid - 1]) continue;
      if (!proposals) fusion_build_proposal(F, planes + 4 * (pid - 1), 1, nullptr, F->prop.p);
      else if (resident) STEREO_HIP_CHECK(hipMemcpyAsync(F->prop.p, all.p + (size_t)4 * N * (pid - 1), sizeof(double) * 4 * N, hipMemcpyDeviceToDevice, 0));
      else F->prop.upload(proposals + (size_t)4 * N * (pid - 1), 4 * N);
      fusion_binary_core(F, improve, nullptr, nullptr, nullptr, nullptr, wall_ms());
      E.push_back(F->energy);
      if (E[E.size() - 2] != E.back()) std::fill(visited.begin(), visited.end(), 0);   // E(end-1) ~= E(end)
      else visited[pid - 1] = 1;
      if (std::all_of(visited.begin(), visited.end(), [](char v) { return v != 0; })) break;
    }
    if (n_energies) *n_energies = (double)E.size();
    if (energies) for (int64_t i = 0; i < cap && i < (int64_t)E.size(); ++i) energies[i] = E[i];
  });
}

int stereo_fusion_fit_planes(stereo_fusion *F, const double *xs, const double *ys, int n, double r, double *planes,
                             double *npoints, char *err, size_t errcap) {
  if (!F || !planes || !xs || !ys || n < 1) return fail("stereo_fusion_fit_planes: bad argument", err, errcap);
  if (F->unary_kind != 1) return fail("stereo_fusion_fit_planes: needs the NCC volume (dispmap_ncc)", err, errcap);
  if (!(r > 0) || (2 * r + 3) * (2 * r + 3) > kFitMax) return fail("stereo_fusion_fit_planes: radius out of range", err, errcap);
  return guarded("stereo_fusion_fit_planes", err, errcap, [&] {
    if (!F->have_best) {  // best_disp_from_ncc (dispmap_ncc.m:208-221), once per volume
      F->best.alloc(F->N);
      hipLaunchKernelGGL(ncc_best_disp_kernel, dim3(blocks(F->N)), dim3(kTB), 0, 0, F->ncc.p, F->H, F->W, F->D, 0,
                         F->disparities.p, F->best.p);
      F->have_best = true;
    }
    std::vector<double> xy(2 * (size_t)n), h(5 * (size_t)n);
    for (int i = 0; i < n; ++i) { xy[2 * i] = xs[i]; xy[2 * i + 1] = ys[i]; }
    DevBuf<double> d_xy;
    d_xy.upload(xy.data(), xy.size());
    if (F->fit_out.n < h.size()) F->fit_out.alloc(h.size());
    hipLaunchKernelGGL(fit_plane_kernel, dim3((unsigned)n), dim3(64), 0, 0, F->best.p, F->H, F->W, d_xy.p, r, F->kernel, F->fit_out.p);
    STEREO_HIP_CHECK(hipMemcpy(h.data(), F->fit_out.p, sizeof(double) * h.size(), hipMemcpyDeviceToHost));
    for (int i = 0; i < n; ++i) {
      for (int k = 0; k < 4; ++k) planes[4 * i + k] = h[5 * i + k];
      if (npoints) npoints[i] = h[5 * i + 4];
    }
  });
}

int stereo_fusion_fit_plane(stereo_fusion *F, double x, double y, double r, double *plane, double *npoints, char *err,
                            size_t errcap) {
  return stereo_fusion_fit_planes(F, &x, &y, 1, r, plane, npoints, err, errcap);
}

// proposals: 4 x N x K host array, or NULL with plane_tab = 4 x K (one plane per proposal, built on the device)
static int fusion_simultaneous_impl(stereo_fusion *F, const double *proposals, const double *plane_tab, int K,
                                    double maxiter, double max_relgap, double *energy, double *trws_energy,
                                    double *lower_bound, double *iterations, char *err, size_t errcap) {
  return guarded("stereo_fusion_simultaneous", err, errcap, [&] {
    const int64_t N = F->N, E = F->E;
    const int Kt = K + 1;  // proposals{end+1} = self.assignment (dispmap_super.m:160)
    const bool verbose = std::getenv("STEREO_HIP_FUSION_VERBOSE") != nullptr;
    auto now = [] { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
    const double t0 = now();
    if (proposals) check_planes(proposals, N * K); else check_planes(plane_tab, K);
    char e2[256] = {0};
    if (!F->trws || F->trws_K != Kt) {
      if (F->trws) { stereo_trws_plan_destroy(F->trws); F->trws = nullptr; }
      if (stereo_trws_plan_create(F->kernel, Kt, N, E, F->h_conn.data(), STEREO_TRWS_MESSAGES_EXACT, &F->trws, e2,
                                  sizeof(e2)) != 0)
        throw std::runtime_error(e2);
      F->trws_K = Kt;
      F->props.alloc((size_t)4 * N * Kt); F->unaryK.alloc((size_t)N * Kt);
      F->qK.alloc((size_t)std::max<int64_t>(E, 1) * Kt); F->qpK.alloc((size_t)std::max<int64_t>(E, 1) * Kt);
      F->labels.alloc(N); F->h_label.resize(N);
    }
    if (proposals) {
      STEREO_HIP_CHECK(hipMemcpy(F->props.p, proposals, sizeof(double) * 4 * N * K, hipMemcpyHostToDevice));
    } else {
      F->plane_tab.upload(plane_tab, (size_t)4 * K);
      for (int k = 0; k < K; ++k)
        hipLaunchKernelGGL(proposal_from_planes_kernel, dim3(blocks(N)), dim3(kTB), 0, 0, N, F->plane_tab.p + 4 * k, 1,
                           (const int32_t *)nullptr, F->props.p + (size_t)k * 4 * N);
    }
    STEREO_HIP_CHECK(hipMemcpy(F->props.p + (size_t)4 * N * K, F->cur.p, sizeof(double) * 4 * N, hipMemcpyDeviceToDevice));
    // unary (K x N, dispmap_super.m:164-172) and positions (:177-183) on the device
    for (int k = 0; k < Kt; ++k) {
      if (k < K) fusion_unary(F, F->props.p + (size_t)k * 4 * N, F->U1.p);
      hipLaunchKernelGGL(fusion_interleave_kernel, dim3(blocks(N)), dim3(kTB), 0, 0, N, Kt, k,
                         k < K ? F->U1.p : F->Ucur.p, F->unaryK.p);
    }
    if (E > 0)
      hipLaunchKernelGGL(trws_positions_kernel, dim3(blocks(E * Kt)), dim3(kTB), 0, 0, E, Kt, N, F->conn.p, F->points.p,
                         F->props.p, F->d_min, F->d_step, F->qK.p, F->qpK.p);
    STEREO_HIP_CHECK(hipDeviceSynchronize());
    const double t1 = now();
    if (stereo_trws_plan_bind_device(F->trws, F->unaryK.p, F->qK.p, F->qpK.p, nullptr, F->weights.p, F->tol, e2,
                                     sizeof(e2)) != 0 ||
        stereo_trws_plan_reset(F->trws, e2, sizeof(e2)) != 0)
      throw std::runtime_error(e2);
    const double t2 = now();
    // Minimize_TRW_S runs at least one iteration and stops at iter >= iterMax (minimize.cpp:100-112)
    int iters = maxiter >= 1 ? (maxiter > 2e9 ? 2000000000 : (int)maxiter) : 1;
    int done = 0, stopped = 0;
    if (stereo_trws_plan_iterate(F->trws, iters, max_relgap, nullptr, &done, &stopped, e2, sizeof(e2)) != 0)
      throw std::runtime_error(e2);
    const double t3 = now();
    double te = 0, tlb = 0, tit = 0;
    if (stereo_trws_plan_result(F->trws, F->h_lab.data(), &te, &tlb, &tit, e2, sizeof(e2)) != 0)
      throw std::runtime_error(e2);
    for (int64_t i = 0; i < N; ++i) F->h_label[i] = (int32_t)F->h_lab[i] - 1;
    F->labels.upload(F->h_label.data(), N);
    hipLaunchKernelGGL(fusion_select_kernel, dim3(blocks(N)), dim3(kTB), 0, 0, N, Kt, F->labels.p, F->props.p,
                       F->unaryK.p, F->cur.p, F->Ucur.p);
    fusion_update_energy(F);
    if (energy) *energy = F->energy;
    if (verbose) {
      int64_t serial = 0;
      (void)stereo_trws_plan_counters(F->trws, &serial, 1);
      std::fprintf(stderr, "[stereo_hip fusion] simultaneous K=%d: plan + unary + positions %.1f ms, bind (argsort) %.1f ms, %d iterations %.1f ms (path %d, %lld serial-envelope messages), scatter + energy %.1f ms\n",
                   Kt, t1 - t0, t2 - t1, done, t3 - t2, stereo_trws_plan_path(F->trws), (long long)serial, now() - t3);
    }
    if (trws_energy) *trws_energy = te;
    if (lower_bound) *lower_bound = tlb;
    if (iterations) *iterations = tit;
  });
}

int stereo_fusion_simultaneous(stereo_fusion *F, const double *proposals, int K, double maxiter,
                               double max_relgap, double *energy, double *trws_energy, double *lower_bound,
                               double *iterations, char *err, size_t errcap) {
  if (!F || !proposals || K < 1) return fail("stereo_fusion_simultaneous: bad argument", err, errcap);
  if (!F->have_assignment) return fail("stereo_fusion: no assignment set", err, errcap);
  return fusion_simultaneous_impl(F, proposals, nullptr, K, maxiter, max_relgap, energy, trws_energy, lower_bound,
                                  iterations, err, errcap);
}

int stereo_fusion_simultaneous_planes(stereo_fusion *F, const double *planes, int K, double maxiter, double max_relgap,
                                      double *energy, double *trws_energy, double *lower_bound, double *iterations,
                                      char *err, size_t errcap) {
  if (!F || !planes || K < 1) return fail("stereo_fusion_simultaneous_planes: bad argument", err, errcap);
  if (!F->have_assignment) return fail("stereo_fusion: no assignment set", err, errcap);
  return fusion_simultaneous_impl(F, nullptr, planes, K, maxiter, max_relgap, energy, trws_energy, lower_bound,
                                  iterations, err, errcap);
}

}  // extern "C"
