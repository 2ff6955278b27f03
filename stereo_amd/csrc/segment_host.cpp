// Image segmentation behind the globalstereo edge weights and the SegPln proposals: the HOST stages.
//
//   vgg_segment_ms(A, h_s, h_r, min_sz)            imrender/vgg/vgg_segment_ms.cxx:18-87 -> seg_ms/msImageProcessor.cpp
//   vgg_segment_gb(A, sigma, k, min_sz, compress)  imrender/vgg/vgg_segment_gb.cxx:21-87 -> seg_gb/segment-image.h
//
// Both segmenters are a per-pixel parallel stage followed by serial graph work whose ORDER is part of the result; the
// reference runs all of it on the host, once per image.  Here the per-pixel stages are kernels (segment.hip: the
// mean-shift filter -- 98 % of the reference's time --, Gaussian smoothing and edge weights) and this file holds the
// rest, written from the algorithms' definitions:
//   mean shift     RGB -> LUV (libm pow: the device's differs in the last place), the bucket lattice the filter walks,
//                  then, behind the filter: connected components of the filtered image, transitive closure of the
//                  region adjacency graph (twice or more), pruning of small regions          msImageProcessor.cpp:703-808
//   graph based    edges sorted by weight (std::sort, whose order of EQUAL weights is part of the result, exactly as
//                  for the gateway's sort of positions, trws_mex.cpp:16-20), Kruskal with the adaptive threshold,
//                  small components joined, first-appearance compression                      segment-graph.h:49-81
// Results equal the reference's label maps pixel for pixel (tests/test_segment_cpu.py, tests/test_segment_gpu.py;
// fixtures tests/golden/*_segments.npz made by the reference's own segmenters).
#include "segment_host.h"

#include <algorithm>
#include <cmath>
#include <cstring>
#include <unordered_map>

namespace stereo {
namespace seg {

// ---- mean shift: colour space (msImageProcessor.cpp:835-875; constants msImageProcessor.h:62-73) -------------------
void rgb_to_luv(const uint8_t *A, int H, int W, float *luv) {
  static const double M[3][3] = {{0.4125, 0.3576, 0.1804}, {0.2125, 0.7154, 0.0721}, {0.0193, 0.1192, 0.9502}};
  const double Yn = 1.0, Un = 0.19784977571475, Vn = 0.46834507665248, Lt = 0.008856;
  const size_t plane = (size_t)H * W;
  for (int y = 0; y < H; ++y)
    for (int x = 0; x < W; ++x) {
      // MATLAB's H x W x 3 column-major -> the library's row-major pixel y * W + x (vgg_segment_ms.cxx:40-50)
      const size_t src = (size_t)x * H + y;
      const double r = A[src], g = A[src + plane], b = A[src + 2 * plane];
      const double X = M[0][0] * r + M[0][1] * g + M[0][2] * b;
      const double Y = M[1][0] * r + M[1][1] * g + M[1][2] * b;
      const double Z = M[2][0] * r + M[2][1] * g + M[2][2] * b;
      float *o = luv + 3 * ((size_t)y * W + x);
      const double L0 = Y / (255.0 * Yn);
      o[0] = L0 > Lt ? (float)(116.0 * std::pow(L0, 1.0 / 3.0) - 16.0) : (float)(903.3 * L0);
      const double den = X + 15 * Y + 3 * Z;
      const double up = den != 0 ? (4 * X) / den : 4.0, vp = den != 0 ? (9 * Y) / den : 9.0 / 15.0;
      o[1] = (float)(13 * o[0] * (up - Un));
      o[2] = (float)(13 * o[0] * (vp - Vn));
    }
}

// ---- mean shift: the lattice of buckets the filter walks (msImageProcessor.cpp:3842-3933) ---------------------------
// Points scaled by the bandwidths, binned into unit cells of (x, y, L); a cell's points in the order the reference's
// linked lists give them (last inserted first = descending pixel index).  The filter adds a window's points bucket by
// bucket in the order of the 27 neighbour offsets, so this order is part of every sum.
void ms_lattice(const float *luv, int H, int W, int sigmaS_i, float sigmaR, MsLattice &lat) {
  const int64_t L = (int64_t)H * W;
  const float sigmaS = (float)sigmaS_i;
  lat.H = H; lat.W = W; lat.sigmaS = sigmaS; lat.sigmaR = sigmaR;
  lat.sdata.resize((size_t)5 * L);
  float *s = lat.sdata.data();
  for (int64_t i = 0; i < L; ++i) {
    s[5 * i + 0] = (int)(i % W) / sigmaS;
    s[5 * i + 1] = (int)(i / W) / sigmaS;
    s[5 * i + 2] = luv[3 * i + 0] / sigmaR;
    s[5 * i + 3] = luv[3 * i + 1] / sigmaR;
    s[5 * i + 4] = luv[3 * i + 2] / sigmaR;
  }
  float smin = s[2], smax = s[2];
  for (int64_t i = 0; i < L; ++i) {   // (the reference's else-if: a new minimum is never tested against the maximum)
    const float c = s[5 * i + 2];
    if (c < smin) smin = c;
    else if (c > smax) smax = c;
  }
  lat.smin = smin;
  lat.nb1 = (int)(W / sigmaS + 3);
  lat.nb2 = (int)(H / sigmaS + 3);
  lat.nb3 = (int)(smax - smin + 3);
  const int64_t nb = (int64_t)lat.nb1 * lat.nb2 * lat.nb3;
  lat.bucket_ptr.assign((size_t)nb + 1, 0);
  std::vector<int32_t> cell((size_t)L);
  for (int64_t i = 0; i < L; ++i) {
    const int c1 = (int)s[5 * i] + 1, c2 = (int)s[5 * i + 1] + 1, c3 = (int)(s[5 * i + 2] - smin) + 1;
    cell[i] = c1 + lat.nb1 * (c2 + lat.nb2 * c3);
    ++lat.bucket_ptr[(size_t)cell[i] + 1];
  }
  for (int64_t b = 0; b < nb; ++b) lat.bucket_ptr[b + 1] += lat.bucket_ptr[b];
  lat.bucket_items.resize((size_t)L);
  std::vector<int32_t> fill(lat.bucket_ptr.begin(), lat.bucket_ptr.end() - 1);
  for (int64_t i = L - 1; i >= 0; --i) lat.bucket_items[fill[cell[i]]++] = (int32_t)i;
  int n = 0;
  for (int a = -1; a <= 1; ++a)
    for (int b = -1; b <= 1; ++b)
      for (int c = -1; c <= 1; ++c) lat.neigh[n++] = a + lat.nb1 * (b + lat.nb2 * c);
}

// ---- mean shift: the filter's serial part ---------------------------------------------------------------------------------
// The reference filters pixels in scan order and lets a pixel take along ("basin of attraction") every pixel whose
// colour comes within speedThreshold of its trajectory; those inherit its mode and are skipped (msImageProcessor.cpp:
// 4015-4022, 4085-4127, 4194-4201, 4256-4270).  With the threshold the gateway ends up with (kSpeedThreshold,
// segment_host.h) that happens where colours are EQUAL: in flat image regions.  The device walks every pixel's whole
// trajectory on its own and reports, next to the pixel's own mode, whether any other pixel's colour ever came that
// close (`events`).  A pixel without such an event neither takes anybody along nor stops early: its result is its own
// mode, unless an earlier pixel took IT along.  The pixels with events -- and only those -- are walked here once more,
// in scan order, with the reference's bookkeeping.
namespace {

struct Walk {
  const MsLattice &lat;
  std::vector<uint8_t> mode_table;   // 0 untouched, 1 has its mode, 2 in the basin of the pixel being walked
  std::vector<int32_t> points;       // that basin
  float thr;
  double hiLTr;

  void window(const double *yk, double *Mh) {
    const float *sd = lat.sdata.data();
    double wsum = 0;
    for (int j = 0; j < 5; ++j) Mh[j] = 0;
    const int c1 = (int)yk[0] + 1, c2 = (int)yk[1] + 1, c3 = (int)(yk[2] - lat.smin) + 1;
    const int cb = c1 + lat.nb1 * (c2 + lat.nb2 * c3);
    for (int j = 0; j < 27; ++j) {
      const int b = cb + lat.neigh[j];
      for (int e = lat.bucket_ptr[b]; e < lat.bucket_ptr[b + 1]; ++e) {
        const int d = lat.bucket_items[e];
        const float *s = sd + 5 * (size_t)d;
        double el = s[0] - yk[0];
        double diff = el * el;
        el = s[1] - yk[1];
        diff += el * el;
        if (!(diff < 1.0)) continue;
        el = s[2] - yk[2];
        diff = yk[2] > hiLTr ? 4 * el * el : el * el;
        el = s[3] - yk[3];
        diff += el * el;
        el = s[4] - yk[4];
        diff += el * el;
        if (!(diff < 1.0)) continue;
        for (int k = 0; k < 5; ++k) Mh[k] += 1.0 * s[k];
        wsum += 1.0;
        if (diff < thr && mode_table[d] == 0) { points.push_back(d); mode_table[d] = 2; }
      }
    }
    if (wsum > 0) for (int j = 0; j < 5; ++j) Mh[j] = Mh[j] / wsum - yk[j];
    else for (int j = 0; j < 5; ++j) Mh[j] = 0;
  }
};

}  // namespace

int64_t ms_filter_finish(const MsLattice &lat, const float *own, const uint8_t *events, float thr, float *out) {
  const int64_t L = (int64_t)lat.H * lat.W;
  const int W = lat.W;
  const float sigmaS = lat.sigmaS, sigmaR = lat.sigmaR;
  const float *sd = lat.sdata.data();
  Walk w{lat, std::vector<uint8_t>((size_t)L, 0), {}, thr, 80.0 / sigmaR};
  int64_t walked = 0;
  for (int64_t i = 0; i < L; ++i) {
    if (w.mode_table[i] == 1) continue;
    if (!events[i]) {
      for (int k = 0; k < 3; ++k) out[3 * i + k] = own[3 * i + k];
      w.mode_table[i] = 1;
      continue;
    }
    ++walked;
    w.points.clear();
    double yk[5], Mh[5];
    for (int j = 0; j < 5; ++j) yk[j] = sd[5 * i + j];
    w.window(yk, Mh);
    double mv = (Mh[0] * Mh[0] + Mh[1] * Mh[1]) * sigmaS * sigmaS;
    mv += (Mh[2] * Mh[2] + Mh[3] * Mh[3] + Mh[4] * Mh[4]) * sigmaR * sigmaR;
    int iter = 1;
    while (mv >= 0.01 && iter < 100) {
      for (int j = 0; j < 5; ++j) yk[j] += Mh[j];
      const int cx = (int)(sigmaS * yk[0] + 0.5), cy = (int)(sigmaS * yk[1] + 0.5);
      const int64_t ci = (int64_t)cy * W + cx;
      if (w.mode_table[ci] != 2 && ci != i) {
        double diff = 0;
        for (int k = 2; k < 5; ++k) { const double el = sd[5 * ci + k] - yk[k]; diff += el * el; }
        if (diff < thr) {
          if (w.mode_table[ci] == 0) { w.points.push_back((int32_t)ci); w.mode_table[ci] = 2; }
          else {   // a pixel that has its mode already: this trajectory ends there
            for (int j = 0; j < 3; ++j) yk[j + 2] = out[3 * ci + j] / sigmaR;
            w.mode_table[i] = 1;
            mv = -1;
            break;
          }
        }
      }
      w.window(yk, Mh);
      mv = (Mh[0] * Mh[0] + Mh[1] * Mh[1]) * sigmaS * sigmaS;
      mv += (Mh[2] * Mh[2] + Mh[3] * Mh[3] + Mh[4] * Mh[4]) * sigmaR * sigmaR;
      ++iter;
    }
    if (mv >= 0) { for (int j = 0; j < 5; ++j) yk[j] += Mh[j]; w.mode_table[i] = 1; }
    float mode[3];
    for (int k = 0; k < 3; ++k) mode[k] = (float)(yk[k + 2] * sigmaR);
    for (int32_t c : w.points) {
      w.mode_table[c] = 1;
      for (int k = 0; k < 3; ++k) out[3 * (size_t)c + k] = mode[k];
    }
    for (int k = 0; k < 3; ++k) out[3 * i + k] = mode[k];
  }
  return walked;
}

// ---- mean shift: regions of the filtered image -----------------------------------------------------------------------
namespace {

struct Regions {
  int H = 0, W = 0, n = 0;
  float hr = 0;                    // range bandwidth h[1] (set by the filter, :3822), offset[1] == 1 (uniform kernel, ms.cpp:1222)
  std::vector<int32_t> labels;     // per pixel
  std::vector<float> modes;        // 3 per region
  std::vector<int32_t> counts;     // pixels per region
  std::vector<int32_t> adj_ptr, adj;
};

// Connect + Fill (msImageProcessor.cpp:1911-2060): regions = connected components of "every channel closer than
// LUV_treshold = 1" between pixels whose LINEAR indices differ by one of the eight neighbour offsets (the reference does
// not test for the image border, so the last pixel of a row neighbours the first of the next), numbered by their first
// pixel in scan order, which also gives the region its mode.
void connect(const float *f, Regions &R) {
  const int W = R.W, L = R.H * R.W;
  const int off[8] = {1, 1 - W, -W, -(1 + W), -1, W - 1, W, W + 1};
  R.labels.assign((size_t)L, -1);
  R.modes.clear(); R.counts.clear();
  std::vector<int32_t> stack;
  int label = -1;
  for (int i = 0; i < L; ++i) {
    if (R.labels[i] >= 0) continue;
    R.labels[i] = ++label;
    R.modes.insert(R.modes.end(), f + 3 * (size_t)i, f + 3 * (size_t)i + 3);
    int count = 1;
    stack.assign(1, i);
    while (!stack.empty()) {
      const int p = stack.back();
      stack.pop_back();
      for (int k = 0; k < 8; ++k) {
        const int q = p + off[k];
        if (q < 0 || q >= L || R.labels[q] >= 0) continue;
        const float *a = f + 3 * (size_t)p, *b = f + 3 * (size_t)q;
        if (std::fabs(a[0] - b[0]) >= 1.0f || std::fabs(a[1] - b[1]) >= 1.0f || std::fabs(a[2] - b[2]) >= 1.0f) continue;
        R.labels[q] = label;
        ++count;
        stack.push_back(q);
      }
    }
    R.counts.push_back(count);
  }
  R.n = label + 1;
}

// BuildRAM (:2085-2240): for every region the ascending list of the regions it touches to the right of or below one
// of its pixels (both ways).  Counted, filled and then sorted region by region (a region has a handful of neighbours):
// one global sort of all boundary pairs was most of this stage's time on the fine maps.
void adjacency(Regions &R) {
  const int H = R.H, W = R.W;
  std::vector<int32_t> &ptr = R.adj_ptr;
  ptr.assign((size_t)R.n + 1, 0);
  auto each_pair = [&](auto &&f) {
    for (int y = 0; y < H; ++y) {
      const int32_t *row = &R.labels[(size_t)y * W], *below = y + 1 < H ? row + W : nullptr;
      for (int x = 0; x < W; ++x) {
        const int c = row[x];
        if (x + 1 < W && row[x + 1] != c) f(c, row[x + 1]);
        if (below && below[x] != c) f(c, below[x]);
      }
    }
  };
  each_pair([&](int a, int b) { ++ptr[(size_t)a + 1]; ++ptr[(size_t)b + 1]; });
  for (int r = 0; r < R.n; ++r) ptr[r + 1] += ptr[r];
  std::vector<int32_t> raw((size_t)ptr[R.n]), at(ptr.begin(), ptr.end() - 1);
  each_pair([&](int a, int b) { raw[at[a]++] = b; raw[at[b]++] = a; });
  R.adj.clear();
  R.adj.reserve(raw.size());
  int32_t begin = 0;
  for (int r = 0; r < R.n; ++r) {
    const int32_t end = ptr[r + 1];
    std::sort(raw.begin() + begin, raw.begin() + end);
    const auto last = std::unique(raw.begin() + begin, raw.begin() + end);
    ptr[r] = (int32_t)R.adj.size();
    R.adj.insert(R.adj.end(), raw.begin() + begin, last);
    begin = end;
  }
  ptr[R.n] = (int32_t)R.adj.size();
}

int find_root(const std::vector<int32_t> &parent, int i) {
  while (parent[i] != i) i = parent[i];
  return i;
}
// (:2411-2431, :2806-2826: the smaller canonical index becomes the root)
void join_min(std::vector<int32_t> &parent, int a, int b) {
  a = find_root(parent, a); b = find_root(parent, b);
  if (a < b) parent[b] = a; else parent[a] = b;
}

// What both merging passes end with (:2444-2530, :2836-2912): modes of the merged regions (pixel-weighted means,
// accumulated in single precision in the order of the old labels), new labels by first appearance.
void merge(Regions &R, std::vector<int32_t> &parent) {
  const int n = R.n;
  for (int i = 0; i < n; ++i) parent[i] = find_root(parent, i);
  std::vector<float> mb((size_t)3 * n, 0.0f);
  std::vector<int32_t> cb((size_t)n, 0), relabel((size_t)n, -1);
  for (int i = 0; i < n; ++i) {
    const int c = parent[i], m = R.counts[i];
    for (int k = 0; k < 3; ++k) mb[3 * (size_t)c + k] += m * R.modes[3 * (size_t)i + k];
    cb[c] += m;
  }
  int label = -1;
  for (int i = 0; i < n; ++i) {
    const int c = parent[i];
    if (relabel[c] >= 0) continue;
    relabel[c] = ++label;
    for (int k = 0; k < 3; ++k) R.modes[3 * (size_t)label + k] = mb[3 * (size_t)c + k] / cb[c];
    R.counts[label] = cb[c];
  }
  R.n = label + 1;
  for (auto &l : R.labels) l = relabel[parent[l]];
}

// InWindow (:3157-3177): the range part of the kernel, lightness counted four times for bright modes (of the FIRST
// region: the test is not symmetric; the adjacency lists offer every pair both ways).
bool in_window(const Regions &R, int a, int b) {
  const float *m1 = &R.modes[3 * (size_t)a], *m2 = &R.modes[3 * (size_t)b];
  const float den = R.hr * 1.0f;
  double diff = 0;
  for (int p = 0; p < 3; ++p) {
    const double el = (m1[p] - m2[p]) / den;
    if (p == 0 && m1[0] > 80) diff += 4 * el * el; else diff += el * el;
  }
  return diff < 0.25;
}

// TransitiveClosure (:2349-2545): regions whose modes lie in each other's window are joined along the adjacency graph.
void transitive_closure(Regions &R) {
  adjacency(R);
  std::vector<int32_t> parent((size_t)R.n);
  for (int i = 0; i < R.n; ++i) parent[i] = i;
  for (int i = 0; i < R.n; ++i)
    for (int k = R.adj_ptr[i]; k < R.adj_ptr[i + 1]; ++k)
      if (in_window(R, i, R.adj[k])) join_min(parent, i, R.adj[k]);   // (edge strengths are zero without a weight map: 0 < epsilon = 1)
  merge(R, parent);
}

// SqDistance (:3196-3216), single precision throughout.
float sq_distance(const Regions &R, int a, int b) {
  const float *m1 = &R.modes[3 * (size_t)a], *m2 = &R.modes[3 * (size_t)b];
  const float den = R.hr * 1.0f;
  float dist = 0;
  for (int p = 0; p < 3; ++p) { const float el = (m1[p] - m2[p]) / den; dist += el * el; }
  return dist;
}

// Prune (:2734-2934): every region below minRegion pixels joins the adjacent region with the closest mode (the first
// of equally close ones in ascending label order), all at once per pass, until a pass finds no small region.
void prune(Regions &R, int min_region) {
  std::vector<int32_t> parent;
  for (;;) {
    int small = 0;
    adjacency(R);
    parent.resize((size_t)R.n);
    for (int i = 0; i < R.n; ++i) parent[i] = i;
    for (int i = 0; i < R.n; ++i) {
      if (R.counts[i] >= min_region) continue;
      ++small;
      if (R.adj_ptr[i] == R.adj_ptr[i + 1]) continue;   // (one region covers the image: the reference would read a null list here)
      int cand = R.adj[R.adj_ptr[i]];
      double best = sq_distance(R, i, cand);
      for (int k = R.adj_ptr[i] + 1; k < R.adj_ptr[i + 1]; ++k) {
        const double d = sq_distance(R, i, R.adj[k]);
        if (d < best) { best = d; cand = R.adj[k]; }
      }
      join_min(parent, i, cand);
    }
    merge(R, parent);
    if (small == 0 || R.n <= 1) break;
  }
}

}  // namespace

// Segment() behind the filter (msImageProcessor.cpp:480, 733-808): filtered = msRawData, H * W x 3, row-major pixels.
void ms_regions(const float *filtered, int H, int W, float sigmaR, int min_region, int32_t *labels) {
  Regions R;
  R.H = H; R.W = W; R.hr = sigmaR;
  connect(filtered, R);
  transitive_closure(R);
  // (:745-753: once more, and then again for as long as a pass merges NOTHING, ten times at most -- the reference's
  //  loop condition; a pass that merges nothing changes nothing, so those repetitions cost time only)
  int old_count = R.n, counter = 0, delta;
  do {
    transitive_closure(R);
    delta = old_count - R.n;
    old_count = R.n;
    ++counter;
  } while (delta <= 0 && counter < 10);
  prune(R, min_region);
  std::memcpy(labels, R.labels.data(), sizeof(int32_t) * (size_t)H * W);
}

// ---- graph based ---------------------------------------------------------------------------------------------------------
// filter.h:33-62: the Gaussian mask (sigma at least 0.01, four sigmas wide, single precision, normalised).
std::vector<float> gb_mask(float sigma) {
  sigma = std::max(sigma, 0.01F);
  const int len = (int)std::ceil(sigma * 4.0) + 1;
  std::vector<float> mask((size_t)len);
  for (int i = 0; i < len; ++i) {
    const float q = i / sigma;
    mask[i] = (float)std::exp(-0.5 * (q * q));
  }
  float sum = 0;
  for (int i = 1; i < len; ++i) sum += std::fabs(mask[i]);
  sum = 2 * sum + std::fabs(mask[0]);
  for (int i = 0; i < len; ++i) mask[i] /= sum;
  return mask;
}

namespace {
struct GbEdge {
  float w;
  int32_t a, b;
};
inline bool operator<(const GbEdge &x, const GbEdge &y) { return x.w < y.w; }
}  // namespace

// segment-image.h:181-247 behind the smoothing, segment-graph.h:49-81, vgg_segment_gb.cxx:57-84.
// weights: 4 per pixel y * W + x (row-major): to (x+1, y), (x, y+1), (x+1, y+1), (x+1, y-1); entries of edges that leave
// the image are ignored.  out: H x W, column-major, zero where the library writes nothing (its last row and column).
void gb_regions(const float *weights, int H, int W, float c, int min_size, int compress, uint32_t *out) {
  const int width = W, height = H;
  std::vector<GbEdge> edges;
  edges.reserve((size_t)4 * H * W);
  for (int y = 0; y < height; ++y)
    for (int x = 0; x < width; ++x) {
      const float *w4 = weights + 4 * ((size_t)y * width + x);
      const int a = y * width + x;
      if (x < width - 1) edges.push_back({w4[0], a, y * width + (x + 1)});
      if (y < height - 1) edges.push_back({w4[1], a, (y + 1) * width + x});
      if (x < width - 1 && y < height - 1) edges.push_back({w4[2], a, (y + 1) * width + (x + 1)});
      if (x < width - 1 && y > 0) edges.push_back({w4[3], a, (y - 1) * width + (x + 1)});
    }
  std::sort(edges.begin(), edges.end());   // (the order of equal weights is std::sort's, as in the reference)
  const int n = width * height;
  // disjoint-set.h: union by rank, ties to the second argument -- kept exactly: without compression the root's index IS
  // the label
  std::vector<int32_t> parent((size_t)n), rank((size_t)n, 0), size((size_t)n, 1);
  for (int i = 0; i < n; ++i) parent[i] = i;
  auto find = [&](int x) {
    int y = x;
    while (y != parent[y]) y = parent[y];
    parent[x] = y;
    return y;
  };
  auto join = [&](int x, int y) {
    if (rank[x] > rank[y]) { parent[y] = x; size[x] += size[y]; }
    else { parent[x] = y; size[y] += size[x]; if (rank[x] == rank[y]) ++rank[y]; }
  };
  std::vector<float> threshold((size_t)n, c / 1);
  for (const GbEdge &e : edges) {
    int a = find(e.a);
    const int b = find(e.b);
    if (a != b && e.w <= threshold[a] && e.w <= threshold[b]) {
      join(a, b);
      a = find(a);
      threshold[a] = e.w + c / size[a];
    }
  }
  for (const GbEdge &e : edges) {
    const int a = find(e.a), b = find(e.b);
    if (a != b && (size[a] < min_size || size[b] < min_size)) join(a, b);
  }
  std::memset(out, 0, sizeof(uint32_t) * (size_t)H * W);
  for (int y = 0; y < height - 1; ++y)
    for (int x = 0; x < width - 1; ++x) out[(size_t)x * height + y] = (uint32_t)find(y * width + x);
  if (compress) {   // first appearance in column-major order, from 1 (the zeros of the unwritten border are a value like any other)
    std::unordered_map<uint32_t, uint32_t> seen;
    for (size_t i = 0; i < (size_t)H * W; ++i) {
      auto it = seen.find(out[i]);
      if (it == seen.end()) it = seen.emplace(out[i], (uint32_t)seen.size() + 1).first;
      out[i] = it->second;
    }
  }
}

}  // namespace seg
}  // namespace stereo
