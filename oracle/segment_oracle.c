/* TEST INFRASTRUCTURE ONLY -- not part of the product path.
 *
 * CPU restatement of the mean-shift FILTER of the reference's segmenter (EDISON, vendored under
 * imrender/vgg/seg_ms), the stage the product runs on the device (stereo_amd/csrc/segment.hip):
 *
 *   msImageProcessor.cpp:835-875    RGBtoLUV                      -> oracle_rgb_to_luv
 *   msImageProcessor.cpp:3803-4303  NewOptimizedFilter2           -> oracle_ms_filter
 *     (what Filter(sigmaS, sigmaR, HIGH_SPEEDUP) runs, :345-503; vgg_segment_ms.cxx:71)
 *
 * Pinned by the reference itself: tests/test_segment_cpu.py runs the reference's whole segmenter
 * (oracle/_ref/libref_segment_ms.so, where the build container made it) and this filter followed by the
 * product's host stages on the same images and compares the label maps, next to the committed fixtures
 * tests/golden/{teddy,baby2}_segments.npz.
 *
 * `speed_threshold` is the reference's msImageProcessor::speedThreshold (msImageProcessor.h:796).  Its
 * constructor (:64-108) never sets it and vgg_segment_ms.cxx never calls SetSpeedThreshold: the gateway
 * reads an uninitialised float.  In the reference's own build here (oracle/_ref, and a probe that printed
 * the member) the object lies on the stack and the float holds the upper half of a stack address
 * (0x00007ffc = 4.59e-41): a positive value below every squared distance that is not exactly zero.  The
 * "basin of attraction" shortcuts of HIGH_SPEEDUP (:4015-4022, :4096-4124, :4194-4201) therefore fire
 * exactly where a pixel's colour EQUALS the trajectory's -- flat image regions: the first pixel in scan
 * order takes the equal pixels of its window along, they inherit its mode and are skipped.  The fixtures
 * tests/golden/{teddy,baby2}_segments.npz were made by that build; any threshold in (0, 1e-30] reproduces them, 0
 * (no shortcut at all) changes 5 of 25074 regions on the Teddy image.  kSpeedThreshold in the product
 * (csrc/segment_host.h) is that value.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

/* msImageProcessor.h:62-73 */
static const double kXYZ[3][3] = {{0.4125, 0.3576, 0.1804}, {0.2125, 0.7154, 0.0721}, {0.0193, 0.1192, 0.9502}};
static const double kYn = 1.0, kUn = 0.19784977571475, kVn = 0.46834507665248, kLt = 0.008856;

/* rgb: L x 3 bytes (row-major pixels, interleaved), luv: L x 3 floats.  msImageProcessor.cpp:835-875 */
void oracle_rgb_to_luv(const uint8_t *rgb, int64_t L, float *luv) {
  for (int64_t i = 0; i < L; ++i) {
    const uint8_t *c = rgb + 3 * i;
    float *o = luv + 3 * i;
    const double x = kXYZ[0][0] * c[0] + kXYZ[0][1] * c[1] + kXYZ[0][2] * c[2];
    const double y = kXYZ[1][0] * c[0] + kXYZ[1][1] * c[1] + kXYZ[1][2] * c[2];
    const double z = kXYZ[2][0] * c[0] + kXYZ[2][1] * c[1] + kXYZ[2][2] * c[2];
    const double L0 = y / (255.0 * kYn);
    if (L0 > kLt) o[0] = (float)(116.0 * pow(L0, 1.0 / 3.0) - 16.0);
    else o[0] = (float)(903.3 * L0);
    const double den = x + 15 * y + 3 * z;
    double up, vp;
    if (den != 0) { up = (4 * x) / den; vp = (9 * y) / den; }
    else { up = 4.0; vp = 9.0 / 15.0; }
    o[1] = (float)(13 * o[0] * (up - kUn));
    o[2] = (float)(13 * o[0] * (vp - kVn));
  }
}

/* One window pass (msImageProcessor.cpp:3955-4040 and again :4130-4215): mean of the lattice points whose
 * spatial and range distances to yk are below one bandwidth, walked bucket by bucket in the order of the
 * 27 offsets and, inside a bucket, along its list (last inserted first).  Returns the shift in Mh. */
typedef struct {
  const float *sdata;  /* L x 5: x / sigmaS, y / sigmaS, L / sigmaR, u / sigmaR, v / sigmaR */
  const int *buckets, *slist;
  int nb1, nb2;
  const int *neigh;
  float smin;
  double hiLTr;
  float thr, event_thr;
  int64_t self;           /* the pixel whose trajectory this is */
  unsigned char *event;   /* set when a point other than `self` comes within event_thr (may be NULL) */
  unsigned char *mode_table;
  int *point_list;
  int *point_count;
} Win;

static void window_shift(const Win *w, const double *yk, double *Mh) {
  /* (the event flag below: what the product's kernel reports next to a pixel's own mode, see oracle_ms_filter) */
  double wsum = 0;
  for (int j = 0; j < 5; ++j) Mh[j] = 0;
  const int c1 = (int)yk[0] + 1, c2 = (int)yk[1] + 1, c3 = (int)(yk[2] - w->smin) + 1;
  const int cb = c1 + w->nb1 * (c2 + w->nb2 * c3);
  for (int j = 0; j < 27; ++j) {
    int d = w->buckets[cb + w->neigh[j]];
    while (d >= 0) {
      const float *s = w->sdata + 5 * (int64_t)d;
      double el = s[0] - yk[0];
      double diff = el * el;
      el = s[1] - yk[1];
      diff += el * el;
      if (diff < 1.0) {
        el = s[2] - yk[2];
        if (yk[2] > w->hiLTr) diff = 4 * el * el;
        else diff = el * el;
        el = s[3] - yk[3];
        diff += el * el;
        el = s[4] - yk[4];
        diff += el * el;
        if (diff < 1.0) {
          const double weight = 1 - 0.0f;  /* weightMap is all zero without a fifth gateway argument (ms.cpp:478) */
          for (int k = 0; k < 5; ++k) Mh[k] += weight * s[k];
          wsum += weight;
          if (w->event && d != w->self && diff < w->event_thr) *w->event = 1;
          if (diff < w->thr && w->mode_table[d] == 0) {
            w->point_list[(*w->point_count)++] = d;
            w->mode_table[d] = 2;
          }
        }
      }
      d = w->slist[d];
    }
  }
  if (wsum > 0) for (int j = 0; j < 5; ++j) Mh[j] = Mh[j] / wsum - yk[j];
  else for (int j = 0; j < 5; ++j) Mh[j] = 0;
}

/* luv: H*W x 3 floats (row-major: pixel i = y * W + x), out: the filtered image, same layout (msRawData).
 * events (may be NULL): per pixel, 1 when on the pixel's trajectory AS WALKED HERE another pixel's colour came within
 * event_threshold of the trajectory's -- in a window pass or at the rounded position (:4085-4100).  With
 * speed_threshold <= 0 (every pixel walks its whole trajectory) this is what the product's kernel returns next to the
 * pixels' own modes; its host stage then walks the flagged pixels again, in scan order, with the shortcuts.
 * Returns 0, or 1 when memory runs out. */
int oracle_ms_filter(const float *luv, int H, int W, int sigmaS_i, float sigmaR, float speed_threshold, float event_threshold,
                     unsigned char *events, float *out) {
  const int64_t L = (int64_t)H * W;
  const float sigmaS = (float)sigmaS_i;   /* Filter() passes (float)(sigmaS), :412 */
  float *sdata = (float *)malloc(sizeof(float) * 5 * L);
  int *slist = (int *)malloc(sizeof(int) * L);
  unsigned char *mode_table = (unsigned char *)calloc(L, 1);
  int *point_list = (int *)malloc(sizeof(int) * L);
  if (!sdata || !slist || !mode_table || !point_list) return 1;
  for (int64_t i = 0; i < L; ++i) {   /* :3842-3852 */
    sdata[5 * i + 0] = (i % W) / sigmaS;
    sdata[5 * i + 1] = (i / W) / sigmaS;
    sdata[5 * i + 2] = luv[3 * i + 0] / sigmaR;
    sdata[5 * i + 3] = luv[3 * i + 1] / sigmaR;
    sdata[5 * i + 4] = luv[3 * i + 2] / sigmaR;
  }
  float smaxs[3], smin;   /* :3881-3897 */
  smaxs[0] = W / sigmaS;
  smaxs[1] = H / sigmaS;
  smin = smaxs[2] = sdata[2];
  for (int64_t i = 0; i < L; ++i) {
    const float c = sdata[5 * i + 2];
    if (c < smin) smin = c;
    else if (c > smaxs[2]) smaxs[2] = c;
  }
  const int nb1 = (int)(smaxs[0] + 3), nb2 = (int)(smaxs[1] + 3), nb3 = (int)(smaxs[2] - smin + 3);
  const int64_t nbuck = (int64_t)nb1 * nb2 * nb3;
  int *buckets = (int *)malloc(sizeof(int) * nbuck);
  if (!buckets) return 1;
  for (int64_t i = 0; i < nbuck; ++i) buckets[i] = -1;
  for (int64_t i = 0; i < L; ++i) {   /* :3909-3921: a bucket's list starts at the point inserted last */
    const int c1 = (int)sdata[5 * i] + 1, c2 = (int)sdata[5 * i + 1] + 1, c3 = (int)(sdata[5 * i + 2] - smin) + 1;
    const int cb = c1 + nb1 * (c2 + nb2 * c3);
    slist[i] = buckets[cb];
    buckets[cb] = (int)i;
  }
  int neigh[27], n = 0;   /* :3923-3933 */
  for (int a = -1; a <= 1; ++a)
    for (int b = -1; b <= 1; ++b)
      for (int c = -1; c <= 1; ++c) neigh[n++] = a + nb1 * (b + nb2 * c);
  int point_count = 0;
  Win w = {sdata, buckets, slist, nb1, nb2, neigh, smin, 80.0 / sigmaR, speed_threshold, event_threshold, 0, NULL, mode_table, point_list, &point_count};
  if (events) memset(events, 0, (size_t)L);
  for (int64_t i = 0; i < L; ++i) {   /* :3948-4287 */
    if (mode_table[i] == 1) continue;
    point_count = 0;
    w.self = i; w.event = events ? events + i : NULL;
    double yk[5], Mh[5];
    for (int j = 0; j < 5; ++j) yk[j] = sdata[5 * i + j];
    window_shift(&w, yk, Mh);
    double mv = (Mh[0] * Mh[0] + Mh[1] * Mh[1]) * sigmaS * sigmaS;
    mv += (Mh[2] * Mh[2] + Mh[3] * Mh[3] + Mh[4] * Mh[4]) * sigmaR * sigmaR;
    int iter = 1;
    while (mv >= 0.01 && iter < 100) {   /* EPSILON, LIMIT: ms.h:106,111 */
      for (int j = 0; j < 5; ++j) yk[j] += Mh[j];
      const int cx = (int)(sigmaS * yk[0] + 0.5), cy = (int)(sigmaS * yk[1] + 0.5);
      const int64_t ci = (int64_t)cy * W + cx;
      if (events && ci != i) {
        double diff = 0;
        for (int k = 2; k < 5; ++k) { const double el = sdata[5 * ci + k] - yk[k]; diff += el * el; }
        if (diff < event_threshold) events[i] = 1;
      }
      if (mode_table[ci] != 2 && ci != i) {   /* :4085-4127 */
        double diff = 0;
        for (int k = 2; k < 5; ++k) { const double el = sdata[5 * ci + k] - yk[k]; diff += el * el; }
        if (diff < speed_threshold) {
          if (mode_table[ci] == 0) { point_list[point_count++] = (int)ci; mode_table[ci] = 2; }
          else {
            for (int j = 0; j < 3; ++j) yk[j + 2] = out[3 * ci + j] / sigmaR;
            mode_table[i] = 1;
            mv = -1;
            break;
          }
        }
      }
      window_shift(&w, yk, Mh);
      mv = (Mh[0] * Mh[0] + Mh[1] * Mh[1]) * sigmaS * sigmaS;
      mv += (Mh[2] * Mh[2] + Mh[3] * Mh[3] + Mh[4] * Mh[4]) * sigmaR * sigmaR;
      ++iter;
    }
    if (mv >= 0) { for (int j = 0; j < 5; ++j) yk[j] += Mh[j]; mode_table[i] = 1; }
    for (int k = 0; k < 3; ++k) yk[k + 2] *= sigmaR;
    for (int j = 0; j < point_count; ++j) {
      const int64_t c = point_list[j];
      mode_table[c] = 1;
      for (int k = 0; k < 3; ++k) out[3 * c + k] = (float)yk[k + 2];
    }
    for (int k = 0; k < 3; ++k) out[3 * i + k] = (float)yk[k + 2];
  }
  free(buckets); free(point_list); free(mode_table); free(slist); free(sdata);
  return 0;
}
