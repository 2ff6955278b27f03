// Host stages of the segmenters (segment_host.cpp); the device stages and the C ABI are in segment.hip.
#pragma once
#include <cstdint>
#include <vector>

namespace stereo {
namespace seg {

// The mean-shift filter's input: points scaled by the bandwidths, binned (msImageProcessor.cpp:3842-3933).
struct MsLattice {
  int H = 0, W = 0;
  float sigmaS = 0, sigmaR = 0, smin = 0;
  int nb1 = 0, nb2 = 0, nb3 = 0;
  int neigh[27] = {0};                 // offsets of the 27 neighbouring buckets, in the order the reference adds them
  std::vector<float> sdata;            // L x 5
  std::vector<int32_t> bucket_ptr;     // buckets + 1
  std::vector<int32_t> bucket_items;   // L: a bucket's points, last inserted first
};

// What msImageProcessor::speedThreshold holds when vgg_segment_ms.cxx reads it uninitialised (msImageProcessor.h:796; the
// constructor :64-108 and the gateway never set it): the object lies on the gateway's stack and the float's four bytes
// are the upper half of a stack address, 0x00007ffc on x86-64 Linux -- the denormal 32764 * 2^-149.  Seen in the
// reference's own build (oracle/_ref) and behind the fixtures tests/golden/*_segments.npz; every value in (0, 1e-30]
// gives the same maps.  In effect: "the colours are equal".
constexpr float kSpeedThreshold = 4.5912142885138307e-41f;

void rgb_to_luv(const uint8_t *A, int H, int W, float *luv);
void ms_lattice(const float *luv, int H, int W, int sigmaS, float sigmaR, MsLattice &lat);
// own: every pixel's own mode (L x 3), events: whether another pixel's colour came within thr of its trajectory; out:
// the reference's filtered image (msRawData).  Returns the number of pixels walked again on the host.
int64_t ms_filter_finish(const MsLattice &lat, const float *own, const uint8_t *events, float thr, float *out);
void ms_regions(const float *filtered, int H, int W, float sigmaR, int min_region, int32_t *labels);
std::vector<float> gb_mask(float sigma);
void gb_regions(const float *weights, int H, int W, float c, int min_size, int compress, uint32_t *out);

}  // namespace seg
}  // namespace stereo
