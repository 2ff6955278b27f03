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

void rgb_to_luv(const uint8_t *A, int H, int W, float *luv);
void ms_lattice(const float *luv, int H, int W, int sigmaS, float sigmaR, MsLattice &lat);
void ms_regions(const float *filtered, int H, int W, float sigmaR, int min_region, int32_t *labels);
std::vector<float> gb_mask(float sigma);
void gb_regions(const float *weights, int H, int W, float c, int min_size, int compress, uint32_t *out);

}  // namespace seg
}  // namespace stereo
