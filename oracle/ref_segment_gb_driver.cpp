// TEST INFRASTRUCTURE ONLY -- not part of the product path.
//
// Driver over the reference's own graph-based segmenter (Felzenszwalb & Huttenlocher, vendored
// header-only under /root/reference/imrender/vgg/seg_gb, standard library only, compiled where it
// lies, never copied) -> oracle/_ref/libref_segment_gb.so.  The mex gateway
// (imrender/vgg/vgg_segment_gb.cxx) includes <mex.h>, which the image lacks and which is not faked;
// this file restates the gateway's marshalling:
//
//   vgg_segment_gb.cxx:33-35   sigma, k (float), min_size (int)
//   vgg_segment_gb.cxx:38-47   uint8 H x W x 3 column-major -> image<rgb>(W, H), row-major interleaved
//   vgg_segment_gb.cxx:50-51   output = mxCreateNumericMatrix: ZERO-initialised uint32 H x W (the
//                              library leaves the last row and column unwritten, segment-image.h:239-243)
//   vgg_segment_gb.cxx:55      segment_image(input, sigma, k, min_size, &num_sets, C)
//   vgg_segment_gb.cxx:58-82   compress: ids replaced by 1 + rank of first appearance in a
//                              column-major scan (a linear search there, a hash map here: same ids)
#include <stdint.h>

#include <cstring>
#include <unordered_map>

#include "image.h"
#include "misc.h"
#include "segment-image.h"

extern "C" {

// A: H x W x 3 uint8 column-major.  out: H x W uint32 column-major.
int ref_segment_gb(const uint8_t* A, int H, int W, float sigma, float k, int min_size, int compress,
                   uint32_t* out) {
    image<rgb>* input = new image<rgb>(W, H);
    uint8_t* B = (uint8_t*)imPtr(input, 0, 0);
    const size_t plane = (size_t)H * W;
    for (int h = 0; h < H; h++)
        for (size_t w = 0; w < plane; w += H) {
            *B++ = A[h + w];
            *B++ = A[h + w + plane];
            *B++ = A[h + w + 2 * plane];
        }
    memset(out, 0, plane * sizeof(uint32_t));
    int num_sets = 0;
    segment_image(input, sigma, k, min_size, &num_sets, out);
    delete input;
    if (compress) {
        std::unordered_map<uint32_t, uint32_t> rank;
        for (size_t i = 0; i < plane; i++) {          // column-major scan = memory order
            auto it = rank.find(out[i]);
            if (it == rank.end()) it = rank.emplace(out[i], (uint32_t)rank.size() + 1).first;
            out[i] = it->second;
        }
    }
    return 0;
}

}  // extern "C"
