// TEST INFRASTRUCTURE ONLY -- not part of the product path.
//
// Driver that links the reference's own mean-shift segmenter (EDISON, vendored under
// /root/reference/imrender/vgg/seg_ms, libc only, compiled where it lies, never copied) into
// oracle/_ref/libref_segment_ms.so.  The reference's mex gateway
// (imrender/vgg/vgg_segment_ms.cxx) includes <mex.h>, which the image lacks and which is not
// faked; this file restates the gateway's marshalling between MATLAB and the library:
//
//   vgg_segment_ms.cxx:33-36   sigmaS (int), sigmaR (float), minRegion (int) from the scalars
//   vgg_segment_ms.cxx:40-50   uint8 H x W x 3 column-major -> row-major interleaved RGB
//   vgg_segment_ms.cxx:51      DefineImage(buffer, COLOR, H, W)
//   vgg_segment_ms.cxx:71      Segment(sigmaS, sigmaR, minRegion, HIGH_SPEEDUP)
//   vgg_segment_ms.cxx:76-84   labels (row-major) + 1 -> uint32 H x W column-major
// The optional fifth argument (edge weight map, :55-68) has no caller in the reference
// (dispmap_globalstereo.m:391-392 passes four arguments) and is not offered.
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <vector>

#include "msImageProcessor.h"

extern "C" {

// A: H x W x 3 uint8, MATLAB column-major.  out: H x W uint32, column-major, labels from 1.
// Returns 0, or 1 with the library's message in err.
int ref_segment_ms(const uint8_t* A, int H, int W, int sigmaS, float sigmaR, int minRegion,
                   uint32_t* out, char* err, size_t errcap) {
    msImageProcessor im_proc;
    std::vector<uint8_t> buf((size_t)H * W * 3);
    uint8_t* B = buf.data();
    const size_t plane = (size_t)H * W;
    for (int h = 0; h < H; h++)
        for (size_t w = 0; w < plane; w += H) {
            *B++ = A[h + w];
            *B++ = A[h + w + plane];
            *B++ = A[h + w + 2 * plane];
        }
    im_proc.DefineImage(buf.data(), COLOR, H, W);
    if (im_proc.ErrorStatus == EL_ERROR) {
        if (err && errcap) snprintf(err, errcap, "%s", im_proc.ErrorMessage);
        return 1;
    }
    im_proc.Segment(sigmaS, sigmaR, minRegion, HIGH_SPEEDUP);
    if (im_proc.ErrorStatus == EL_ERROR) {
        if (err && errcap) snprintf(err, errcap, "%s", im_proc.ErrorMessage);
        return 1;
    }
    const int* labels = im_proc.GetLabels();
    for (int h = 0; h < H; h++)
        for (size_t w = 0; w < plane; w += H) out[h + w] = (uint32_t)(*labels++) + 1;
    return 0;
}

}  // extern "C"
