// Gateway with the contract of imrender/vgg/vgg_segment_ms.cxx:18-87:
//   S = vgg_segment_ms(A, h_s, h_r, min_sz)
// A: H x W x 3 uint8; S: H x W uint32, labels from 1.  Marshals pointers to stereo_segment_ms (include/stereo_hip.h:
// mean-shift filter on the device, region graph on the host); same argument checks and messages as the reference.
// The optional fifth argument of the reference (a synergistic weight map, :29-32,55-68) has no caller in the reference
// tree (dispmap_globalstereo.m:129,391 pass four) and is refused.
// Build inside MATLAB:  mex -I<repo>/include mex/vgg_segment_ms.cpp -L<repo>/stereo_amd -lstereo_hip
#include <cstdint>

#include "mex.h"
#include "stereo_hip.h"

void mexFunction(int nlhs, mxArray *plhs[], int nrhs, const mxArray *prhs[]) {
  if (nrhs < 4 || nrhs > 5) mexErrMsgTxt("Unexpected number of input arguments.");      // :21-22
  if (nlhs < 1 || nlhs > 2) mexErrMsgTxt("Unexpected number of output arguments.");     // :23-24
  const mwSize *dims = mxGetDimensions(prhs[0]);
  if (!mxIsUint8(prhs[0]) || mxGetNumberOfDimensions(prhs[0]) != 3 || dims[2] != 3)
    mexErrMsgTxt("A must be an HxWx3 uint8 array.");                                     // :26-27
  if (nrhs > 4) mexErrMsgTxt("vgg_segment_ms: the edge weight map (fifth argument) is not supported by this library.");
  const int H = (int)dims[0], W = (int)dims[1];
  plhs[0] = mxCreateNumericMatrix(dims[0], dims[1], mxUINT32_CLASS, mxREAL);
  char err[512] = "";
  if (stereo_segment_ms((const uint8_t *)mxGetData(prhs[0]), H, W, mxGetScalar(prhs[1]), mxGetScalar(prhs[2]), mxGetScalar(prhs[3]),
                        (uint32_t *)mxGetData(plhs[0]), err, sizeof(err)))
    mexErrMsgTxt(err);
}
