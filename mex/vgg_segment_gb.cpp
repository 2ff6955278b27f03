// Gateway with the contract of imrender/vgg/vgg_segment_gb.cxx:21-87:
//   S = vgg_segment_gb(A, sigma, k, min_sz, compress)
// A: H x W x 3 uint8; S: H x W uint32 (union-find roots, or 1, 2, ... by first appearance with compress; the last row
// and column stay what the library leaves them, see include/stereo_hip.h).  Marshals pointers to stereo_segment_gb
// (smoothing and edge weights on the device, sort + union-find on the host).
// Build inside MATLAB:  mex -I<repo>/include mex/vgg_segment_gb.cpp -L<repo>/stereo_amd -lstereo_hip
#include <cstdint>

#include "mex.h"
#include "stereo_hip.h"

void mexFunction(int nlhs, mxArray *plhs[], int nrhs, const mxArray *prhs[]) {
  if (nrhs < 4 || nrhs > 5) mexErrMsgTxt("Unexpected number of input arguments.");      // :24-25
  if (nlhs != 1) mexErrMsgTxt("Unexpected number of output arguments.");                // :26-27
  const mwSize *dims = mxGetDimensions(prhs[0]);
  if (!mxIsUint8(prhs[0]) || mxGetNumberOfDimensions(prhs[0]) != 3 || dims[2] != 3)
    mexErrMsgTxt("A must be an HxWx3 uint8 array.");                                     // :29-30
  const int H = (int)dims[0], W = (int)dims[1];
  const int compress = nrhs > 4 && mxGetScalar(prhs[4]) != 0;                            // :57
  plhs[0] = mxCreateNumericMatrix(dims[0], dims[1], mxUINT32_CLASS, mxREAL);
  char err[512] = "";
  if (stereo_segment_gb((const uint8_t *)mxGetData(prhs[0]), H, W, mxGetScalar(prhs[1]), mxGetScalar(prhs[2]), mxGetScalar(prhs[3]), compress,
                        (uint32_t *)mxGetData(plhs[0]), err, sizeof(err)))
    mexErrMsgTxt(err);
}
