// Optional extra mex gateway: the device-resident fusion context of libstereo_hip.so
// (stereo_fusion_*, include/stereo_hip.h).  Not a replacement of a reference file: it is the one
// additional gateway a maintainer puts behind dispmap_super.binary_fusion / simultaneous_fusion /
// update_energy (dispmap_super.m:61-84, :153-198, :263-274) so that terms never cross PCIe.
//
//   h  = fusion_mex('create', H, W, int32(kernel), tol, uint32(connectivity-1) 2xE, weights 1xE, d_min, d_step)
//        fusion_mex('unary_ncc', h, ncc HxWxD, disparities, unary_weight)            % dispmap_ncc
//        fusion_mex('unary_globalstereo', h, im0, im1, P2 4x3, col_thresh)            % dispmap_globalstereo
//   e  = fusion_mex('set_assignment', h, assignment 4xN)                             % set.assignment + update_energy
//   [a, e] = fusion_mex('get_assignment', h, N)
//   [e, rd_e, lb, unl] = fusion_mex('binary', h, proposal 4xN, improve)
//   [e, trws_e, lb, it] = fusion_mex('simultaneous', h, proposals 4xNxK, maxiter, max_relgap)
//        fusion_mex('destroy', h)
// h is a uint64 scalar.  Build inside MATLAB:
//   mex -I<repo>/include mex/fusion_mex.cpp -L<repo>/stereo_amd -lstereo_hip
#include <cstdint>
#include <cstring>

#include "mex.h"
#include "stereo_hip.h"

static void need(bool ok, const char *what) {
  if (!ok) mexErrMsgTxt(what);
}
static stereo_fusion *handle(const mxArray *a) {
  need(mxIsUint64(a) && mxGetNumberOfElements(a) == 1, "handle must be a uint64 scalar");
  return (stereo_fusion *)(uintptr_t)(*(const uint64_t *)mxGetData(a));
}
static void check(int rc, const char *err) {
  if (rc) mexErrMsgTxt(err);
}

void mexFunction(int nlhs, mxArray *plhs[], int nrhs, const mxArray *prhs[]) {
  need(nrhs >= 1 && mxIsChar(prhs[0]), "first argument: command string");
  char cmd[32];
  need(!mxGetString(prhs[0], cmd, sizeof(cmd)), "command string too long");
  char err[512] = "";
  if (!std::strcmp(cmd, "create")) {
    need(nrhs == 9 && mxIsInt32(prhs[3]) && mxIsUint32(prhs[5]) && mxGetM(prhs[5]) == 2, "create: bad arguments");
    const int64_t E = (int64_t)mxGetN(prhs[5]);
    need((int64_t)mxGetNumberOfElements(prhs[6]) == E, "create: one weight per edge");
    stereo_fusion *F = nullptr;
    check(stereo_fusion_create((int)mxGetScalar(prhs[1]), (int)mxGetScalar(prhs[2]), *(const int32_t *)mxGetData(prhs[3]),
                               mxGetScalar(prhs[4]), E, (const uint32_t *)mxGetData(prhs[5]), mxGetPr(prhs[6]),
                               mxGetScalar(prhs[7]), mxGetScalar(prhs[8]), &F, err, sizeof(err)), err);
    plhs[0] = mxCreateNumericMatrix(1, 1, mxUINT64_CLASS, mxREAL);
    *(uint64_t *)mxGetData(plhs[0]) = (uint64_t)(uintptr_t)F;
    return;
  }
  need(nrhs >= 2, "handle missing");
  stereo_fusion *F = handle(prhs[1]);
  if (!std::strcmp(cmd, "destroy")) {
    stereo_fusion_destroy(F);
  } else if (!std::strcmp(cmd, "unary_ncc")) {
    need(nrhs == 5 && mxIsDouble(prhs[2]) && mxIsDouble(prhs[3]), "unary_ncc: bad arguments");
    check(stereo_fusion_unary_ncc(F, mxGetPr(prhs[2]), (int)mxGetNumberOfElements(prhs[3]), mxGetPr(prhs[3]),
                                  mxGetScalar(prhs[4]), err, sizeof(err)), err);
  } else if (!std::strcmp(cmd, "unary_globalstereo")) {
    need(nrhs == 6 && mxIsDouble(prhs[2]) && mxIsDouble(prhs[3]) && mxGetNumberOfElements(prhs[4]) == 12,
         "unary_globalstereo: bad arguments");
    const mwSize nd = mxGetNumberOfDimensions(prhs[2]);
    const int C = nd > 2 ? (int)mxGetDimensions(prhs[2])[2] : 1;
    check(stereo_fusion_unary_globalstereo(F, mxGetPr(prhs[2]), mxGetPr(prhs[3]), C, mxGetPr(prhs[4]),
                                           mxGetScalar(prhs[5]), err, sizeof(err)), err);
  } else if (!std::strcmp(cmd, "set_assignment")) {
    need(nrhs == 3 && mxIsDouble(prhs[2]) && mxGetM(prhs[2]) == 4, "set_assignment: assignment is 4 x N");
    double e = 0;
    check(stereo_fusion_set_assignment(F, mxGetPr(prhs[2]), &e, err, sizeof(err)), err);
    plhs[0] = mxCreateDoubleScalar(e);
  } else if (!std::strcmp(cmd, "get_assignment")) {
    need(nrhs == 3 && nlhs >= 1, "get_assignment: pass N as third argument");
    const mwSize N = (mwSize)mxGetScalar(prhs[2]);
    plhs[0] = mxCreateDoubleMatrix(4, N, mxREAL);
    double e = 0;
    check(stereo_fusion_get_assignment(F, mxGetPr(plhs[0]), &e, err, sizeof(err)), err);
    if (nlhs > 1) plhs[1] = mxCreateDoubleScalar(e);
  } else if (!std::strcmp(cmd, "binary")) {
    need(nrhs == 4 && mxIsDouble(prhs[2]) && mxGetM(prhs[2]) == 4, "binary: proposal is 4 x N");
    double e = 0, re = 0, lb = 0, unl = 0;
    check(stereo_fusion_binary(F, mxGetPr(prhs[2]), mxGetScalar(prhs[3]) != 0, &e, &re, &lb, &unl, err, sizeof(err)), err);
    plhs[0] = mxCreateDoubleScalar(e);
    if (nlhs > 1) plhs[1] = mxCreateDoubleScalar(re);
    if (nlhs > 2) plhs[2] = mxCreateDoubleScalar(lb);
    if (nlhs > 3) plhs[3] = mxCreateDoubleScalar(unl);
  } else if (!std::strcmp(cmd, "simultaneous")) {
    need(nrhs == 5 && mxIsDouble(prhs[2]) && mxGetM(prhs[2]) == 4, "simultaneous: proposals are 4 x N x K");
    const mwSize nd = mxGetNumberOfDimensions(prhs[2]);
    const int K = nd > 2 ? (int)mxGetDimensions(prhs[2])[2] : 1;
    double e = 0, te = 0, lb = 0, it = 0;
    check(stereo_fusion_simultaneous(F, mxGetPr(prhs[2]), K, mxGetScalar(prhs[3]), mxGetScalar(prhs[4]), &e, &te, &lb,
                                     &it, err, sizeof(err)), err);
    plhs[0] = mxCreateDoubleScalar(e);
    if (nlhs > 1) plhs[1] = mxCreateDoubleScalar(te);
    if (nlhs > 2) plhs[2] = mxCreateDoubleScalar(lb);
    if (nlhs > 3) plhs[3] = mxCreateDoubleScalar(it);
  } else {
    mexErrMsgTxt("fusion_mex: unknown command");
  }
}
