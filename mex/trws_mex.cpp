// mex gateway for libstereo_hip.so -- drop-in replacement of the reference's
// cpp/trws_mex.cpp (same 8-in / 4-out contract, trws.m:33 calls it unchanged):
//   [labelling, energy, lower_bound, iterations] =
//       trws_mex(int32 kernel, unary KxN, uint32 connectivity-1 2xE, q KxE, qprim KxE,
//                alphas Ex1, tol 1x1, options)
// Build inside MATLAB:  mex -I<repo>/include mex/trws_mex.cpp -L<repo>/stereo_amd -lstereo_hip
// (this image has no MATLAB: tests/test_mex_gateways.py compiles this file against the small
// MATLAB-API host of tests/mexhost and calls mexFunction.)
#include <cstring>
#include <string>

#include "mex.h"
#include "stereo_hip.h"

static void need(bool ok, const char *what) {
  if (!ok) mexErrMsgTxt(what);
}

// options: struct with fields, or trailing key/value pairs (cpp/utils/mexutils.h:56-82)
static double option(int nopt, const mxArray *opt[], const char *name, double def) {
  if (nopt == 1 && mxIsStruct(opt[0])) {
    const mxArray *f = mxGetField(opt[0], 0, name);
    return f ? mxGetScalar(f) : def;
  }
  for (int i = 0; i + 1 < nopt; i += 2) {
    char key[64];
    if (mxIsChar(opt[i]) && !mxGetString(opt[i], key, sizeof(key)) && !std::strcmp(key, name))
      return mxGetScalar(opt[i + 1]);
  }
  return def;
}

void mexFunction(int nlhs, mxArray *plhs[], int nrhs, const mxArray *prhs[]) {
  need(nrhs == 8, "Error: nrhs == 8");   // trws_mex.cpp:152
  need(nlhs == 4, "Error: nlhs == 4");   // trws_mex.cpp:153
  need(mxIsInt32(prhs[0]), "kernel must be int32");
  need(mxIsDouble(prhs[1]) && mxIsUint32(prhs[2]) && mxIsDouble(prhs[3]) && mxIsDouble(prhs[4]) &&
           mxIsDouble(prhs[5]) && mxIsDouble(prhs[6]), "wrong argument class");   // cppmatrix.h:126
  const int kernel = *(const int *)mxGetData(prhs[0]);
  const mwSize K = mxGetM(prhs[1]), N = mxGetN(prhs[1]), E = mxGetN(prhs[3]);
  need(mxGetM(prhs[2]) == 2, "connectivity.M == 2");                             // trws_mex.cpp:43-52
  need(mxGetN(prhs[4]) == E && mxGetN(prhs[2]) == E, "q.N == qprim.N == connectivity.N");
  need(mxGetM(prhs[3]) == K && mxGetM(prhs[4]) == K, "unary.M == q.M == qprim.M");
  need(mxGetM(prhs[5]) == E && mxGetN(prhs[5]) == 1, "alphas is E x 1");
  need(mxGetNumberOfElements(prhs[6]) == 1, "tol.numel() == 1");
  const double maxiter = option(nrhs - 7, prhs + 7, "maxiter", 1000);            // trws_mex.cpp:40
  const double max_relgap = option(nrhs - 7, prhs + 7, "max_relgap", 0);        // trws_mex.cpp:41
  plhs[0] = mxCreateDoubleMatrix(N, 1, mxREAL);
  double energy = 0, lb = 0, iters = 0;
  char err[512] = "";
  const int rc = stereo_trws(kernel, mxGetPr(prhs[1]), (const uint32_t *)mxGetData(prhs[2]), mxGetPr(prhs[3]),
                             mxGetPr(prhs[4]), mxGetPr(prhs[5]), mxGetScalar(prhs[6]), maxiter, max_relgap,
                             (int)K, (int64_t)N, (int64_t)E, mxGetPr(plhs[0]), &energy, &lb, &iters, err,
                             sizeof(err));
  if (rc) mexErrMsgTxt(err);                                                      // "Unsupported kernel", ...
  plhs[1] = mxCreateDoubleScalar(energy);
  plhs[2] = mxCreateDoubleScalar(lb);
  plhs[3] = mxCreateDoubleScalar(iters);
}
