// mex gateway for libstereo_hip.so -- drop-in replacement of the reference's
// cpp/rd_mex.cpp (same 7|8-in / 4-out contract, rd.m:21 calls it unchanged):
//   [labelling, energy, lower_bound, num_unlabelled] =
//       rd_mex(U0 Nx1, U1 Nx1, E00 1xE, E01, E10, E11, uint32 connectivity-1 2xE, options)
// Build inside MATLAB:  mex -I<repo>/include mex/rd_mex.cpp -L<repo>/stereo_amd -lstereo_hip
#include <cstring>

#include "mex.h"
#include "stereo_hip.h"

static void need(bool ok, const char *what) {
  if (!ok) mexErrMsgTxt(what);
}

static bool improve_option(int nopt, const mxArray *opt[]) {   // rd_mex.cpp:33-34, default false
  if (nopt == 1 && mxIsStruct(opt[0])) {
    const mxArray *f = mxGetField(opt[0], 0, "improve");
    return f && mxGetScalar(f) != 0;
  }
  for (int i = 0; i + 1 < nopt; i += 2) {
    char key[64];
    if (mxIsChar(opt[i]) && !mxGetString(opt[i], key, sizeof(key)) && !std::strcmp(key, "improve"))
      return mxGetScalar(opt[i + 1]) != 0;
  }
  return false;
}

void mexFunction(int nlhs, mxArray *plhs[], int nrhs, const mxArray *prhs[]) {
  need(nrhs == 7 || nrhs == 8, "Error: nrhs == 7 || nrhs == 8");   // rd_mex.cpp:20
  need(nlhs == 4, "Error: nlhs == 4");                             // rd_mex.cpp:21
  for (int i = 0; i < 6; ++i) need(mxIsDouble(prhs[i]), "wrong argument class");
  need(mxIsUint32(prhs[6]), "connectivity must be uint32");
  const mwSize N = mxGetM(prhs[0]), E = mxGetN(prhs[2]);
  need(mxGetM(prhs[1]) == N && mxGetN(prhs[0]) == 1 && mxGetN(prhs[1]) == 1, "U0, U1 are N x 1");   // :36-39
  need(mxGetN(prhs[3]) == E && mxGetN(prhs[4]) == E && mxGetN(prhs[5]) == E && mxGetN(prhs[6]) == E,
       "E00, E01, E10, E11, connectivity agree in length");                                       // :41-44
  need(mxGetM(prhs[2]) == 1 && mxGetM(prhs[3]) == 1 && mxGetM(prhs[4]) == 1 && mxGetM(prhs[6]) == 2,
       "E** are 1 x E, connectivity is 2 x E");                                                   // :46-49
  plhs[0] = mxCreateDoubleMatrix(N, 1, mxREAL);
  double energy = 0, lb = 0, unl = 0;
  char err[512] = "";
  const int rc = stereo_rd(mxGetPr(prhs[0]), mxGetPr(prhs[1]), mxGetPr(prhs[2]), mxGetPr(prhs[3]), mxGetPr(prhs[4]),
                           mxGetPr(prhs[5]), (const uint32_t *)mxGetData(prhs[6]), (int64_t)N, (int64_t)E,
                           improve_option(nrhs - 7, prhs + 7) ? 1 : 0, mxGetPr(plhs[0]), &energy, &lb, &unl, err,
                           sizeof(err));
  if (rc) mexErrMsgTxt(err);
  plhs[1] = mxCreateDoubleScalar(energy);
  plhs[2] = mxCreateDoubleScalar(lb);
  plhs[3] = mxCreateDoubleScalar(unl);
}
