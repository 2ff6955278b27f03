// Test infrastructure, NOT MATLAB's header: a small in-memory host for the subset of the documented
// MATLAB C Matrix / MEX API that mex/*.cpp use (column-major arrays, class ids, scalar structs,
// char row vectors, mexErrMsgTxt as a C++ exception).  It exists so that the gateway sources are
// compiled and their mexFunction is actually called in tests/ (this image has no MATLAB); written
// from the public API documentation, implementation in mexhost.cpp.
#pragma once
#include <cstddef>
#include <cstdint>

typedef size_t mwSize;
typedef size_t mwIndex;
typedef enum { mxUNKNOWN_CLASS = 0, mxSTRUCT_CLASS = 2, mxCHAR_CLASS = 4, mxDOUBLE_CLASS = 6, mxUINT8_CLASS = 9, mxINT32_CLASS = 12,
               mxUINT32_CLASS = 13, mxUINT64_CLASS = 15 } mxClassID;
typedef enum { mxREAL = 0, mxCOMPLEX = 1 } mxComplexity;
struct mxArray_tag;
typedef struct mxArray_tag mxArray;

extern "C" {
void mexFunction(int nlhs, mxArray *plhs[], int nrhs, const mxArray *prhs[]);
void mexErrMsgTxt(const char *msg);  // does not return

mxArray *mxCreateDoubleMatrix(mwSize m, mwSize n, mxComplexity flag);
mxArray *mxCreateDoubleScalar(double v);
mxArray *mxCreateNumericMatrix(mwSize m, mwSize n, mxClassID cls, mxComplexity flag);
void mxDestroyArray(mxArray *a);

bool mxIsDouble(const mxArray *a);
bool mxIsInt32(const mxArray *a);
bool mxIsUint8(const mxArray *a);
bool mxIsUint32(const mxArray *a);
bool mxIsUint64(const mxArray *a);
bool mxIsChar(const mxArray *a);
bool mxIsStruct(const mxArray *a);

mwSize mxGetM(const mxArray *a);
mwSize mxGetN(const mxArray *a);  // product of the dimensions after the first
mwSize mxGetNumberOfElements(const mxArray *a);
mwSize mxGetNumberOfDimensions(const mxArray *a);
const mwSize *mxGetDimensions(const mxArray *a);
double *mxGetPr(const mxArray *a);
void *mxGetData(const mxArray *a);
double mxGetScalar(const mxArray *a);
int mxGetString(const mxArray *a, char *buf, mwSize buflen);  // 0 on success
mxArray *mxGetField(const mxArray *a, mwIndex index, const char *name);
}
