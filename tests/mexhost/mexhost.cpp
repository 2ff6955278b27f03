// Test infrastructure: the in-memory MATLAB-API host declared in mex.h, plus a small C interface
// (mh_*) through which tests/test_mex_gateways*.py build arguments, call the gateway's mexFunction
// and read the results back.  One shared object per gateway: gateway.cpp + this file.
#include "mex.h"

#include <cstring>
#include <map>
#include <stdexcept>
#include <string>
#include <vector>

struct mxArray_tag {
  mxClassID cls = mxUNKNOWN_CLASS;
  std::vector<mwSize> dims;
  std::vector<unsigned char> data;             // numeric / char (one byte per char: ASCII keys only)
  std::map<std::string, mxArray *> fields;     // 1 x 1 struct
};

namespace {
size_t elem_size(mxClassID c) {
  switch (c) {
    case mxDOUBLE_CLASS: case mxUINT64_CLASS: return 8;
    case mxINT32_CLASS: case mxUINT32_CLASS: return 4;
    case mxCHAR_CLASS: case mxUINT8_CLASS: return 1;
    default: return 0;
  }
}
mwSize numel(const mxArray *a) {
  mwSize n = 1;
  for (mwSize d : a->dims) n *= d;
  return n;
}
mxArray *make(mxClassID cls, const std::vector<mwSize> &dims) {
  mxArray *a = new mxArray_tag;
  a->cls = cls; a->dims = dims;
  a->data.assign(numel(a) * elem_size(cls), 0);
  return a;
}
struct MexError : std::runtime_error { using std::runtime_error::runtime_error; };
}  // namespace

extern "C" {
void mexErrMsgTxt(const char *msg) { throw MexError(msg ? msg : ""); }

mxArray *mxCreateDoubleMatrix(mwSize m, mwSize n, mxComplexity) { return make(mxDOUBLE_CLASS, {m, n}); }
mxArray *mxCreateDoubleScalar(double v) {
  mxArray *a = make(mxDOUBLE_CLASS, {1, 1});
  std::memcpy(a->data.data(), &v, 8);
  return a;
}
mxArray *mxCreateNumericMatrix(mwSize m, mwSize n, mxClassID cls, mxComplexity) { return make(cls, {m, n}); }
void mxDestroyArray(mxArray *a) {
  if (!a) return;
  for (auto &f : a->fields) mxDestroyArray(f.second);
  delete a;
}
bool mxIsDouble(const mxArray *a) { return a && a->cls == mxDOUBLE_CLASS; }
bool mxIsInt32(const mxArray *a) { return a && a->cls == mxINT32_CLASS; }
bool mxIsUint32(const mxArray *a) { return a && a->cls == mxUINT32_CLASS; }
bool mxIsUint8(const mxArray *a) { return a && a->cls == mxUINT8_CLASS; }
bool mxIsUint64(const mxArray *a) { return a && a->cls == mxUINT64_CLASS; }
bool mxIsChar(const mxArray *a) { return a && a->cls == mxCHAR_CLASS; }
bool mxIsStruct(const mxArray *a) { return a && a->cls == mxSTRUCT_CLASS; }
mwSize mxGetM(const mxArray *a) { return a->dims.empty() ? 0 : a->dims[0]; }
mwSize mxGetN(const mxArray *a) {
  mwSize n = 1;
  for (size_t i = 1; i < a->dims.size(); ++i) n *= a->dims[i];
  return a->dims.size() < 2 ? 0 : n;
}
mwSize mxGetNumberOfElements(const mxArray *a) { return numel(a); }
mwSize mxGetNumberOfDimensions(const mxArray *a) { return a->dims.size(); }
const mwSize *mxGetDimensions(const mxArray *a) { return a->dims.data(); }
double *mxGetPr(const mxArray *a) { return a->cls == mxDOUBLE_CLASS ? (double *)a->data.data() : nullptr; }
void *mxGetData(const mxArray *a) { return (void *)a->data.data(); }
double mxGetScalar(const mxArray *a) {
  if (numel(a) == 0) return 0;
  switch (a->cls) {
    case mxDOUBLE_CLASS: { double v; std::memcpy(&v, a->data.data(), 8); return v; }
    case mxINT32_CLASS: { int32_t v; std::memcpy(&v, a->data.data(), 4); return v; }
    case mxUINT32_CLASS: { uint32_t v; std::memcpy(&v, a->data.data(), 4); return v; }
    case mxUINT64_CLASS: { uint64_t v; std::memcpy(&v, a->data.data(), 8); return (double)v; }
    case mxCHAR_CLASS: return a->data[0];
    default: return 0;
  }
}
int mxGetString(const mxArray *a, char *buf, mwSize buflen) {
  if (!a || a->cls != mxCHAR_CLASS || buflen == 0) return 1;
  const mwSize n = numel(a);
  const mwSize k = n < buflen - 1 ? n : buflen - 1;
  std::memcpy(buf, a->data.data(), k);
  buf[k] = 0;
  return n > buflen - 1 ? 1 : 0;
}
mxArray *mxGetField(const mxArray *a, mwIndex index, const char *name) {
  if (!a || a->cls != mxSTRUCT_CLASS || index != 0) return nullptr;
  auto it = a->fields.find(name);
  return it == a->fields.end() ? nullptr : it->second;
}

// ---- what the Python tests call -------------------------------------------------------------
// class ids: 6 double, 12 int32, 13 uint32, 15 uint64; dims column-major as MATLAB has them
mxArray *mh_numeric(int cls, int ndim, const uint64_t *dims, const void *data) {
  std::vector<mwSize> d(dims, dims + ndim);
  mxArray *a = make((mxClassID)cls, d);
  if (data && !a->data.empty()) std::memcpy(a->data.data(), data, a->data.size());
  return a;
}
mxArray *mh_string(const char *s) {
  mxArray *a = make(mxCHAR_CLASS, {1, std::strlen(s)});
  std::memcpy(a->data.data(), s, std::strlen(s));
  return a;
}
mxArray *mh_struct() {
  mxArray *a = new mxArray_tag;
  a->cls = mxSTRUCT_CLASS; a->dims = {1, 1};
  return a;
}
void mh_set_field(mxArray *s, const char *name, mxArray *value) { s->fields[name] = value; }
void mh_free(mxArray *a) { mxDestroyArray(a); }
uint64_t mh_numel(const mxArray *a) { return numel(a); }
int mh_class(const mxArray *a) { return (int)a->cls; }
const void *mh_data(const mxArray *a) { return a->data.data(); }
// Calls the gateway; 0 on success, 1 if it raised through mexErrMsgTxt (message in err).
int mh_call(int nlhs, mxArray **plhs, int nrhs, const mxArray **prhs, char *err, uint64_t errcap) {
  try {
    mexFunction(nlhs, plhs, nrhs, prhs);
    return 0;
  } catch (const MexError &e) {
    if (err && errcap) { std::strncpy(err, e.what(), errcap - 1); err[errcap - 1] = 0; }
    return 1;
  }
}
}
