// Shared helpers of libstereo_hip.so (error reporting, HIP checks).
#pragma once
#include <cstdlib>
#include <hip/hip_runtime.h>

#include <cstddef>
#include <cstdio>
#include <cstring>
#include <string>

namespace stereo {

std::string &last_error();  // thread local

inline int fail(const std::string &msg, char *err, size_t errcap, int code = 1) {
  last_error() = msg;
  if (err && errcap) {
    std::strncpy(err, msg.c_str(), errcap - 1);
    err[errcap - 1] = 0;
  }
  return code;
}

struct HipError {
  std::string msg;
};

#define STEREO_HIP_CHECK(expr)                                                          \
  do {                                                                                  \
    hipError_t e_ = (expr);                                                             \
    if (e_ != hipSuccess) {                                                             \
      char b_[512];                                                                     \
      std::snprintf(b_, sizeof(b_), "HIP error %d (%s) at %s:%d: %s", (int)e_,          \
                    hipGetErrorString(e_), __FILE__, __LINE__, #expr);                  \
      throw ::stereo::HipError{b_};                                                     \
    }                                                                                   \
  } while (0)

// A plan lives on the device that was current when it was created.  Every entry point that
// allocates, launches, copies or synchronises for a plan makes that device current for its own
// duration and restores the caller's afterwards (a process may drive one plan per GPU).
struct DeviceScope {
  int prev = -1;
  bool switched = false;
  explicit DeviceScope(int device) {
    if (device >= 0 && hipGetDevice(&prev) == hipSuccess && prev != device) switched = hipSetDevice(device) == hipSuccess;
  }
  DeviceScope(const DeviceScope &) = delete;
  DeviceScope &operator=(const DeviceScope &) = delete;
  ~DeviceScope() {
    if (switched) (void)hipSetDevice(prev);
  }
};

template <class T>
struct DevBuf {
  T *p = nullptr;
  size_t n = 0;
  DevBuf() = default;
  DevBuf(const DevBuf &) = delete;
  DevBuf &operator=(const DevBuf &) = delete;
  ~DevBuf() { release(); }
  void release() {
    if (p) (void)hipFree(p);
    p = nullptr;
    n = 0;
  }
  void alloc(size_t count) {
    release();
    if (count == 0) count = 1;
    STEREO_HIP_CHECK(hipMalloc((void **)&p, count * sizeof(T)));
    n = count;
    // development aid: STEREO_HIP_POISON=<byte> fills every fresh allocation, so that a read of
    // memory nobody wrote shows up as a changed result instead of depending on what was there
    static const char *poison = std::getenv("STEREO_HIP_POISON");
    if (poison) STEREO_HIP_CHECK(hipMemset(p, std::atoi(poison), count * sizeof(T)));
  }
  // Fine-grained device memory: what one GPU's kernel stores there is visible to a kernel running on
  // another GPU (system-scope accesses) without waiting for a kernel boundary -- the arrays a row
  // strip's neighbours write into while it runs.  (Ordinary hipMalloc memory is coarse-grained: the
  // runtime only promises cross-device visibility at synchronisation points.)
  void alloc_fine_grained(size_t count) {
    release();
    if (count == 0) count = 1;
    STEREO_HIP_CHECK(hipExtMallocWithFlags((void **)&p, count * sizeof(T), hipDeviceMallocFinegrained));
    n = count;
  }
  void upload(const T *src, size_t count, hipStream_t s = nullptr) {
    if (count > n || !p) alloc(count);
    if (count == 0 || !src) return;
    STEREO_HIP_CHECK(hipMemcpyAsync(p, src, count * sizeof(T), hipMemcpyHostToDevice, s));
  }
};

// Scratch that lives for a few launches on ONE stream: allocated and released in stream order
// (hipMallocAsync / hipFreeAsync), so releasing it neither waits for the kernels that use it nor
// synchronises the device the way hipFree does.
template <class T>
struct StreamBuf {
  T *p = nullptr;
  hipStream_t s = nullptr;
  StreamBuf() = default;
  StreamBuf(const StreamBuf &) = delete;
  StreamBuf &operator=(const StreamBuf &) = delete;
  ~StreamBuf() {
    if (p) (void)hipFreeAsync(p, s);
  }
  void alloc(size_t count, hipStream_t stream = nullptr) {
    if (p) (void)hipFreeAsync(p, s);
    p = nullptr;
    s = stream;
    if (count == 0) count = 1;
    STEREO_HIP_CHECK(hipMallocAsync((void **)&p, count * sizeof(T), stream));
  }
};

template <class T>
struct PinnedBuf {
  T *p = nullptr;
  size_t n = 0;
  ~PinnedBuf() {
    if (p) (void)hipHostFree(p);
  }
  void alloc(size_t count) {
    if (p) (void)hipHostFree(p);
    p = nullptr;
    if (count == 0) count = 1;
    STEREO_HIP_CHECK(hipHostMalloc((void **)&p, count * sizeof(T), hipHostMallocDefault));
    n = count;
  }
};

}  // namespace stereo
