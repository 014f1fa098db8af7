// mlf_ctx.hpp -- what the C-ABI translation units (mlf_api.hip, mlf_walk_api.hip) share: the
// grow-only device buffer, the library context (one device, one stream) and error reporting.
#pragma once
#include <hip/hip_runtime.h>

#include <cstddef>

namespace mlf {

// grow-only device buffer
struct DevBuf {
  void *p = nullptr;
  size_t cap = 0;
  hipError_t reserve(size_t bytes) {
    if (bytes <= cap) return hipSuccess;
    if (p) {
      hipError_t e = hipFree(p);
      p = nullptr;
      cap = 0;
      if (e != hipSuccess) return e;
    }
    // grow by half and in 64 KiB steps: scratch buffers shared by calls of slightly different sizes would otherwise be
    // freed and re-allocated (hipFree synchronises the device) every time a new maximum comes along
    const size_t want = (bytes + bytes / 2 + 65535) / 65536 * 65536;
    hipError_t e = hipMalloc(&p, want);
    if (e != hipSuccess) return e;
    cap = want;
    return hipSuccess;
  }
  template <class T>
  T *as() const { return static_cast<T *>(p); }
  void release() {
    if (p) (void)hipFree(p);
    p = nullptr;
    cap = 0;
  }
};

// defined in mlf_api.hip
int ctx_ensure();                    // 0 or MLF_E_NODEVICE (message set)
hipStream_t ctx_stream();
int ctx_fail_hip(hipError_t e, const char *what, const char *file, int line);   // returns -(int)e
int ctx_fail_arg(int code, const char *msg);                                    // returns code

}  // namespace mlf
