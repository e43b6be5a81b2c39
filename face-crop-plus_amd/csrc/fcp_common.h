// Shared helpers for the fcp_hip C ABI (error plumbing, launch checks).
#pragma once
#include <hip/hip_runtime.h>
#include <cstdarg>
#include <cstdio>
#include <atomic>

#define FCP_ERR_ARG (-1)
#define FCP_ERR_HIP (-2)

void fcp_set_error(const char* fmt, ...);

#define FCP_REQUIRE(cond, ...)                \
  do {                                        \
    if (!(cond)) {                            \
      fcp_set_error(__VA_ARGS__);             \
      return FCP_ERR_ARG;                     \
    }                                         \
  } while (0)

#define FCP_HIP_OK(expr)                                                        \
  do {                                                                          \
    hipError_t _e = (expr);                                                     \
    if (_e != hipSuccess) {                                                     \
      fcp_set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e),      \
                    __FILE__, __LINE__);                                        \
      return FCP_ERR_HIP;                                                       \
    }                                                                           \
  } while (0)

#define FCP_LAUNCH_OK()                                                         \
  do {                                                                          \
    hipError_t _e = hipGetLastError();                                          \
    if (_e != hipSuccess) {                                                     \
      fcp_set_error("kernel launch failed: %s (%s:%d)", hipGetErrorString(_e),  \
                    __FILE__, __LINE__);                                        \
      return FCP_ERR_HIP;                                                       \
    }                                                                           \
  } while (0)

static inline int fcp_cdiv(long a, long b) { return (int)((a + b - 1) / b); }

// Per-device, thread-safe opt-in to more than 64 KiB of dynamic LDS.  hipFuncSetAttribute is per device, so a
// process-wide "done" flag would leave a second GPU of the same process without the opt-in: `mask` (one static
// atomic per kernel instantiation) holds one bit per device ordinal; concurrent first launches may both set
// the attribute, which is idempotent.
#define FCP_LDS_OPT_IN(kernel_ptr, bytes)                                                          \
  do {                                                                                             \
    static std::atomic<unsigned long long> _mask{0ull};                                            \
    int _dev = 0;                                                                                  \
    FCP_HIP_OK(hipGetDevice(&_dev));                                                               \
    const unsigned long long _bit = 1ull << (_dev & 63);                                           \
    if (!(_mask.load(std::memory_order_acquire) & _bit)) {                                         \
      FCP_HIP_OK(hipFuncSetAttribute(reinterpret_cast<const void*>(kernel_ptr),                    \
                                     hipFuncAttributeMaxDynamicSharedMemorySize, (int)(bytes)));   \
      _mask.fetch_or(_bit, std::memory_order_release);                                             \
    }                                                                                              \
  } while (0)

// Compute units of the current device, cached per device ordinal (a process may drive several GPUs).
static inline int fcp_cu_count() {
  static std::atomic<int> cache[64];
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) return 256;
  std::atomic<int>& slot = cache[dev & 63];
  int n = slot.load(std::memory_order_relaxed);
  if (n == 0) {
    if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) n = 256;
    slot.store(n, std::memory_order_relaxed);
  }
  return n;
}
