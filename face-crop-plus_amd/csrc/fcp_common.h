// Shared helpers for the fcp_hip C ABI (error plumbing, launch checks).
#pragma once
#include <hip/hip_runtime.h>
#include <cstdarg>
#include <cstdio>

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
