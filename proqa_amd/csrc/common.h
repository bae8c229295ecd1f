// Shared host-side helpers for libproqa_hip.so (error plumbing, HIP checks).
#pragma once
#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdint>
#include <cstdio>

#include "../../include/proqa_hip.h"

namespace proqa {

// thread-local last-error message, surfaced through proqa_last_error()
char* error_buffer();
int fail(int code, const char* fmt, ...) __attribute__((format(printf, 2, 3)));
// one line on stderr ("[proqa] ...") when the environment variable PROQA_LOG is set to anything but "" / "0": decisions the
// library takes on its own (which scan a search ran on, a suspended / resumed int8 copy); silent otherwise
bool log_enabled();
void log_line(const char* fmt, ...) __attribute__((format(printf, 1, 2)));

inline int hip_fail(hipError_t e, const char* what, const char* file, int line) {
  return fail(PROQA_EHIP, "%s failed: %s (%s:%d)", what, hipGetErrorString(e), file, line);
}

#define PROQA_HIP(call)                                                          \
  do {                                                                           \
    hipError_t _e = (call);                                                      \
    if (_e != hipSuccess) return ::proqa::hip_fail(_e, #call, __FILE__, __LINE__); \
  } while (0)

#define PROQA_LAUNCH_CHECK()                                                              \
  do {                                                                                    \
    hipError_t _e = hipGetLastError();                                                    \
    if (_e != hipSuccess) return ::proqa::hip_fail(_e, "kernel launch", __FILE__, __LINE__); \
  } while (0)

inline hipStream_t as_stream(void* s) { return reinterpret_cast<hipStream_t>(s); }

// Makes `want` the current HIP device for the lifetime of the guard and restores the caller's device afterwards: a
// handle that lives on another GPU than the caller's current one must not change the device under the caller (its
// later allocations and stream look-ups would land on the wrong GPU).
struct DeviceGuard {
  int prev = -1;
  bool switched = false;
  hipError_t err = hipSuccess;
  explicit DeviceGuard(int want) {
    err = hipGetDevice(&prev);
    if (err == hipSuccess && prev != want) {
      err = hipSetDevice(want);
      switched = err == hipSuccess;
    }
  }
  ~DeviceGuard() {
    if (switched) (void)hipSetDevice(prev);
  }
  DeviceGuard(const DeviceGuard&) = delete;
  DeviceGuard& operator=(const DeviceGuard&) = delete;
};
#define PROQA_ON_DEVICE(dev)                   \
  ::proqa::DeviceGuard _device_guard(dev);     \
  PROQA_HIP(_device_guard.err)

// hipMalloc whose failure is an expected, recoverable condition (the caller falls back or reports PROQA_ENOMEM): HIP's
// last-error slot is sticky until read, so it is cleared here -- otherwise the next launch wrapper's hipGetLastError()
// would report this allocation failure as its own.
inline hipError_t try_malloc(void** p, size_t bytes) {
  const hipError_t e = hipMalloc(p, bytes);
  if (e != hipSuccess) (void)hipGetLastError();
  return e;
}

template <typename T>
inline T ceil_div(T a, T b) {
  return (a + b - 1) / b;
}
template <typename T>
inline T round_up(T a, T b) {
  return ceil_div(a, b) * b;
}

// number of compute units of the current device (256 on MI355X); cached
int device_cu_count();

#if defined(__HIPCC__)
// Barrier that publishes LDS-DMA'd data.  `global_load_lds` completion is tracked by vmcnt only, and
// hipcc does NOT reliably drain vmcnt at __syncthreads() (observed: `s_waitcnt lgkmcnt(0); s_barrier`
// inside a loop), so a stage could be read before its DMA had landed -- rare, data-dependent garbage.
// Every barrier that makes a DMA'd LDS stage readable goes through this.
__device__ __forceinline__ void dma_wait_barrier() {
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
}
#endif

}  // namespace proqa
