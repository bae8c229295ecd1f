// Error plumbing and device queries for libproqa_hip.so.
#include "common.h"

#include <cstring>

namespace proqa {

char* error_buffer() {
  static thread_local char buf[512] = {0};
  return buf;
}

int fail(int code, const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(error_buffer(), 512, fmt, ap);
  va_end(ap);
  return code;
}

int device_cu_count() {
  static int cached = 0;
  if (cached > 0) return cached;
  int dev = 0, n = 0;
  if (hipGetDevice(&dev) != hipSuccess) return 256;
  if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0)
    return 256;
  cached = n;
  return n;
}

}  // namespace proqa

extern "C" {

const char* proqa_last_error(void) { return proqa::error_buffer(); }

int proqa_abi_version(void) { return PROQA_ABI_VERSION; }

int proqa_device_info(int* n_devices, char* arch_name, size_t arch_name_len) {
  int n = 0;
  hipError_t e = hipGetDeviceCount(&n);
  if (e != hipSuccess || n <= 0) {
    if (n_devices) *n_devices = 0;
    if (arch_name && arch_name_len) arch_name[0] = 0;
    return proqa::fail(PROQA_ENOGPU, "no HIP device visible: %s", hipGetErrorString(e));
  }
  if (n_devices) *n_devices = n;
  if (arch_name && arch_name_len) {
    int dev = 0;
    PROQA_HIP(hipGetDevice(&dev));
    hipDeviceProp_t prop;
    PROQA_HIP(hipGetDeviceProperties(&prop, dev));
    strncpy(arch_name, prop.gcnArchName, arch_name_len - 1);
    arch_name[arch_name_len - 1] = 0;
  }
  return PROQA_OK;
}

}  // extern "C"
