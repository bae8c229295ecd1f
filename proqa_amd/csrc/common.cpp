// Error plumbing and device queries for libproqa_hip.so.
#include "common.h"

#include <cstdlib>
#include <cstring>
#include <random>
#include <utility>

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

bool log_enabled() {
  static const bool on = [] {
    const char* e = getenv("PROQA_LOG");
    return e && e[0] && !(e[0] == '0' && !e[1]);
  }();
  return on;
}

void log_line(const char* fmt, ...) {
  if (!log_enabled()) return;
  char buf[512];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof buf, fmt, ap);
  va_end(ap);
  fprintf(stderr, "[proqa] %s\n", buf);
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

// faiss/utils/random.cpp rand_perm, restated: Fisher-Yates driven by std::mt19937
int proqa_rand_perm(int64_t n, int64_t seed, int32_t* perm_out) {
  if (n < 0 || n >= (1ll << 31) || (!perm_out && n > 0)) return proqa::fail(PROQA_EINVAL, "rand_perm: bad argument");
  for (int64_t i = 0; i < n; ++i) perm_out[i] = (int32_t)i;
  std::mt19937 mt((unsigned)seed);
  for (int64_t i = 0; i + 1 < n; ++i) {
    const int64_t i2 = i + (int64_t)(mt() % (uint32_t)(n - i));
    std::swap(perm_out[i], perm_out[i2]);
  }
  return PROQA_OK;
}

}  // extern "C"
