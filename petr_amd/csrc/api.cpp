// Library-level entry points: version, thread-local error text, device capabilities.
#include "common.h"

static thread_local char g_err[512] = "";

void petr_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

extern "C" int petr_version(void) { return PETR_HIP_VERSION; }
extern "C" const char* petr_last_error(void) { return g_err; }

extern "C" int petr_device_caps(int* num_cu, char* arch, int arch_len) {
  int dev = 0;
  hipError_t e = hipGetDevice(&dev);
  PETR_CHECK(e == hipSuccess, PETR_ERR_LAUNCH, "hipGetDevice: %s", hipGetErrorString(e));
  hipDeviceProp_t prop;
  e = hipGetDeviceProperties(&prop, dev);
  PETR_CHECK(e == hipSuccess, PETR_ERR_LAUNCH, "hipGetDeviceProperties: %s", hipGetErrorString(e));
  if (num_cu) *num_cu = prop.multiProcessorCount;
  if (arch && arch_len > 0) snprintf(arch, (size_t)arch_len, "%s", prop.gcnArchName);
  return PETR_OK;
}
