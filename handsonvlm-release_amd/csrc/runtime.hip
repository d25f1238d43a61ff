// Error plumbing and device queries of libhvlm_hip.so.
#include "common.h"
#include <stdarg.h>
#include <stdio.h>
#include <string.h>

namespace hvlm {

static thread_local char g_err[512] = "";

void set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

int check_launch(const char* what) {
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) {
    set_error("%s: launch failed: %s", what, hipGetErrorString(e));
    return HVLM_ERR_LAUNCH;
  }
  return HVLM_OK;
}

}  // namespace hvlm

extern "C" int hvlm_abi_version(void) { return HVLM_ABI_VERSION; }

extern "C" const char* hvlm_last_error(void) { return hvlm::g_err; }

extern "C" int hvlm_device_info(int device, int* cu_count, int* lds_bytes, char* arch, int arch_len) {
  hipDeviceProp_t prop;
  hipError_t e = hipGetDeviceProperties(&prop, device);
  if (e != hipSuccess) {
    hvlm::set_error("device_info: %s", hipGetErrorString(e));
    return HVLM_ERR_LAUNCH;
  }
  if (cu_count) *cu_count = prop.multiProcessorCount;
  if (lds_bytes) *lds_bytes = (int)prop.sharedMemPerBlock;
  if (arch && arch_len > 0) {
    strncpy(arch, prop.gcnArchName, arch_len - 1);
    arch[arch_len - 1] = 0;
  }
  return HVLM_OK;
}
