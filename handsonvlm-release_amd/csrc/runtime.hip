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

// ---- diagnostic: occupy `workgroups` CUs for ~`cycles` clock cycles (64 KiB of LDS each, so that a workgroup
// of the persistent GEMM cannot share the CU).  Stands in for an RCCL collective running beside the compute stream when
// measuring how the GEMM behaves with fewer CUs than it was launched for (tools/contention_bench.py).
namespace hvlm {
namespace {
__global__ __launch_bounds__(256) void occupy_kernel(long cycles, unsigned* sink) {
  __shared__ unsigned pad[16384];
  pad[threadIdx.x] = threadIdx.x;
  __syncthreads();
  const long t0 = __builtin_readcyclecounter();
  unsigned acc = pad[(threadIdx.x * 7) & 16383];
  while ((long)__builtin_readcyclecounter() - t0 < cycles) {
    __builtin_amdgcn_s_sleep(32);
    acc += pad[acc & 16383];
  }
  if (acc == 0xdeadbeefu && sink) *sink = acc;
}
}  // namespace
}  // namespace hvlm

extern "C" int hvlm_debug_occupy(int workgroups, long cycles, void* stream) {
  HVLM_REQUIRE(workgroups > 0 && cycles > 0, "debug_occupy: bad args");
  hipLaunchKernelGGL(hvlm::occupy_kernel, dim3(workgroups), dim3(256), 0, hvlm::as_stream(stream), cycles, (unsigned*)nullptr);
  return hvlm::check_launch("debug_occupy");
}
