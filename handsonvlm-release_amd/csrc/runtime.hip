// Error plumbing and device queries of libhvlm_hip.so.
#include "common.h"
#include "gemm_args.h"
#include <atomic>
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
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

// ---- switches ---------------------------------------------------------------------------------------------------------
namespace {
struct Key { const char* name; int Config::*field; bool dev_only; int dflt; };
const Key kKeys[] = {
    {"gemm_group_m", &Config::gemm_group_m, false, 4},     {"gemm_static", &Config::gemm_static, false, 0},
    {"gemm_tail_half", &Config::gemm_tail_half, true, 0}, {"gemm_epi_swap", &Config::gemm_epi_swap, true, 0},
    {"attn_no_resident", &Config::attn_no_resident, false, 0}, {"attn_fwd_pipe", &Config::attn_fwd_pipe, true, 0},
    {"decode_unfused", &Config::decode_unfused, false, 0}, {"decode_attn_v1", &Config::decode_attn_v1, false, 0},
    {"decode_attn_ns", &Config::decode_attn_ns, false, 0}, {"decode_engine", &Config::decode_engine, false, 0}, {"gemv_early", &Config::gemv_early, false, 1}, {"skinny_per", &Config::skinny_per, false, 0},
    {"attn_delta_separate", &Config::attn_delta_separate, false, 0},
    {"attn_fwd_asm", &Config::attn_fwd_asm, false, 1},     {"attn_clip_asm", &Config::attn_clip_asm, false, 1},
    {"gemm_dbg", &Config::gemm_dbg, true, 0},              {"gemm_seg4", &Config::gemm_seg4, true, 1},
    {"gemm_m16", &Config::gemm_m16, true, 1},              {"gemm_tn_seg4", &Config::gemm_tn_seg4, true, 1},
    {"gemm_fp8_m16", &Config::gemm_fp8_m16, true, 0},      {"attn_dkv_fused", &Config::attn_dkv_fused, true, 1},
};
#ifdef HVLM_DEV
constexpr bool kDevBuild = true;
#else
constexpr bool kDevBuild = false;
#endif
const Key* find_key(const char* name) {
  for (const Key& k : kKeys)
    if (name && strcmp(k.name, name) == 0) return &k;
  return nullptr;
}
Config load_env() {
  Config c;
  for (const Key& k : kKeys) {
    char env[64] = "HVLM_";
    size_t n = 5;
    for (const char* s = k.name; *s && n + 1 < sizeof(env); ++s) env[n++] = (char)((*s >= 'a' && *s <= 'z') ? *s - 32 : *s);
    env[n] = 0;
    const char* v = getenv(env);
    if (!v) continue;
    const int iv = atoi(v);
    if (k.dev_only && !kDevBuild && iv != k.dflt) {
      fprintf(stderr, "libhvlm_hip: %s=%d ignored: that variant is compiled into development builds only (make DEV=1)\n", env, iv);
      continue;
    }
    c.*(k.field) = iv;
  }
  if (c.gemm_group_m < 1) c.gemm_group_m = 1;
  return c;
}
}  // namespace

Config& config() {
  static Config c = load_env();       // (function-local static: initialised once, thread-safe)
  return c;
}

}  // namespace hvlm

extern "C" int hvlm_configure(const char* key, int value) {
  const hvlm::Key* k = hvlm::find_key(key);
  HVLM_REQUIRE(k, "configure: unknown switch '%s'", key ? key : "(null)");
  if (k->dev_only && !hvlm::kDevBuild && value != k->dflt) {
    hvlm::set_error("configure: %s=%d selects a variant that is compiled into development builds only (make DEV=1)", key, value);
    return HVLM_ERR_UNSUPPORTED;
  }
  HVLM_REQUIRE(strcmp(key, "gemm_group_m") != 0 || value >= 1, "configure: gemm_group_m must be >= 1");
  hvlm::config().*(k->field) = value;
  return HVLM_OK;
}

extern "C" int hvlm_config_get(const char* key, int* value) {
  const hvlm::Key* k = hvlm::find_key(key);
  HVLM_REQUIRE(k && value, "config_get: unknown switch '%s'", key ? key : "(null)");
  *value = hvlm::config().*(k->field);
  return HVLM_OK;
}

extern "C" int hvlm_is_dev_build(void) { return hvlm::kDevBuild ? 1 : 0; }

// ---- per-device handle (include/hvlm.h: hvlm_create / hvlm_destroy) ------------------------------------------------------
struct hvlm_handle {
  int device;
  void* sched_ws;       // caller-owned scheduler-state block of the persistent GEMM, or nullptr (the library's built-in block)
};
namespace hvlm {
namespace {
constexpr int kMaxDevices = 64;
std::atomic<hvlm_handle*> g_handles[kMaxDevices];      // the one table of mutable library state: device -> its handle
}  // namespace
int attn_asm_preload(int dev);    // attention_asm.hip
void* pp_sched_ws() {
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= kMaxDevices) return nullptr;
  hvlm_handle* h = g_handles[dev].load(std::memory_order_acquire);
  return h ? h->sched_ws : nullptr;
}
}  // namespace hvlm

extern "C" int hvlm_create(int device, void* sched_ws, size_t sched_ws_bytes, void* stream, hvlm_handle** out) {
  HVLM_REQUIRE(out != nullptr, "create: out is NULL");
  *out = nullptr;
  int cur = -1;
  if (hipGetDevice(&cur) != hipSuccess) { hvlm::set_error("create: no current device"); return HVLM_ERR_LAUNCH; }
  HVLM_REQUIRE(device >= 0 && device < hvlm::kMaxDevices && device == cur,
               "create: device %d is not the calling thread's current device (%d)", device, cur);
  if (sched_ws != nullptr) {
    HVLM_REQUIRE(sched_ws_bytes >= hvlm_gemm_sched_ws_bytes() && ((uintptr_t)sched_ws & 255) == 0,
                 "create: sched_ws needs %zu bytes, 256-byte aligned", hvlm_gemm_sched_ws_bytes());
    if (hipMemsetAsync(sched_ws, 0, hvlm_gemm_sched_ws_bytes(), hvlm::as_stream(stream)) != hipSuccess) {
      hvlm::set_error("create: hipMemsetAsync failed");
      return HVLM_ERR_LAUNCH;
    }
  }
  // the embedded assembly code objects are loaded here, not at their first launch (which may sit inside a hipGraph capture);
  // a load failure is not an error of create: attention.hip then dispatches the C++ kernels
  (void)hvlm::attn_asm_preload(device);
  hvlm_handle* h = new hvlm_handle{device, sched_ws};
  hvlm_handle* expected = nullptr;
  if (!hvlm::g_handles[device].compare_exchange_strong(expected, h, std::memory_order_acq_rel)) {
    delete h;
    hvlm::set_error("create: device %d already has a handle (hvlm_destroy it first)", device);
    return HVLM_ERR_INVALID;
  }
  *out = h;
  return HVLM_OK;
}

extern "C" int hvlm_destroy(hvlm_handle* handle) {
  HVLM_REQUIRE(handle != nullptr && handle->device >= 0 && handle->device < hvlm::kMaxDevices, "destroy: bad handle");
  hvlm_handle* expected = handle;
  HVLM_REQUIRE(hvlm::g_handles[handle->device].compare_exchange_strong(expected, nullptr, std::memory_order_acq_rel),
               "destroy: this handle is not the one registered for device %d", handle->device);
  delete handle;
  return HVLM_OK;
}

extern "C" int hvlm_handle_device(const hvlm_handle* handle) { return handle ? handle->device : HVLM_ERR_INVALID; }

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

#ifdef HVLM_DEV
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
#endif  // HVLM_DEV
