// Shared device/host helpers for the hvlm HIP kernels (gfx950 / CDNA4 only).
// Wavefront = 64 lanes everywhere in this tree.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/hvlm.h"

namespace hvlm {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(2))) float f32x2;
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;
typedef __attribute__((ext_vector_type(2))) unsigned int u32x2;
typedef __attribute__((ext_vector_type(4))) unsigned short u16x4;
typedef __attribute__((ext_vector_type(8))) unsigned short u16x8;

typedef unsigned short bf16_t;  // raw storage type used in kernel signatures

// ---- bf16 <-> f32 -----------------------------------------------------------------------------
// f32 -> bf16 uses the gfx950 hardware conversion (v_cvt_pk_bf16_f32: round-to-nearest-even, NaN
// quieted) through clang's native __bf16 conversion; a software rounding sequence costs ~8 VALU and a
// divergent NaN branch per element, which dominated the attention and GEMM epilogues.
__device__ __forceinline__ float bf2f(bf16_t h) {
  return __uint_as_float(((unsigned int)h) << 16);
}
__device__ __forceinline__ bf16_t f2bf(float f) {
  const __bf16 h = (__bf16)f;
  return __builtin_bit_cast(bf16_t, h);
}
// two floats -> packed bf16x2 in one u32 (lo in low half)
__device__ __forceinline__ unsigned int pack_bf2(float lo, float hi) {
  const f32x2 v = {lo, hi};
  const bf16x2 r = __builtin_convertvector(v, bf16x2);
  return __builtin_bit_cast(unsigned int, r);
}
__device__ __forceinline__ float bf_lo(unsigned int p) { return __uint_as_float(p << 16); }
__device__ __forceinline__ float bf_hi(unsigned int p) { return __uint_as_float(p & 0xffff0000u); }

// ---- wave / block reductions (wave = 64) -----------------------------------------------------
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}
// block-wide sum for blocks of NW waves; `red` is NW floats of LDS. All threads get the result.
template <int NW>
__device__ __forceinline__ float block_sum(float v, float* red) {
  v = wave_sum(v);
  if constexpr (NW == 1) return v;
  const int w = threadIdx.x >> 6, l = threadIdx.x & 63;
  __syncthreads();
  if (l == 0) red[w] = v;
  __syncthreads();
  float t = 0.f;
#pragma unroll
  for (int i = 0; i < NW; ++i) t += red[i];
  return t;
}
template <int NW>
__device__ __forceinline__ float block_max(float v, float* red) {
  v = wave_max(v);
  if constexpr (NW == 1) return v;
  const int w = threadIdx.x >> 6, l = threadIdx.x & 63;
  __syncthreads();
  if (l == 0) red[w] = v;
  __syncthreads();
  float t = red[0];
#pragma unroll
  for (int i = 1; i < NW; ++i) t = fmaxf(t, red[i]);
  return t;
}

// ---- error plumbing (host) ------------------------------------------------------------------
void set_error(const char* fmt, ...);
int check_launch(const char* what);

#define HVLM_REQUIRE(cond, ...)                 \
  do {                                          \
    if (!(cond)) {                              \
      ::hvlm::set_error(__VA_ARGS__);           \
      return HVLM_ERR_INVALID;                  \
    }                                           \
  } while (0)

static inline hipStream_t as_stream(void* s) { return reinterpret_cast<hipStream_t>(s); }
static inline int cdiv(long a, long b) { return (int)((a + b - 1) / b); }

}  // namespace hvlm
