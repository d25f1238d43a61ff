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

// ---- fp16 <-> f32 (the CLIP tower runs in fp16 like the reference: handsonvlm/train/train.py:316) -------------------
typedef __attribute__((ext_vector_type(2))) _Float16 f16x2;
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;
__device__ __forceinline__ float h2f(bf16_t h) { return (float)__builtin_bit_cast(_Float16, h); }
__device__ __forceinline__ bf16_t f2h(float f) { return __builtin_bit_cast(bf16_t, (_Float16)f); }     // round-to-nearest-even
__device__ __forceinline__ unsigned int pack_h2(float lo, float hi) {
  const f16x2 r = {(_Float16)lo, (_Float16)hi};
  return __builtin_bit_cast(unsigned int, r);
}
__device__ __forceinline__ float h_lo(unsigned int p) { return (float)__builtin_bit_cast(f16x2, p)[0]; }
__device__ __forceinline__ float h_hi(unsigned int p) { return (float)__builtin_bit_cast(f16x2, p)[1]; }

// 16-bit element type selected at compile time: F16 = false -> bfloat16, true -> IEEE half.  Storage is always uint16.
template <bool F16>
struct E16 {
  static __device__ __forceinline__ float to_f(bf16_t h) { if constexpr (F16) return h2f(h); else return bf2f(h); }
  static __device__ __forceinline__ bf16_t from_f(float f) { if constexpr (F16) return f2h(f); else return f2bf(f); }
  static __device__ __forceinline__ unsigned int pack2(float lo, float hi) { if constexpr (F16) return pack_h2(lo, hi); else return pack_bf2(lo, hi); }
  static __device__ __forceinline__ float lo(unsigned int p) { if constexpr (F16) return h_lo(p); else return bf_lo(p); }
  static __device__ __forceinline__ float hi(unsigned int p) { if constexpr (F16) return h_hi(p); else return bf_hi(p); }
  static __device__ __forceinline__ f32x16 mfma32(bf16x8 a, bf16x8 b, f32x16 c) {
    if constexpr (F16) return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
    else return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
  }
  // acc + lo(p) + hi(p) in one instruction (v_dot2c_f32_f16 / _bf16 against packed ones)
  static __device__ __forceinline__ float add_pair(unsigned int p, float acc) {
    typedef __attribute__((ext_vector_type(2))) _Float16 h2_t;
    typedef __attribute__((ext_vector_type(2))) __bf16 b2_t;
    if constexpr (F16) return __builtin_amdgcn_fdot2(__builtin_bit_cast(h2_t, p), __builtin_bit_cast(h2_t, 0x3c003c00u), acc, false);
    else return __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(b2_t, p), __builtin_bit_cast(b2_t, 0x3f803f80u), acc, false);
  }
};

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
// The same reductions on DPP row operations + four readlanes (no LDS-crossbar shuffles: ~80 cycles instead of ~500 of dependent
// ds_bpermute latency).  The max is order-independent, so wave_max_dpp == wave_max bit for bit; wave_sum_dpp adds in a different
// order than wave_sum and is used only where no other kernel has to reproduce the bits.
template <bool MAX>
__device__ __forceinline__ float wave_reduce_dpp(float v) {
  auto op = [](float a, float b) { return MAX ? fmaxf(a, b) : a + b; };
  v = op(v, __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0xB1, 0xF, 0xF, true)));    // quad_perm [1,0,3,2]
  v = op(v, __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x4E, 0xF, 0xF, true)));    // quad_perm [2,3,0,1]
  v = op(v, __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x141, 0xF, 0xF, true)));   // row_half_mirror
  v = op(v, __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x140, 0xF, 0xF, true)));   // row_mirror
  const int b = __float_as_int(v);                       // every lane of a 16-lane row now holds its row's result
  const float r0 = __int_as_float(__builtin_amdgcn_readlane(b, 0)), r1 = __int_as_float(__builtin_amdgcn_readlane(b, 16));
  const float r2 = __int_as_float(__builtin_amdgcn_readlane(b, 32)), r3 = __int_as_float(__builtin_amdgcn_readlane(b, 48));
  return op(op(r0, r1), op(r2, r3));
}
__device__ __forceinline__ float wave_max_dpp(float v) { return wave_reduce_dpp<true>(v); }
__device__ __forceinline__ float wave_sum_dpp(float v) { return wave_reduce_dpp<false>(v); }

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

// ---- library switches (host) ----------------------------------------------------------------
// Every A/B / development switch of the library is ONE named integer held here.  It is initialised once, when the library
// is first used, from the environment variable HVLM_<KEY IN CAPITALS> and can be changed at run time through the C ABI
// (hvlm_configure); no entry point calls getenv.  Switches that select a kernel variant compiled only into development
// builds (`make DEV=1`, -DHVLM_DEV) are refused by hvlm_configure in a production build.
struct Config {
  int gemm_group_m = 4;       // row panels per group of the XCD-aware tile order
  int gemm_static = 0;        // 1: static tile assignment even when there are >= 3 tiles per workgroup
  int attn_no_resident = 0;   // 1: CLIP attention on the streaming kernel instead of the resident-K/V kernel
  int gemv_early = 1;         // 1 (default, round 6): the decode GEMVs request x, then their first weight block, and run the prologue's
                              // arithmetic while the block is in flight (gemv_early_kernel: 2.87 -> 2.75 ms per token); 0: gemv_kernel
  int decode_engine = 0;      // 1: the batch-1 decode step as ONE persistent launch (decode_stack_kernel) instead of 5 launches per layer
  int decode_attn_ns = 0;     // EXPERIMENT: cap on the key splits of the lane-group decode attention (0 = min(32, 1024 / H))
  int decode_attn_v1 = 0;     // 1: the round-3 decode attention (one key per thread) instead of the lane-group kernel of round 6 (A/B)
  int decode_unfused = 0;     // 1: RoPE and the split combine of the native decode step as separate launches
  int attn_delta_separate = 0;  // 1: delta = rowsum(O * dO) of the attention backward by its own pass instead of inside the dQ kernel
  int skinny_per = 0;         // weight-streaming GEMM: forced K-slice length (x 256 k), 0 = balance model
  int attn_clip_asm = 1;      // 1: the CLIP tower's attention (fp16, S = 257, dh = 64) on the hand-scheduled kernel (csrc/asm/attn_clip_d64.py)
  int attn_fwd_asm = 1;       // 0: the dh = 128 causal forward on the C++ kernel instead of the hand-scheduled one (attention_asm.hip)
  // --- variants that exist in DEV builds only ---
  int gemm_tail_half = 0;     // 1: the last, partial round of tiles of an XCD's chunk runs as half-tile units (gemm_pp.hip)
  int gemm_epi_swap = 0;      // 1: register-exchange GEMM epilogue (v_permlane swaps) instead of the wave-private LDS buffers
  int gemm_dbg = 0;           // K-loop ablations of the ping-pong kernel (WRONG RESULTS by construction; timing / PMC only)
  int gemm_seg4 = 1;          // 0: 8-phase schedule for the K-contiguous kernels
  int gemm_m16 = 1;           // 0: 32x32x16 MFMAs instead of 16x16x32
  int gemm_tn_seg4 = 1;       // 0: 8-phase schedule for the K-strided (TN) kernel
  int gemm_fp8_m16 = 0;       // 1: 16x16x128 form of the e4m3 kernel
  int attn_dkv_fused = 1;     // 0: dV and dK in two passes
  int attn_fwd_pipe = 0;      // 1: dh = 128 bf16 forward on the intra-wave software-pipelined kernel (round-4 experiment, slower)
};
Config& config();

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
