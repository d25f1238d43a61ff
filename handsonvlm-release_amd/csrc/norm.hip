// RMSNorm (fwd/bwd) and LayerNorm (fwd) for gfx950.  HBM-bound: every row is read once with
// 16-byte loads (8 bf16 per lane), kept in registers, reduced with wave shuffles, written once.
//
// Reference arithmetic: transformers LlamaRMSNorm.forward (fp32 variance, cast back to the
// activation dtype, then multiply by the bf16 weight) and torch LayerNorm inside
// CLIPEncoderLayer / CLIPVisionTransformer.pre_layrnorm (fp32 statistics, one rounding).
#include "common.h"

namespace hvlm {
namespace {

constexpr int MAXC = 4;  // register chunks of 8 elements per thread: D <= TPR*8*MAXC

// TPR = threads per row (64 -> one wave per row, 256 -> one block per row).
template <int TPR>
__device__ __forceinline__ float row_sum(float v, float* red) {
  if constexpr (TPR == 64) return wave_sum(v);
  else return block_sum<TPR / 64>(v, red);
}

template <int TPR>
__global__ __launch_bounds__(256) void rmsnorm_fwd_kernel(const bf16_t* __restrict__ x,
                                                          const bf16_t* __restrict__ w,
                                                          bf16_t* __restrict__ y,
                                                          float* __restrict__ rstd_out, int rows,
                                                          int D, float eps) {
  __shared__ float red[4];
  constexpr int RPB = 256 / TPR;
  const int t = threadIdx.x % TPR;
  const int row = blockIdx.x * RPB + threadIdx.x / TPR;
  const bool active = row < rows;
  const bf16_t* xr = x + (size_t)(active ? row : 0) * D;
  u32x4 v[MAXC];
  float ss = 0.f;
#pragma unroll
  for (int c = 0; c < MAXC; ++c) {
    const int i = (c * TPR + t) * 8;
    if (i < D) {
      v[c] = *reinterpret_cast<const u32x4*>(xr + i);
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float a = bf_lo(v[c][e]), b = bf_hi(v[c][e]);
        ss += a * a + b * b;
      }
    }
  }
  ss = row_sum<TPR>(ss, red);
  const float rstd = rsqrtf(ss / (float)D + eps);
  if (!active) return;
  if (rstd_out && t == 0) rstd_out[row] = rstd;
  bf16_t* yr = y + (size_t)row * D;
#pragma unroll
  for (int c = 0; c < MAXC; ++c) {
    const int i = (c * TPR + t) * 8;
    if (i < D) {
      const u32x4 wv = *reinterpret_cast<const u32x4*>(w + i);
      u32x4 o;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        // bf16(x*rstd) first (the reference casts back before the weight multiply)
        const float a = bf2f(f2bf(bf_lo(v[c][e]) * rstd)) * bf_lo(wv[e]);
        const float b = bf2f(f2bf(bf_hi(v[c][e]) * rstd)) * bf_hi(wv[e]);
        o[e] = pack_bf2(a, b);
      }
      *reinterpret_cast<u32x4*>(yr + i) = o;
    }
  }
}

// grid = HVLM_NORM_BWD_BLOCKS blocks of 256 threads; block b walks rows b, b+grid, ...
// dx = add + rstd * (g - xhat * mean(g * xhat)),  g = dy * w ; dw_partial[b] = sum dy * xhat.
// NC = register chunks of 8 elements per thread = ceil(D / 2048): D = 4096 (7B) needs 2, and with NC = NC = 4 for every D the
// kernel held 240 VGPRs (2 waves per SIMD); templated, the 7B instantiation keeps 4 waves per SIMD in flight.
template <int NC>
__global__ __launch_bounds__(256) void rmsnorm_bwd_kernel(const bf16_t* __restrict__ dy,
                                                          const bf16_t* __restrict__ x,
                                                          const bf16_t* __restrict__ w,
                                                          const float* __restrict__ rstd_in,
                                                          const bf16_t* __restrict__ add,
                                                          bf16_t* __restrict__ dx,
                                                          float* __restrict__ dw_partial, int rows,
                                                          int D) {
  __shared__ float red[4];
  const int t = threadIdx.x;
  float wreg[NC][8];
  float dwacc[NC][8];
#pragma unroll
  for (int c = 0; c < NC; ++c) {
    const int i = (c * 256 + t) * 8;
#pragma unroll
    for (int e = 0; e < 8; ++e) { dwacc[c][e] = 0.f; wreg[c][e] = 0.f; }
    if (i < D) {
      const u32x4 wv = *reinterpret_cast<const u32x4*>(w + i);
#pragma unroll
      for (int e = 0; e < 4; ++e) { wreg[c][2 * e] = bf_lo(wv[e]); wreg[c][2 * e + 1] = bf_hi(wv[e]); }
    }
  }
  // the x / dy (/ add) chunks of the NEXT row are fetched before the block-wide reduction of the current one: the row loop
  // has two barriers per row and nothing else to hide the HBM latency behind (4.1 TB/s without the prefetch)
  u32x4 nx[NC], ng[NC], na[NC];
  float nrstd = 0.f;
  auto fetch = [&](int row) {
    const size_t off = (size_t)row * D;
    nrstd = rstd_in[row];
#pragma unroll
    for (int c = 0; c < NC; ++c) {
      const int i = (c * 256 + t) * 8;
      if (i < D) {
        nx[c] = *reinterpret_cast<const u32x4*>(x + off + i);
        ng[c] = *reinterpret_cast<const u32x4*>(dy + off + i);
        if (add) na[c] = *reinterpret_cast<const u32x4*>(add + off + i);
      }
    }
  };
  if ((int)blockIdx.x < rows) fetch(blockIdx.x);
  for (int row = blockIdx.x; row < rows; row += gridDim.x) {
    const size_t off = (size_t)row * D;
    const float rstd = nrstd;
    u32x4 cx[NC], cg[NC], ca[NC];
#pragma unroll
    for (int c = 0; c < NC; ++c) { cx[c] = nx[c]; cg[c] = ng[c]; ca[c] = na[c]; }
    if (row + (int)gridDim.x < rows) fetch(row + gridDim.x);
    float xh[NC][8], g[NC][8];
    float dot = 0.f;
#pragma unroll
    for (int c = 0; c < NC; ++c) {
      const int i = (c * 256 + t) * 8;
      if (i < D) {
        const u32x4 xv = cx[c];
        const u32x4 gv = cg[c];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const float x0 = bf_lo(xv[e]) * rstd, x1 = bf_hi(xv[e]) * rstd;
          const float d0 = bf_lo(gv[e]), d1 = bf_hi(gv[e]);
          xh[c][2 * e] = x0; xh[c][2 * e + 1] = x1;
          dwacc[c][2 * e] += d0 * x0; dwacc[c][2 * e + 1] += d1 * x1;
          g[c][2 * e] = d0 * wreg[c][2 * e]; g[c][2 * e + 1] = d1 * wreg[c][2 * e + 1];
          dot += g[c][2 * e] * x0 + g[c][2 * e + 1] * x1;
        }
      }
    }
    dot = block_sum<4>(dot, red);
    const float mean = dot / (float)D;
#pragma unroll
    for (int c = 0; c < NC; ++c) {
      const int i = (c * 256 + t) * 8;
      if (i < D) {
        u32x4 av = {0u, 0u, 0u, 0u};
        if (add) av = ca[c];
        u32x4 o;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const float r0 = rstd * (g[c][2 * e] - xh[c][2 * e] * mean) + bf_lo(av[e]);
          const float r1 = rstd * (g[c][2 * e + 1] - xh[c][2 * e + 1] * mean) + bf_hi(av[e]);
          o[e] = pack_bf2(r0, r1);
        }
        *reinterpret_cast<u32x4*>(dx + off + i) = o;
      }
    }
  }
  float* dwp = dw_partial + (size_t)blockIdx.x * D;
#pragma unroll
  for (int c = 0; c < NC; ++c) {
    const int i = (c * 256 + t) * 8;
    if (i < D) {
      f32x4 a = {dwacc[c][0], dwacc[c][1], dwacc[c][2], dwacc[c][3]};
      f32x4 b = {dwacc[c][4], dwacc[c][5], dwacc[c][6], dwacc[c][7]};
      *reinterpret_cast<f32x4*>(dwp + i) = a;
      *reinterpret_cast<f32x4*>(dwp + i + 4) = b;
    }
  }
}

// out[d] = sum_r in[r][d] (fixed summation order: deterministic).  A block owns 64 columns; its 4 waves take rows
// r = wave, wave+4, ... with 4 independent accumulators each (loads in flight), then combine through LDS.
__global__ __launch_bounds__(256) void reduce_rows_kernel(const float* __restrict__ in,
                                                          void* __restrict__ out, int rows, int D,
                                                          int out_bf16) {
  __shared__ float part[4][64];
  const int col = threadIdx.x & 63, rg = threadIdx.x >> 6;
  const int d = blockIdx.x * 64 + col;
  float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
  if (d < D) {
    int r = rg;
    for (; r + 12 < rows; r += 16) {
      s0 += in[(size_t)r * D + d];
      s1 += in[(size_t)(r + 4) * D + d];
      s2 += in[(size_t)(r + 8) * D + d];
      s3 += in[(size_t)(r + 12) * D + d];
    }
    for (; r < rows; r += 4) s0 += in[(size_t)r * D + d];
  }
  part[rg][col] = (s0 + s1) + (s2 + s3);
  __syncthreads();
  if (rg == 0 && d < D) {
    const float s = (part[0][col] + part[1][col]) + (part[2][col] + part[3][col]);
    if (out_bf16) reinterpret_cast<bf16_t*>(out)[d] = f2bf(s);
    else reinterpret_cast<float*>(out)[d] = s;
  }
}

// Q8: the output row is written as e4m3 with one scale per row (the input of an fp8 GEMM: BASELINE config 5's CLIP tower),
// bit-identical to quantize_fp8_rowwise of the 16-bit output — the row is already in registers, so the separate pass (a
// read of the 16-bit row and a second launch) disappears and the write is one byte per element.
template <int TPR, bool F16, bool Q8 = false>
__global__ __launch_bounds__(256) void layernorm_fwd_kernel(const bf16_t* __restrict__ x,
                                                            const bf16_t* __restrict__ w,
                                                            const bf16_t* __restrict__ b,
                                                            bf16_t* __restrict__ y, int rows, int D,
                                                            float eps, float* __restrict__ inv_scale = nullptr,
                                                            float* __restrict__ next_inv = nullptr, float bound_mul = 0.f,
                                                            float bound_add = 0.f) {
  __shared__ float red[4];
  constexpr int RPB = 256 / TPR;
  const int t = threadIdx.x % TPR;
  const int row = blockIdx.x * RPB + threadIdx.x / TPR;
  const bool active = row < rows;
  const bf16_t* xr = x + (size_t)(active ? row : 0) * D;
  float v[MAXC][8];
  float s = 0.f;
#pragma unroll
  for (int c = 0; c < MAXC; ++c) {
    const int i = (c * TPR + t) * 8;
    if (i < D) {
      const u32x4 xv = *reinterpret_cast<const u32x4*>(xr + i);
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        v[c][2 * e] = E16<F16>::lo(xv[e]); v[c][2 * e + 1] = E16<F16>::hi(xv[e]);
        s += v[c][2 * e] + v[c][2 * e + 1];
      }
    }
  }
  s = row_sum<TPR>(s, red);
  const float mean = s / (float)D;
  float ss = 0.f;
#pragma unroll
  for (int c = 0; c < MAXC; ++c) {
    const int i = (c * TPR + t) * 8;
    if (i < D) {
#pragma unroll
      for (int e = 0; e < 8; ++e) { const float d = v[c][e] - mean; ss += d * d; }
    }
  }
  ss = row_sum<TPR>(ss, red);
  const float rstd = rsqrtf(ss / (float)D + eps);
  if constexpr (!Q8) {
    if (!active) return;
    bf16_t* yr = y + (size_t)row * D;
#pragma unroll
    for (int c = 0; c < MAXC; ++c) {
      const int i = (c * TPR + t) * 8;
      if (i < D) {
        const u32x4 wv = *reinterpret_cast<const u32x4*>(w + i);
        const u32x4 bv = *reinterpret_cast<const u32x4*>(b + i);
        u32x4 o;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const float r0 = (v[c][2 * e] - mean) * rstd * E16<F16>::lo(wv[e]) + E16<F16>::lo(bv[e]);
          const float r1 = (v[c][2 * e + 1] - mean) * rstd * E16<F16>::hi(wv[e]) + E16<F16>::hi(bv[e]);
          o[e] = E16<F16>::pack2(r0, r1);
        }
        *reinterpret_cast<u32x4*>(yr + i) = o;
      }
    }
  } else {
    float mx = 0.f, sq = 0.f;
#pragma unroll
    for (int c = 0; c < MAXC; ++c) {
      const int i = (c * TPR + t) * 8;
      if (i < D) {
        const u32x4 wv = *reinterpret_cast<const u32x4*>(w + i);
        const u32x4 bv = *reinterpret_cast<const u32x4*>(b + i);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const float r0 = (v[c][2 * e] - mean) * rstd * E16<F16>::lo(wv[e]) + E16<F16>::lo(bv[e]);
          const float r1 = (v[c][2 * e + 1] - mean) * rstd * E16<F16>::hi(wv[e]) + E16<F16>::hi(bv[e]);
          const unsigned pk = E16<F16>::pack2(r0, r1);                    // the 16-bit output the two-pass path would store
          v[c][2 * e] = E16<F16>::lo(pk); v[c][2 * e + 1] = E16<F16>::hi(pk);
          mx = fmaxf(mx, fmaxf(fabsf(v[c][2 * e]), fabsf(v[c][2 * e + 1])));
          sq += v[c][2 * e] * v[c][2 * e] + v[c][2 * e + 1] * v[c][2 * e + 1];
        }
      }
    }
    if constexpr (TPR == 64) mx = wave_max_dpp(mx); else mx = block_max<TPR / 64>(mx, red);
    if (next_inv) {
      if constexpr (TPR == 64) sq = wave_sum_dpp(sq); else sq = row_sum<TPR>(sq, red);
      if (active && t == 0) next_inv[row] = (sqrtf(sq) * bound_mul + bound_add) * (1.f / 448.f);
    }
    if (!active) return;
    const float amax = fmaxf(mx, 1e-12f);
    const float sc = 448.f / amax;
    if (t == 0) inv_scale[row] = amax / 448.f;
    unsigned char* yr = reinterpret_cast<unsigned char*>(y) + (size_t)row * D;
#pragma unroll
    for (int c = 0; c < MAXC; ++c) {
      const int i = (c * TPR + t) * 8;
      if (i < D) {
        float f[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) f[e] = __builtin_amdgcn_fmed3f(v[c][e] * sc, -448.f, 448.f);
        unsigned lo = 0, hi = 0;
        lo = __builtin_amdgcn_cvt_pk_fp8_f32(f[0], f[1], lo, false);
        lo = __builtin_amdgcn_cvt_pk_fp8_f32(f[2], f[3], lo, true);
        hi = __builtin_amdgcn_cvt_pk_fp8_f32(f[4], f[5], hi, false);
        hi = __builtin_amdgcn_cvt_pk_fp8_f32(f[6], f[7], hi, true);
        const u32x2 o = {lo, hi};
        *reinterpret_cast<u32x2*>(yr + i) = o;
      }
    }
  }
}

// RMSNorm -> e4m3 for rows of D = 2048 / 4096 / 5120 elements (the decoder's hidden size: 7B 4096, 13B 5120; NC = ceil(D / 2048)
// register chunks per thread, the last one predicated when D is not a multiple of 2048), one workgroup per row, straight
// line: the input of the q|k|v and gate|up GEMMs of the fp8 prefill, bit-identical to rmsnorm_fwd_kernel followed by the
// rowwise quantise kernel (same element -> thread mapping and reduction order; same roundings: bf16(x * rstd), then * w, then bf16).
template <int NC, int D>
__global__ __launch_bounds__(256) void rmsnorm_fwd_q8_kernel(const bf16_t* __restrict__ x, const bf16_t* __restrict__ w,
                                                             unsigned char* __restrict__ y8, float* __restrict__ inv_scale,
                                                             float eps) {
  __shared__ float red[4];
  static_assert(D <= 2048 * NC && D > 2048 * (NC - 1) && D % 8 == 0, "NC = ceil(D / 2048)");
  constexpr bool FULL = (D == 2048 * NC);
  const int t = threadIdx.x;
  const size_t row = blockIdx.x;
  const bf16_t* xr = x + row * D;
  u32x4 xv[NC], wv[NC];
  bool on[NC];
#pragma unroll
  for (int c = 0; c < NC; ++c) on[c] = FULL || (c * 256 + t) * 8 < D;
#pragma unroll
  for (int c = 0; c < NC; ++c) { xv[c] = u32x4{0u, 0u, 0u, 0u}; if (on[c]) xv[c] = *reinterpret_cast<const u32x4*>(xr + (c * 256 + t) * 8); }
#pragma unroll
  for (int c = 0; c < NC; ++c) { wv[c] = u32x4{0u, 0u, 0u, 0u}; if (on[c]) wv[c] = *reinterpret_cast<const u32x4*>(w + (c * 256 + t) * 8); }
  float ss = 0.f;
#pragma unroll
  for (int c = 0; c < NC; ++c)
    if (on[c]) {
#pragma unroll
      for (int e = 0; e < 4; ++e) { const float a = bf_lo(xv[c][e]), b = bf_hi(xv[c][e]); ss += a * a + b * b; }
    }
  ss = block_sum<4>(ss, red);
  const float rstd = rsqrtf(ss / (float)D + eps);
  float v[NC][8];
  float mx = 0.f;
#pragma unroll
  for (int c = 0; c < NC; ++c)
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const float a = bf2f(f2bf(bf_lo(xv[c][e]) * rstd)) * bf_lo(wv[c][e]);
      const float b = bf2f(f2bf(bf_hi(xv[c][e]) * rstd)) * bf_hi(wv[c][e]);
      const unsigned pk = pack_bf2(a, b);
      v[c][2 * e] = bf_lo(pk); v[c][2 * e + 1] = bf_hi(pk);
      mx = fmaxf(mx, fmaxf(fabsf(v[c][2 * e]), fabsf(v[c][2 * e + 1])));        // (predicated-off chunks hold zeros)
    }
  mx = block_max<4>(mx, red);
  const float amax = fmaxf(mx, 1e-12f);
  const float sc = 448.f / amax;
  if (t == 0) inv_scale[row] = amax / 448.f;
  unsigned char* yr = y8 + row * D;
#pragma unroll
  for (int c = 0; c < NC; ++c) {
    float f[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) f[e] = __builtin_amdgcn_fmed3f(v[c][e] * sc, -448.f, 448.f);
    unsigned lo = 0, hi = 0;
    lo = __builtin_amdgcn_cvt_pk_fp8_f32(f[0], f[1], lo, false);
    lo = __builtin_amdgcn_cvt_pk_fp8_f32(f[2], f[3], lo, true);
    hi = __builtin_amdgcn_cvt_pk_fp8_f32(f[4], f[5], hi, false);
    hi = __builtin_amdgcn_cvt_pk_fp8_f32(f[6], f[7], hi, true);
    const u32x2 o = {lo, hi};
    if (on[c]) *reinterpret_cast<u32x2*>(yr + (c * 256 + t) * 8) = o;
  }
}

// LayerNorm -> e4m3 for rows of exactly 512 * NC elements (CLIP ViT-L: 1024), one wave per row, no predicates: the NC loads of
// the row and the 2 * NC loads of weight / bias are all in flight before the first wait.  The general kernel above issues each
// chunk under its own `i < D` branch with a vmcnt(0) behind it — six dependent memory latencies per row, which the 16-bit
// LayerNorm hides (it is HBM-bound at 6.3 TB/s) and the shorter-write e4m3 form does not (307 us for 1.26 GB).  Same
// arithmetic, same order: bit-identical to the general kernel.
template <int NC>
__global__ __launch_bounds__(256) void layernorm_fwd_q8_fast_kernel(const bf16_t* __restrict__ x, const bf16_t* __restrict__ w,
                                                                    const bf16_t* __restrict__ b, unsigned char* __restrict__ y8,
                                                                    int rows, float eps, float* __restrict__ inv_scale,
                                                                    float* __restrict__ next_inv, float bound_mul, float bound_add) {
  constexpr int D = 512 * NC;
  const int t = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;                           // whole waves: no barrier in this kernel
  const bf16_t* xr = x + (size_t)row * D;
  u32x4 xv[NC], wv[NC], bv[NC];
#pragma unroll
  for (int c = 0; c < NC; ++c) xv[c] = *reinterpret_cast<const u32x4*>(xr + (c * 64 + t) * 8);
#pragma unroll
  for (int c = 0; c < NC; ++c) {
    wv[c] = *reinterpret_cast<const u32x4*>(w + (c * 64 + t) * 8);
    bv[c] = *reinterpret_cast<const u32x4*>(b + (c * 64 + t) * 8);
  }
  float v[NC][8];
  float s = 0.f;
#pragma unroll
  for (int c = 0; c < NC; ++c)
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      v[c][2 * e] = bf_lo(xv[c][e]); v[c][2 * e + 1] = bf_hi(xv[c][e]);
      s += v[c][2 * e] + v[c][2 * e + 1];
    }
  s = wave_sum(s);
  const float mean = s / (float)D;
  float ss = 0.f;
#pragma unroll
  for (int c = 0; c < NC; ++c)
#pragma unroll
    for (int e = 0; e < 8; ++e) { const float d = v[c][e] - mean; ss += d * d; }
  ss = wave_sum(ss);
  const float rstd = rsqrtf(ss / (float)D + eps);
  float mx = 0.f, sq = 0.f;
#pragma unroll
  for (int c = 0; c < NC; ++c)
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const float r0 = (v[c][2 * e] - mean) * rstd * bf_lo(wv[c][e]) + bf_lo(bv[c][e]);
      const float r1 = (v[c][2 * e + 1] - mean) * rstd * bf_hi(wv[c][e]) + bf_hi(bv[c][e]);
      const unsigned pk = pack_bf2(r0, r1);
      v[c][2 * e] = bf_lo(pk); v[c][2 * e + 1] = bf_hi(pk);
      mx = fmaxf(mx, fmaxf(fabsf(v[c][2 * e]), fabsf(v[c][2 * e + 1])));
      sq += v[c][2 * e] * v[c][2 * e] + v[c][2 * e + 1] * v[c][2 * e + 1];
    }
  mx = wave_max_dpp(mx);
  if (next_inv) {
    sq = wave_sum_dpp(sq);
    if (t == 0) next_inv[row] = (sqrtf(sq) * bound_mul + bound_add) * (1.f / 448.f);
  }
  const float amax = fmaxf(mx, 1e-12f);
  const float sc = 448.f / amax;
  if (t == 0) inv_scale[row] = amax / 448.f;
  unsigned char* yr = y8 + (size_t)row * D;
#pragma unroll
  for (int c = 0; c < NC; ++c) {
    float f[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) f[e] = __builtin_amdgcn_fmed3f(v[c][e] * sc, -448.f, 448.f);
    unsigned lo = 0, hi = 0;
    lo = __builtin_amdgcn_cvt_pk_fp8_f32(f[0], f[1], lo, false);
    lo = __builtin_amdgcn_cvt_pk_fp8_f32(f[2], f[3], lo, true);
    hi = __builtin_amdgcn_cvt_pk_fp8_f32(f[4], f[5], hi, false);
    hi = __builtin_amdgcn_cvt_pk_fp8_f32(f[6], f[7], hi, true);
    const u32x2 o = {lo, hi};
    *reinterpret_cast<u32x2*>(yr + (c * 64 + t) * 8) = o;
  }
}

}  // namespace
}  // namespace hvlm

using namespace hvlm;

extern "C" int hvlm_rmsnorm_fwd(const void* x, const void* w, void* y, float* rstd, int rows, int D,
                                float eps, void* stream) {
  HVLM_REQUIRE(x && w && y, "rmsnorm_fwd: null pointer");
  HVLM_REQUIRE(rows > 0 && D > 0 && D % 8 == 0 && D <= 256 * 8 * MAXC, "rmsnorm_fwd: bad D=%d", D);
  if (D <= 64 * 8 * MAXC && D <= 2048) {
    hipLaunchKernelGGL((rmsnorm_fwd_kernel<64>), dim3(cdiv(rows, 4)), dim3(256), 0, as_stream(stream),
                       (const bf16_t*)x, (const bf16_t*)w, (bf16_t*)y, rstd, rows, D, eps);
  } else {
    hipLaunchKernelGGL((rmsnorm_fwd_kernel<256>), dim3(rows), dim3(256), 0, as_stream(stream),
                       (const bf16_t*)x, (const bf16_t*)w, (bf16_t*)y, rstd, rows, D, eps);
  }
  return check_launch("rmsnorm_fwd");
}

extern "C" int hvlm_rmsnorm_bwd(const void* dy, const void* x, const void* w, const float* rstd,
                                const void* add, void* dx, float* dw_partial, int rows, int D,
                                void* stream) {
  HVLM_REQUIRE(dy && x && w && rstd && dx && dw_partial, "rmsnorm_bwd: null pointer");
  HVLM_REQUIRE(rows > 0 && D > 0 && D % 8 == 0 && D <= 256 * 8 * MAXC, "rmsnorm_bwd: bad D=%d", D);
#define HVLM_RB(NC_)                                                                                            \
  hipLaunchKernelGGL(rmsnorm_bwd_kernel<NC_>, dim3(HVLM_NORM_BWD_BLOCKS), dim3(256), 0, as_stream(stream),     \
                     (const bf16_t*)dy, (const bf16_t*)x, (const bf16_t*)w, rstd, (const bf16_t*)add,           \
                     (bf16_t*)dx, dw_partial, rows, D)
  const int nc = (D + 2047) / 2048;
  if (nc <= 1) HVLM_RB(1); else if (nc == 2) HVLM_RB(2); else if (nc == 3) HVLM_RB(3); else HVLM_RB(4);
#undef HVLM_RB
  return check_launch("rmsnorm_bwd");
}

extern "C" int hvlm_reduce_rows_f32(const float* in, void* out, int rows, int D, int out_bf16,
                                    void* stream) {
  HVLM_REQUIRE(in && out && rows > 0 && D > 0, "reduce_rows: bad args");
  hipLaunchKernelGGL(reduce_rows_kernel, dim3(cdiv(D, 64)), dim3(256), 0, as_stream(stream), in, out,
                     rows, D, out_bf16);
  return check_launch("reduce_rows_f32");
}

static int layernorm_impl(const void* x, const void* w, const void* b, void* y, int rows, int D, float eps, bool f16,
                          void* stream) {
  HVLM_REQUIRE(x && w && b && y, "layernorm_fwd: null pointer");
  HVLM_REQUIRE(rows > 0 && D > 0 && D % 8 == 0 && D <= 256 * 8 * MAXC, "layernorm_fwd: bad D=%d", D);
#define HVLM_LN(TPR, F, GRID)                                                                                         \
  hipLaunchKernelGGL((layernorm_fwd_kernel<TPR, F>), dim3(GRID), dim3(256), 0, as_stream(stream), (const bf16_t*)x,   \
                     (const bf16_t*)w, (const bf16_t*)b, (bf16_t*)y, rows, D, eps)
  if (D <= 2048) { if (f16) HVLM_LN(64, true, cdiv(rows, 4)); else HVLM_LN(64, false, cdiv(rows, 4)); }
  else { if (f16) HVLM_LN(256, true, rows); else HVLM_LN(256, false, rows); }
#undef HVLM_LN
  return check_launch("layernorm_fwd");
}

extern "C" int hvlm_layernorm_fwd(const void* x, const void* w, const void* b, void* y, int rows, int D, float eps,
                                  void* stream) {
  return layernorm_impl(x, w, b, y, rows, D, eps, false, stream);
}

extern "C" int hvlm_rmsnorm_fwd_fp8(const void* x, const void* w, void* out8, float* inv_scale, int rows, int D, float eps,
                                    void* stream) {
  HVLM_REQUIRE(x && w && out8 && inv_scale && rows > 0, "rmsnorm_fwd_fp8: bad args");
  HVLM_REQUIRE(D == 4096 || D == 2048 || D == 5120,
               "rmsnorm_fwd_fp8: D must be 2048, 4096 or 5120 (got %d): use hvlm_rmsnorm_fwd + hvlm_quantize_fp8_rowwise", D);
  HVLM_REQUIRE(((uintptr_t)x & 15) == 0 && ((uintptr_t)w & 15) == 0 && ((uintptr_t)out8 & 7) == 0, "rmsnorm_fwd_fp8: alignment");
  if (D == 4096)
    hipLaunchKernelGGL((rmsnorm_fwd_q8_kernel<2, 4096>), dim3(rows), dim3(256), 0, as_stream(stream), (const bf16_t*)x, (const bf16_t*)w,
                       (unsigned char*)out8, inv_scale, eps);
  else if (D == 5120)
    hipLaunchKernelGGL((rmsnorm_fwd_q8_kernel<3, 5120>), dim3(rows), dim3(256), 0, as_stream(stream), (const bf16_t*)x, (const bf16_t*)w,
                       (unsigned char*)out8, inv_scale, eps);
  else
    hipLaunchKernelGGL((rmsnorm_fwd_q8_kernel<1, 2048>), dim3(rows), dim3(256), 0, as_stream(stream), (const bf16_t*)x, (const bf16_t*)w,
                       (unsigned char*)out8, inv_scale, eps);
  return check_launch("rmsnorm_fwd_fp8");
}

namespace hvlm {
namespace {
// (mean, rstd) of every row from the per-wave partial sums the producer GEMM's epilogue wrote (hvlm_gemm_f16_nt_ln): one thread
// per row, the n_parts partials added in index order — deterministic.  var = E[x^2] - mean^2 in fp32.
__global__ __launch_bounds__(256) void ln_stats_finalize_kernel(const float* __restrict__ part, float* __restrict__ stats, int rows,
                                                                int n_parts, float inv_d, float eps) {
  const int r = blockIdx.x * 256 + threadIdx.x;
  if (r >= rows) return;
  const f32x2* p = reinterpret_cast<const f32x2*>(part) + (size_t)r * n_parts;
  float s1 = 0.f, s2 = 0.f;
  for (int i = 0; i < n_parts; ++i) { const f32x2 v = p[i]; s1 += v[0]; s2 += v[1]; }
  const float mean = s1 * inv_d;
  const float var = fmaxf(s2 * inv_d - mean * mean, 0.f);
  const f32x2 o = {mean, rsqrtf(var + eps)};
  reinterpret_cast<f32x2*>(stats)[r] = o;
}
// n_parts a power of two <= 64 (D = 1024: 16): the n_parts partials of a row sit in adjacent lanes, so a wave reads 512
// contiguous bytes per instruction (the per-row loop above reads 8 bytes at a 8*n_parts-byte lane stride: 121 us instead of
// ~20 us for the ViT-L tower's 411 200 rows, profiles/r03i) and the row sum is a log2(n_parts)-step lane exchange.
template <int NP>
__global__ __launch_bounds__(256) void ln_stats_finalize_lanes_kernel(const float* __restrict__ part, float* __restrict__ stats,
                                                                      long total, float inv_d, float eps) {
  const long e = (long)blockIdx.x * 256 + threadIdx.x;
  f32x2 v = {0.f, 0.f};
  if (e < total) v = reinterpret_cast<const f32x2*>(part)[e];
  float s1 = v[0], s2 = v[1];
#pragma unroll
  for (int o = 1; o < NP; o <<= 1) {
    s1 += __shfl_xor(s1, o);
    s2 += __shfl_xor(s2, o);
  }
  if ((threadIdx.x & (NP - 1)) == 0 && e < total) {
    const float mean = s1 * inv_d;
    const float var = fmaxf(s2 * inv_d - mean * mean, 0.f);
    const f32x2 o = {mean, rsqrtf(var + eps)};
    reinterpret_cast<f32x2*>(stats)[e / NP] = o;
  }
}
}  // namespace
}  // namespace hvlm

extern "C" int hvlm_ln_stats_finalize(const float* part, float* stats, int rows, int n_parts, int D, float eps, void* stream) {
  HVLM_REQUIRE(part && stats && rows > 0 && n_parts > 0 && D > 0, "ln_stats_finalize: bad args");
  HVLM_REQUIRE(((uintptr_t)part & 7) == 0 && ((uintptr_t)stats & 7) == 0, "ln_stats_finalize: 8-byte alignment");
  const long total = (long)rows * n_parts;
  const dim3 grid((unsigned)((total + 255) / 256));
#define HVLM_LNF(NP)                                                                                                              \
  hipLaunchKernelGGL(ln_stats_finalize_lanes_kernel<NP>, grid, dim3(256), 0, as_stream(stream), part, stats, total, 1.0f / (float)D, eps)
  if (n_parts == 4) HVLM_LNF(4);
  else if (n_parts == 8) HVLM_LNF(8);
  else if (n_parts == 16) HVLM_LNF(16);
  else if (n_parts == 32) HVLM_LNF(32);
  else
    hipLaunchKernelGGL(ln_stats_finalize_kernel, dim3(cdiv(rows, 256)), dim3(256), 0, as_stream(stream), part, stats, rows, n_parts,
                       1.0f / (float)D, eps);
#undef HVLM_LNF
  return check_launch("ln_stats_finalize");
}

extern "C" int hvlm_layernorm_fwd_fp8(const void* x, const void* w, const void* b, void* out8, float* inv_scale, float* next_inv,
                                      float bound_mul, float bound_add, int rows, int D, float eps, void* stream) {
  HVLM_REQUIRE(x && w && b && out8 && inv_scale, "layernorm_fwd_fp8: null pointer");
  HVLM_REQUIRE(rows > 0 && D > 0 && D % 8 == 0 && D <= 256 * 8 * MAXC, "layernorm_fwd_fp8: bad D=%d", D);
  if (D == 1024 || D == 2048) {
    if (D == 1024)
      hipLaunchKernelGGL(layernorm_fwd_q8_fast_kernel<2>, dim3(cdiv(rows, 4)), dim3(256), 0, as_stream(stream), (const bf16_t*)x,
                         (const bf16_t*)w, (const bf16_t*)b, (unsigned char*)out8, rows, eps, inv_scale, next_inv, bound_mul, bound_add);
    else
      hipLaunchKernelGGL(layernorm_fwd_q8_fast_kernel<4>, dim3(cdiv(rows, 4)), dim3(256), 0, as_stream(stream), (const bf16_t*)x,
                         (const bf16_t*)w, (const bf16_t*)b, (unsigned char*)out8, rows, eps, inv_scale, next_inv, bound_mul, bound_add);
    return check_launch("layernorm_fwd_fp8");
  }
  if (D <= 2048)
    hipLaunchKernelGGL((layernorm_fwd_kernel<64, false, true>), dim3(cdiv(rows, 4)), dim3(256), 0, as_stream(stream),
                       (const bf16_t*)x, (const bf16_t*)w, (const bf16_t*)b, (bf16_t*)out8, rows, D, eps, inv_scale, next_inv,
                       bound_mul, bound_add);
  else
    hipLaunchKernelGGL((layernorm_fwd_kernel<256, false, true>), dim3(rows), dim3(256), 0, as_stream(stream),
                       (const bf16_t*)x, (const bf16_t*)w, (const bf16_t*)b, (bf16_t*)out8, rows, D, eps, inv_scale, next_inv,
                       bound_mul, bound_add);
  return check_launch("layernorm_fwd_fp8");
}

extern "C" int hvlm_layernorm_fwd_f16(const void* x, const void* w, const void* b, void* y, int rows, int D, float eps,
                                      void* stream) {
  return layernorm_impl(x, w, b, y, rows, D, eps, true, stream);
}
