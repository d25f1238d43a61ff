// HBM-bound elementwise / data-movement kernels: SwiGLU, RoPE, add, casts, transpose, column sums.
// All bf16 traffic is 16 bytes per lane; grids are capped and grid-strided.
#include "common.h"

namespace hvlm {
namespace {

constexpr int EW_THREADS = 256;
static inline int ew_grid(long items) {
  long g = (items + EW_THREADS - 1) / EW_THREADS;
  return (int)(g < 1 ? 1 : (g > 256 * 16 ? 256 * 16 : g));
}

__device__ __forceinline__ float sigmoidf_(float x) { return 1.f / (1.f + __expf(-x)); }

// ---- SwiGLU: act = bf16(bf16(silu(g)) * u)  (transformers LlamaMLP, bf16 op-by-op) ------------
__global__ __launch_bounds__(EW_THREADS) void swiglu_fwd_kernel(const bf16_t* __restrict__ gu,
                                                                bf16_t* __restrict__ act, long M,
                                                                int F) {
  const int fv = F / 8;
  const long total = M * fv;
  for (long i = (long)blockIdx.x * EW_THREADS + threadIdx.x; i < total;
       i += (long)gridDim.x * EW_THREADS) {
    const long m = i / fv;
    const int c = (int)(i - m * fv) * 8;
    const bf16_t* row = gu + m * (2L * F);
    const u32x4 g = *reinterpret_cast<const u32x4*>(row + c);
    const u32x4 u = *reinterpret_cast<const u32x4*>(row + F + c);
    u32x4 o;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const float g0 = bf_lo(g[e]), g1 = bf_hi(g[e]);
      const float s0 = bf2f(f2bf(g0 * sigmoidf_(g0))), s1 = bf2f(f2bf(g1 * sigmoidf_(g1)));
      o[e] = pack_bf2(s0 * bf_lo(u[e]), s1 * bf_hi(u[e]));
    }
    *reinterpret_cast<u32x4*>(act + m * (long)F + c) = o;
  }
}

// SwiGLU whose output row is written as e4m3 with one scale per row — the input of the down-projection in the fp8 prefill
// (BASELINE config 5), bit-identical to swiglu_fwd_kernel followed by the rowwise quantise kernel.  One workgroup per row, the
// row (F <= 256 * 8 * NCH) in registers: gate and up are read once, one byte per element is written, and the separate pass
// (a read of the bf16 row and a second launch) disappears.
template <int NCH>
__global__ __launch_bounds__(256) void swiglu_fwd_q8_kernel(const bf16_t* __restrict__ gu, unsigned char* __restrict__ act8,
                                                            float* __restrict__ inv_scale, int F) {
  __shared__ float red[4];
  const int t = threadIdx.x;
  const size_t m = blockIdx.x;
  const bf16_t* row = gu + m * (2 * (size_t)F);
  u32x4 g[NCH], u[NCH];
#pragma unroll
  for (int c = 0; c < NCH; ++c) {
    const int i = min((c * 256 + t) * 8, F - 8);        // always in bounds; chunks past F are masked below
    g[c] = *reinterpret_cast<const u32x4*>(row + i);
    u[c] = *reinterpret_cast<const u32x4*>(row + F + i);
  }
  float v[NCH][8];
  float mx = 0.f;
#pragma unroll
  for (int c = 0; c < NCH; ++c) {
    const bool live = (c * 256 + t) * 8 < F;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const float g0 = bf_lo(g[c][e]), g1 = bf_hi(g[c][e]);
      const float s0 = bf2f(f2bf(g0 * sigmoidf_(g0))), s1 = bf2f(f2bf(g1 * sigmoidf_(g1)));
      const unsigned pk = pack_bf2(s0 * bf_lo(u[c][e]), s1 * bf_hi(u[c][e]));
      v[c][2 * e] = live ? bf_lo(pk) : 0.f; v[c][2 * e + 1] = live ? bf_hi(pk) : 0.f;
      mx = fmaxf(mx, fmaxf(fabsf(v[c][2 * e]), fabsf(v[c][2 * e + 1])));
    }
  }
  mx = block_max<4>(mx, red);
  const float amax = fmaxf(mx, 1e-12f);
  const float sc = 448.f / amax;
  if (t == 0) inv_scale[m] = amax / 448.f;
  unsigned char* orow = act8 + m * (size_t)F;
#pragma unroll
  for (int c = 0; c < NCH; ++c) {
    const int i = (c * 256 + t) * 8;
    if (i < F) {
      float f[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) f[e] = __builtin_amdgcn_fmed3f(v[c][e] * sc, -448.f, 448.f);
      unsigned lo = 0, hi = 0;
      lo = __builtin_amdgcn_cvt_pk_fp8_f32(f[0], f[1], lo, false);
      lo = __builtin_amdgcn_cvt_pk_fp8_f32(f[2], f[3], lo, true);
      hi = __builtin_amdgcn_cvt_pk_fp8_f32(f[4], f[5], hi, false);
      hi = __builtin_amdgcn_cvt_pk_fp8_f32(f[6], f[7], hi, true);
      const u32x2 o = {lo, hi};
      *reinterpret_cast<u32x2*>(orow + i) = o;
    }
  }
}

__global__ __launch_bounds__(EW_THREADS) void swiglu_bwd_kernel(const bf16_t* __restrict__ gu,
                                                                const bf16_t* __restrict__ dact,
                                                                bf16_t* __restrict__ dgu, long M,
                                                                int F) {
  const int fv = F / 8;
  const long total = M * fv;
  for (long i = (long)blockIdx.x * EW_THREADS + threadIdx.x; i < total;
       i += (long)gridDim.x * EW_THREADS) {
    const long m = i / fv;
    const int c = (int)(i - m * fv) * 8;
    const bf16_t* row = gu + m * (2L * F);
    const u32x4 g = *reinterpret_cast<const u32x4*>(row + c);
    const u32x4 u = *reinterpret_cast<const u32x4*>(row + F + c);
    const u32x4 d = *reinterpret_cast<const u32x4*>(dact + m * (long)F + c);
    u32x4 og, ou;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      float gg[2] = {bf_lo(g[e]), bf_hi(g[e])};
      float uu[2] = {bf_lo(u[e]), bf_hi(u[e])};
      float dd[2] = {bf_lo(d[e]), bf_hi(d[e])};
      float rg[2], ru[2];
#pragma unroll
      for (int k = 0; k < 2; ++k) {
        const float s = sigmoidf_(gg[k]);
        const float silu = gg[k] * s;
        ru[k] = dd[k] * silu;
        rg[k] = dd[k] * uu[k] * (s * (1.f + gg[k] * (1.f - s)));
      }
      og[e] = pack_bf2(rg[0], rg[1]);
      ou[e] = pack_bf2(ru[0], ru[1]);
    }
    bf16_t* orow = dgu + m * (2L * F);
    *reinterpret_cast<u32x4*>(orow + c) = og;
    *reinterpret_cast<u32x4*>(orow + F + c) = ou;
  }
}

// ---- RoPE (half-split): o1 = bf16(bf16(x1*c) + bf16(-x2*s)), o2 = bf16(bf16(x2*c) + bf16(x1*s))
// one work item = 8 rotation pairs of one (token, q|k, head).
__global__ __launch_bounds__(EW_THREADS) void rope_kernel(bf16_t* __restrict__ qkv,
                                                          const bf16_t* __restrict__ cos_tab,
                                                          const bf16_t* __restrict__ sin_tab,
                                                          const int32_t* __restrict__ pos, long M,
                                                          int H, int dh, int ld, int inverse) {
  const int half = dh / 2;
  const int pv = half / 8;               // items per head
  const long per_row = 2L * H * pv;      // q heads then k heads
  const long total = M * per_row;
  for (long i = (long)blockIdx.x * EW_THREADS + threadIdx.x; i < total;
       i += (long)gridDim.x * EW_THREADS) {
    const long m = i / per_row;
    const int r = (int)(i - m * per_row);
    const int hh = r / pv;               // 0..2H-1 (q heads, then k heads: contiguous in qkv)
    const int c = (r - hh * pv) * 8;
    bf16_t* base = qkv + m * (long)ld + (long)hh * dh + c;
    const int p = pos[m];
    const u32x4 cv = *reinterpret_cast<const u32x4*>(cos_tab + (long)p * half + c);
    const u32x4 sv = *reinterpret_cast<const u32x4*>(sin_tab + (long)p * half + c);
    const u32x4 x1 = *reinterpret_cast<const u32x4*>(base);
    const u32x4 x2 = *reinterpret_cast<const u32x4*>(base + half);
    u32x4 o1, o2;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      float a[2] = {bf_lo(x1[e]), bf_hi(x1[e])};
      float b[2] = {bf_lo(x2[e]), bf_hi(x2[e])};
      float cc[2] = {bf_lo(cv[e]), bf_hi(cv[e])};
      float ss[2] = {bf_lo(sv[e]), bf_hi(sv[e])};
      float r1[2], r2[2];
#pragma unroll
      for (int k = 0; k < 2; ++k) {
        const float s = inverse ? -ss[k] : ss[k];
        r1[k] = bf2f(f2bf(a[k] * cc[k])) + bf2f(f2bf(-b[k] * s));
        r2[k] = bf2f(f2bf(b[k] * cc[k])) + bf2f(f2bf(a[k] * s));
      }
      o1[e] = pack_bf2(r1[0], r1[1]);
      o2[e] = pack_bf2(r2[0], r2[1]);
    }
    *reinterpret_cast<u32x4*>(base) = o1;
    *reinterpret_cast<u32x4*>(base + half) = o2;
  }
}

// K | V columns of the prefill's fused qkv rows -> the per-sample cache slabs (lock-step generation): row (b, s) of
// qkv [B * S, 3D] goes to cache[b, s, D .. 3D) of a [B, max_len, 3D] cache; the q columns are never read from the cache and are
// not copied.  16 bytes per lane, a row's 2D columns contiguous on both sides.
__global__ __launch_bounds__(EW_THREADS) void kv_cache_fill_kernel(const bf16_t* __restrict__ qkv, bf16_t* __restrict__ cache,
                                                                   long rows, int S, long max_len, int D) {
  const int cpr = 2 * D / 8;                         // 16-byte chunks per row
  const long total = rows * cpr;
  for (long i = (long)blockIdx.x * EW_THREADS + threadIdx.x; i < total; i += (long)gridDim.x * EW_THREADS) {
    const long r = i / cpr;
    const int c = (int)(i - r * cpr) * 8;
    const long b = r / S, s_ = r - b * S;
    const u32x4 v = *reinterpret_cast<const u32x4*>(qkv + r * (3L * D) + D + c);
    *reinterpret_cast<u32x4*>(cache + (b * max_len + s_) * (3L * D) + D + c) = v;
  }
}

__global__ __launch_bounds__(EW_THREADS) void add_kernel(const bf16_t* __restrict__ a,
                                                         const bf16_t* __restrict__ b,
                                                         bf16_t* __restrict__ y, long n8) {
  for (long i = (long)blockIdx.x * EW_THREADS + threadIdx.x; i < n8;
       i += (long)gridDim.x * EW_THREADS) {
    const u32x4 x = reinterpret_cast<const u32x4*>(a)[i];
    const u32x4 z = reinterpret_cast<const u32x4*>(b)[i];
    u32x4 o;
#pragma unroll
    for (int e = 0; e < 4; ++e) o[e] = pack_bf2(bf_lo(x[e]) + bf_lo(z[e]), bf_hi(x[e]) + bf_hi(z[e]));
    reinterpret_cast<u32x4*>(y)[i] = o;
  }
}
__global__ void add_tail_kernel(const bf16_t* a, const bf16_t* b, bf16_t* y, long start, long n) {
  const long i = start + threadIdx.x;
  if (i < n) y[i] = f2bf(bf2f(a[i]) + bf2f(b[i]));
}

__global__ __launch_bounds__(EW_THREADS) void cast_f2b_kernel(const float* __restrict__ in,
                                                              bf16_t* __restrict__ out, long n) {
  for (long i = ((long)blockIdx.x * EW_THREADS + threadIdx.x) * 4; i < n;
       i += (long)gridDim.x * EW_THREADS * 4) {
    if (i + 3 < n) {
      const f32x4 v = *reinterpret_cast<const f32x4*>(in + i);
      u32x2 o = {pack_bf2(v[0], v[1]), pack_bf2(v[2], v[3])};
      *reinterpret_cast<u32x2*>(out + i) = o;
    } else {
      for (long k = i; k < n; ++k) out[k] = f2bf(in[k]);
    }
  }
}
__global__ __launch_bounds__(EW_THREADS) void cast_b2f_kernel(const bf16_t* __restrict__ in,
                                                              float* __restrict__ out, long n) {
  for (long i = ((long)blockIdx.x * EW_THREADS + threadIdx.x) * 4; i < n;
       i += (long)gridDim.x * EW_THREADS * 4) {
    if (i + 3 < n) {
      const u32x2 v = *reinterpret_cast<const u32x2*>(in + i);
      f32x4 o = {bf_lo(v[0]), bf_hi(v[0]), bf_lo(v[1]), bf_hi(v[1])};
      *reinterpret_cast<f32x4*>(out + i) = o;
    } else {
      for (long k = i; k < n; ++k) out[k] = bf2f(in[k]);
    }
  }
}

// 16-bit <-> 16-bit: bf16 -> fp16 (TO_F16) or fp16 -> bf16, 8 elements per thread, n % 8 == 0.  The reference casts the
// frames to the tower's dtype (clip_encoder.py:44,48: images.to(dtype=self.dtype)) and the features back (.to(images.dtype)).
template <bool TO_F16>
__global__ __launch_bounds__(EW_THREADS) void cast16_kernel(const bf16_t* __restrict__ in, bf16_t* __restrict__ out, long n8) {
  for (long i = (long)blockIdx.x * EW_THREADS + threadIdx.x; i < n8; i += (long)gridDim.x * EW_THREADS) {
    const u32x4 v = reinterpret_cast<const u32x4*>(in)[i];
    u32x4 o;
#pragma unroll
    for (int e = 0; e < 4; ++e)
      o[e] = TO_F16 ? pack_h2(bf_lo(v[e]), bf_hi(v[e])) : pack_bf2(h_lo(v[e]), h_hi(v[e]));
    reinterpret_cast<u32x4*>(out)[i] = o;
  }
}

// ---- transpose: out[C, R_pad] = in[R, C]^T, 64x64 tiles through LDS ----------------------------
// LDS tile: 64 rows x 128 B, the 16-B chunk c of row r stored at chunk position c ^ ((r>>3)&7): one conflict-free
// ds_write_b128 per loaded chunk; the gather reads one dword (TWO adjacent columns) from 8 consecutive rows and splits
// it with v_perm_b32 into two output rows of 8 elements (16 B each) — 10 LDS instructions per 32 output bytes (the
// first version read 2-byte elements one by one: 24, and ran at 3 TB/s).
__global__ __launch_bounds__(256) void transpose_kernel(const bf16_t* __restrict__ in,
                                                        bf16_t* __restrict__ out, int R, int C,
                                                        int ld_in, int R_pad) {
  __shared__ __attribute__((aligned(16))) char tile[64 * 128];
  const int r0 = blockIdx.y * 64, c0 = blockIdx.x * 64;
  const int t = threadIdx.x;
  // load: 64 rows x 8 chunks of 8 elements = 512 chunks, 2 per thread
#pragma unroll
  for (int k = 0; k < 2; ++k) {
    const int idx = t + k * 256;
    const int r = idx >> 3, c = (idx & 7) * 8;
    u32x4 v = {0u, 0u, 0u, 0u};
    if (r0 + r < R) {
      if (c0 + c + 7 < C) {
        v = *reinterpret_cast<const u32x4*>(in + (size_t)(r0 + r) * ld_in + c0 + c);
      } else {
        bf16_t tmp[8];
        for (int e = 0; e < 8; ++e) tmp[e] = (c0 + c + e < C) ? in[(size_t)(r0 + r) * ld_in + c0 + c + e] : (bf16_t)0;
        v[0] = tmp[0] | ((unsigned)tmp[1] << 16); v[1] = tmp[2] | ((unsigned)tmp[3] << 16);
        v[2] = tmp[4] | ((unsigned)tmp[5] << 16); v[3] = tmp[6] | ((unsigned)tmp[7] << 16);
      }
    }
    *reinterpret_cast<u32x4*>(tile + r * 128 + (((idx & 7) ^ ((r >> 3) & 7)) << 4)) = v;
  }
  __syncthreads();
  const int c2 = t >> 3, rb = t & 7;           // column pair (2*c2, 2*c2+1), input rows rb*8 .. rb*8+7
  const int r = rb * 8;
  if (r0 + r >= R_pad) return;
  unsigned w[8];
#pragma unroll
  for (int j = 0; j < 8; ++j)
    w[j] = *reinterpret_cast<const unsigned*>(tile + (r + j) * 128 + ((((c2 >> 2) ^ rb)) << 4) + (c2 & 3) * 4);
#pragma unroll
  for (int h = 0; h < 2; ++h) {                // h = 0: column 2*c2 (low halves), 1: column 2*c2+1 (high halves)
    const int c = 2 * c2 + h;
    if (c0 + c >= C) continue;
    const unsigned sel = h ? 0x07060302u : 0x05040100u;
    const u32x4 o = {__builtin_amdgcn_perm(w[1], w[0], sel), __builtin_amdgcn_perm(w[3], w[2], sel),
                     __builtin_amdgcn_perm(w[5], w[4], sel), __builtin_amdgcn_perm(w[7], w[6], sel)};
    bf16_t* op = out + (size_t)(c0 + c) * R_pad + r0 + r;
    if (r0 + r + 7 < R_pad) {
      *reinterpret_cast<u32x4*>(op) = o;
    } else {
      for (int j = 0; j < 8; ++j)
        if (r0 + r + j < R_pad) op[j] = (bf16_t)((o[j >> 1] >> ((j & 1) * 16)) & 0xffffu);
    }
  }
}

// ---- SwiGLU with a fused TRANSPOSED second output (64x64 tiles through LDS, same tile image and gather as
// transpose_kernel): the weight-gradient GEMMs need their operands token-contiguous ([features, tokens_pad]); writing that
// copy here saves the read half of a separate transpose pass.
//   forward : act[m, f] = swiglu(gu[m, f], gu[m, F + f]) row-major (feeds down_proj) AND actT[f, m] (saved for dW_down)
//   backward: dgu[m, c] (c < F: d gate, c >= F: d up) row-major (feeds the dX GEMM) AND dguT[c, m] (feeds dW_gate|up)
// Rows m >= M of the transposed output (columns M .. R_pad-1) are written as zeros (K padding of the GEMM).
template <bool BWD>
__global__ __launch_bounds__(256) void swiglu_t_kernel(const bf16_t* __restrict__ gu, const bf16_t* __restrict__ dact,
                                                       bf16_t* __restrict__ out, bf16_t* __restrict__ outT, int M, int F,
                                                       int R_pad) {
  // backward: ONE workgroup produces the gate AND the up gradient of its 64 x 64 (token, feature) tile — g, u, d are read
  // and the sigmoid evaluated once (the first version ran one workgroup per output column block and read them twice)
  constexpr int NT = BWD ? 2 : 1;
  __shared__ __attribute__((aligned(16))) char tile[NT][64 * 128];
  const int C = BWD ? 2 * F : F;                  // columns of the row-major output
  const int r0 = blockIdx.y * 64, c0 = blockIdx.x * 64;
  const int t = threadIdx.x;
#pragma unroll
  for (int k = 0; k < 2; ++k) {
    const int idx = t + k * 256;
    const int r = idx >> 3, f = c0 + (idx & 7) * 8;
    u32x4 o[NT];
#pragma unroll
    for (int n = 0; n < NT; ++n) o[n] = u32x4{0u, 0u, 0u, 0u};
    if (r0 + r < M) {
      const bf16_t* row = gu + (size_t)(r0 + r) * (2 * (size_t)F);
      const u32x4 g = *reinterpret_cast<const u32x4*>(row + f);
      const u32x4 u = *reinterpret_cast<const u32x4*>(row + F + f);
      if constexpr (!BWD) {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const float g0 = bf_lo(g[e]), g1 = bf_hi(g[e]);
          const float s0 = bf2f(f2bf(g0 * sigmoidf_(g0))), s1 = bf2f(f2bf(g1 * sigmoidf_(g1)));
          o[0][e] = pack_bf2(s0 * bf_lo(u[e]), s1 * bf_hi(u[e]));
        }
        *reinterpret_cast<u32x4*>(out + (size_t)(r0 + r) * C + f) = o[0];
      } else {
        const u32x4 d = *reinterpret_cast<const u32x4*>(dact + (size_t)(r0 + r) * F + f);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          float gg[2] = {bf_lo(g[e]), bf_hi(g[e])};
          float uu[2] = {bf_lo(u[e]), bf_hi(u[e])};
          float dd[2] = {bf_lo(d[e]), bf_hi(d[e])};
          float rg[2], ru[2];
#pragma unroll
          for (int q = 0; q < 2; ++q) {
            const float sg = sigmoidf_(gg[q]);
            rg[q] = dd[q] * uu[q] * (sg * (1.f + gg[q] * (1.f - sg)));
            ru[q] = dd[q] * (gg[q] * sg);
          }
          o[0][e] = pack_bf2(rg[0], rg[1]);
          o[NT - 1][e] = pack_bf2(ru[0], ru[1]);
        }
        *reinterpret_cast<u32x4*>(out + (size_t)(r0 + r) * C + f) = o[0];
        *reinterpret_cast<u32x4*>(out + (size_t)(r0 + r) * C + F + f) = o[NT - 1];
      }
    }
#pragma unroll
    for (int n = 0; n < NT; ++n)
      *reinterpret_cast<u32x4*>(tile[n] + r * 128 + (((idx & 7) ^ ((r >> 3) & 7)) << 4)) = o[n];
  }
  __syncthreads();
  const int c2 = t >> 3, rb = t & 7;           // column pair (2*c2, 2*c2+1), input rows rb*8 .. rb*8+7
  const int r = rb * 8;
  if (r0 + r >= R_pad) return;
#pragma unroll
  for (int n = 0; n < NT; ++n) {
    unsigned w[8];
#pragma unroll
    for (int j = 0; j < 8; ++j)
      w[j] = *reinterpret_cast<const unsigned*>(tile[n] + (r + j) * 128 + ((((c2 >> 2) ^ rb)) << 4) + (c2 & 3) * 4);
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const int c = 2 * c2 + h;
      const unsigned sel = h ? 0x07060302u : 0x05040100u;
      const u32x4 ov = {__builtin_amdgcn_perm(w[1], w[0], sel), __builtin_amdgcn_perm(w[3], w[2], sel),
                        __builtin_amdgcn_perm(w[5], w[4], sel), __builtin_amdgcn_perm(w[7], w[6], sel)};
      *reinterpret_cast<u32x4*>(outT + ((size_t)n * F + c0 + c) * R_pad + r0 + r) = ov;
    }
  }
}

// ---- column sums: out[c] = sum_r in[r][c].  One workgroup per 64 columns: 32 row groups x 8 lanes x 8 columns (16-byte
// loads), the row groups combined through LDS in a fixed order -> deterministic (no atomics, no pre-zeroed output). ----
__global__ __launch_bounds__(256) void colsum_kernel(const bf16_t* __restrict__ in,
                                                     float* __restrict__ out, int R, int C,
                                                     int ld_in) {
  __shared__ float part[32][65];
  const int lane8 = threadIdx.x & 7, rg = threadIdx.x >> 3;
  const int c = blockIdx.x * 64 + lane8 * 8;
  float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  if (c + 7 < C) {
    for (int r = rg; r < R; r += 32) {
      const u32x4 v = *reinterpret_cast<const u32x4*>(in + (size_t)r * ld_in + c);
#pragma unroll
      for (int e = 0; e < 4; ++e) { acc[2 * e] += bf_lo(v[e]); acc[2 * e + 1] += bf_hi(v[e]); }
    }
  } else {
    for (int r = rg; r < R; r += 32)
      for (int e = 0; e < 8; ++e)
        if (c + e < C) acc[e] += bf2f(in[(size_t)r * ld_in + c + e]);
  }
#pragma unroll
  for (int e = 0; e < 8; ++e) part[rg][lane8 * 8 + e] = acc[e];
  __syncthreads();
  if (threadIdx.x < 64 && blockIdx.x * 64 + threadIdx.x < C) {
    float s = 0.f;
#pragma unroll
    for (int g = 0; g < 32; ++g) s += part[g][threadIdx.x];
    out[blockIdx.x * 64 + threadIdx.x] = s;
  }
}

}  // namespace
}  // namespace hvlm

using namespace hvlm;

extern "C" int hvlm_swiglu_fwd(const void* gu, void* act, int M, int F, void* stream) {
  HVLM_REQUIRE(gu && act && M > 0 && F > 0 && F % 8 == 0, "swiglu_fwd: bad args (F %% 8)");
  hipLaunchKernelGGL(swiglu_fwd_kernel, dim3(ew_grid((long)M * F / 8)), dim3(EW_THREADS), 0,
                     as_stream(stream), (const bf16_t*)gu, (bf16_t*)act, (long)M, F);
  return check_launch("swiglu_fwd");
}

extern "C" int hvlm_swiglu_fwd_fp8(const void* gu, void* act8, float* inv_scale, int M, int F, void* stream) {
  HVLM_REQUIRE(gu && act8 && inv_scale && M > 0 && F >= 8 && F % 8 == 0 && F <= 256 * 8 * 7,
               "swiglu_fwd_fp8: bad args (F %% 8, F <= 14336)");
  if (F <= 256 * 8 * 6)
    hipLaunchKernelGGL(swiglu_fwd_q8_kernel<6>, dim3(M), dim3(256), 0, as_stream(stream), (const bf16_t*)gu, (unsigned char*)act8,
                       inv_scale, F);
  else
    hipLaunchKernelGGL(swiglu_fwd_q8_kernel<7>, dim3(M), dim3(256), 0, as_stream(stream), (const bf16_t*)gu, (unsigned char*)act8,
                       inv_scale, F);
  return check_launch("swiglu_fwd_fp8");
}

extern "C" int hvlm_swiglu_bwd(const void* gu, const void* dact, void* dgu, int M, int F, void* stream) {
  HVLM_REQUIRE(gu && dact && dgu && M > 0 && F > 0 && F % 8 == 0, "swiglu_bwd: bad args (F %% 8)");
  hipLaunchKernelGGL(swiglu_bwd_kernel, dim3(ew_grid((long)M * F / 8)), dim3(EW_THREADS), 0,
                     as_stream(stream), (const bf16_t*)gu, (const bf16_t*)dact, (bf16_t*)dgu, (long)M, F);
  return check_launch("swiglu_bwd");
}

extern "C" int hvlm_swiglu_fwd_t(const void* gu, void* act, void* actT, int M, int F, int R_pad, void* stream) {
  HVLM_REQUIRE(gu && act && actT && M > 0 && F > 0 && F % 64 == 0 && R_pad >= M && R_pad % 64 == 0,
               "swiglu_fwd_t: bad args (F %% 64, R_pad %% 64, R_pad >= M)");
  hipLaunchKernelGGL(swiglu_t_kernel<false>, dim3(F / 64, R_pad / 64), dim3(256), 0, as_stream(stream), (const bf16_t*)gu,
                     (const bf16_t*)nullptr, (bf16_t*)act, (bf16_t*)actT, M, F, R_pad);
  return check_launch("swiglu_fwd_t");
}

extern "C" int hvlm_swiglu_bwd_t(const void* gu, const void* dact, void* dgu, void* dguT, int M, int F, int R_pad,
                                 void* stream) {
  HVLM_REQUIRE(gu && dact && dgu && dguT && M > 0 && F > 0 && F % 64 == 0 && R_pad >= M && R_pad % 64 == 0,
               "swiglu_bwd_t: bad args (F %% 64, R_pad %% 64, R_pad >= M)");
  hipLaunchKernelGGL(swiglu_t_kernel<true>, dim3(F / 64, R_pad / 64), dim3(256), 0, as_stream(stream), (const bf16_t*)gu,
                     (const bf16_t*)dact, (bf16_t*)dgu, (bf16_t*)dguT, M, F, R_pad);
  return check_launch("swiglu_bwd_t");
}

extern "C" int hvlm_rope_qk(void* qkv, const void* cos_tab, const void* sin_tab, const int32_t* pos,
                            int M, int H, int dh, int ld, int inverse, void* stream) {
  HVLM_REQUIRE(qkv && cos_tab && sin_tab && pos, "rope: null pointer");
  HVLM_REQUIRE(M > 0 && H > 0 && dh % 16 == 0 && ld % 8 == 0 && ld >= 2 * H * dh, "rope: bad shape");
  const long items = (long)M * 2 * H * (dh / 16);
  hipLaunchKernelGGL(rope_kernel, dim3(ew_grid(items)), dim3(EW_THREADS), 0, as_stream(stream),
                     (bf16_t*)qkv, (const bf16_t*)cos_tab, (const bf16_t*)sin_tab, pos, (long)M, H, dh,
                     ld, inverse);
  return check_launch("rope_qk");
}

extern "C" int hvlm_kv_cache_fill(const void* qkv, void* cache, int B, int S, int max_len, int D, void* stream) {
  HVLM_REQUIRE(qkv && cache && B > 0 && S > 0 && max_len >= S && D > 0 && D % 8 == 0, "kv_cache_fill: bad args (D %% 8, max_len >= S)");
  HVLM_REQUIRE(((uintptr_t)qkv & 15) == 0 && ((uintptr_t)cache & 15) == 0, "kv_cache_fill: alignment");
  const long rows = (long)B * S;
  hipLaunchKernelGGL(kv_cache_fill_kernel, dim3(ew_grid(rows * (2 * D / 8))), dim3(EW_THREADS), 0, as_stream(stream),
                     (const bf16_t*)qkv, (bf16_t*)cache, rows, S, (long)max_len, D);
  return check_launch("kv_cache_fill");
}

extern "C" int hvlm_add_bf16(const void* a, const void* b, void* y, long n, void* stream) {
  HVLM_REQUIRE(a && b && y && n > 0, "add: bad args");
  const long n8 = n / 8;
  if (n8 > 0)
    hipLaunchKernelGGL(add_kernel, dim3(ew_grid(n8)), dim3(EW_THREADS), 0, as_stream(stream),
                       (const bf16_t*)a, (const bf16_t*)b, (bf16_t*)y, n8);
  if (n8 * 8 < n)
    hipLaunchKernelGGL(add_tail_kernel, dim3(1), dim3(8), 0, as_stream(stream), (const bf16_t*)a,
                       (const bf16_t*)b, (bf16_t*)y, n8 * 8, n);
  return check_launch("add_bf16");
}

extern "C" int hvlm_cast_f32_to_bf16(const float* in, void* out, long n, void* stream) {
  HVLM_REQUIRE(in && out && n > 0, "cast: bad args");
  hipLaunchKernelGGL(cast_f2b_kernel, dim3(ew_grid(n / 4 + 1)), dim3(EW_THREADS), 0, as_stream(stream),
                     in, (bf16_t*)out, n);
  return check_launch("cast_f32_to_bf16");
}

extern "C" int hvlm_cast_bf16_to_f32(const void* in, float* out, long n, void* stream) {
  HVLM_REQUIRE(in && out && n > 0, "cast: bad args");
  hipLaunchKernelGGL(cast_b2f_kernel, dim3(ew_grid(n / 4 + 1)), dim3(EW_THREADS), 0, as_stream(stream),
                     (const bf16_t*)in, out, n);
  return check_launch("cast_bf16_to_f32");
}

extern "C" int hvlm_cast_bf16_to_f16(const void* in, void* out, long n, void* stream) {
  HVLM_REQUIRE(in && out && n > 0 && n % 8 == 0, "cast_bf16_to_f16: n must be a positive multiple of 8");
  hipLaunchKernelGGL(cast16_kernel<true>, dim3(ew_grid(n / 8)), dim3(EW_THREADS), 0, as_stream(stream), (const bf16_t*)in,
                     (bf16_t*)out, n / 8);
  return check_launch("cast_bf16_to_f16");
}

extern "C" int hvlm_cast_f16_to_bf16(const void* in, void* out, long n, void* stream) {
  HVLM_REQUIRE(in && out && n > 0 && n % 8 == 0, "cast_f16_to_bf16: n must be a positive multiple of 8");
  hipLaunchKernelGGL(cast16_kernel<false>, dim3(ew_grid(n / 8)), dim3(EW_THREADS), 0, as_stream(stream), (const bf16_t*)in,
                     (bf16_t*)out, n / 8);
  return check_launch("cast_f16_to_bf16");
}

extern "C" int hvlm_transpose_bf16(const void* in, void* out, int R, int C, int ld_in, int R_pad,
                                   void* stream) {
  HVLM_REQUIRE(in && out && R > 0 && C > 0 && ld_in >= C && R_pad >= R, "transpose: bad args");
  HVLM_REQUIRE(ld_in % 8 == 0 && R_pad % 8 == 0, "transpose: ld_in, R_pad must be multiples of 8");
  dim3 grid(cdiv(C, 64), cdiv(R_pad, 64));
  hipLaunchKernelGGL(transpose_kernel, grid, dim3(256), 0, as_stream(stream), (const bf16_t*)in,
                     (bf16_t*)out, R, C, ld_in, R_pad);
  return check_launch("transpose_bf16");
}

extern "C" int hvlm_colsum_bf16(const void* in, float* out, int R, int C, int ld_in, void* stream) {
  HVLM_REQUIRE(in && out && R > 0 && C > 0 && ld_in >= C, "colsum: bad args");
  HVLM_REQUIRE(ld_in % 8 == 0, "colsum: ld_in %% 8");
  hipLaunchKernelGGL(colsum_kernel, dim3(cdiv(C, 64)), dim3(256), 0, as_stream(stream), (const bf16_t*)in, out, R, C, ld_in);
  return check_launch("colsum_bf16");
}
