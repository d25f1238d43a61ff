// Flash-style attention for gfx950 (MFMA 32x32x16 bf16, online softmax in fp32), forward and
// backward, reading q/k/v straight from the fused, token-major qkv buffer [B*S, 3*H*dh] and
// writing O / dqkv in the layouts the neighbouring GEMMs consume (no [B,H,S,dh] permutes).
//
// Design ("every lane owns one query / one key"):
//   forward  : S^T = K Q^T (MFMA A = K tile from LDS, B = Q fragments in registers) puts one QUERY
//              per lane (column l&31) with 16 keys per accumulator, so row max / row sum / the
//              O rescale are per-lane scalars (one cross-lane op with lane^32).  Then
//              O^T = V^T P^T (A = V^T tile from LDS, B = P straight from the S^T accumulator
//              registers, converted to bf16 — the k-slot order of an MFMA is free as long as A
//              and B agree, so no shuffle is needed).  O^T again has one query per lane and four
//              consecutive head-dim elements per register quad -> 8-byte row stores.
//   backward : two kernels. dQ (one query per lane):  S^T = K Q^T, dP^T = V dO^T,
//              dS^T = P^T o (dP^T - delta), dQ^T = K^T dS^T.   dK/dV (one key per lane):
//              S = Q K^T, dP = dO V^T, dV^T = dO^T P, dK^T = Q^T dS.  No atomics, deterministic.
//   The k-contiguous "transposed" operand tiles (V^T, K^T, Q^T, dO^T: [dh][tokens]) come from a
//   cheap HBM-bound head-transpose pre-pass into a caller-provided workspace.
//   K/V/Q/dO tiles go HBM -> LDS by LDS-DMA (global_load_lds_dwordx4) with the XOR swizzle
//   applied to the per-lane source address (conflict-free ds_read_b128 fragment reads).
//
// Reference arithmetic replaced: LlamaAttention softmax(QK^T/sqrt(dh) + causal mask) V
// (transformers modeling_llama; flash path llava/train/llama_flash_attn_monkey_patch.py:81-94) and
// CLIPAttention (non-causal, dh = 64).
#include "common.h"
#include <type_traits>

namespace hvlm {
namespace {

typedef __attribute__((address_space(3))) void* lds_ptr_t;
typedef const __attribute__((address_space(1))) void* gbl_ptr_t;

__device__ __forceinline__ void glds16(const bf16_t* g, unsigned lds_off, char* smem) {
  __builtin_amdgcn_global_load_lds((gbl_ptr_t)g, (lds_ptr_t)(smem + lds_off), 16, 0, 0);
}

// byte offset of 16-byte chunk `c` (logical) of row `r` inside a swizzled [rows][COLS] bf16 tile
template <int COLS>
__device__ __forceinline__ unsigned tile_addr(int r, int c) {
  if constexpr (COLS == 128) return (unsigned)r * 256u + (unsigned)((c ^ (r & 15)) << 4);
  else return (unsigned)r * 128u + (unsigned)((c ^ ((r >> 1) & 7)) << 4);
}
template <int COLS>
__device__ __forceinline__ int swz_bits(int r) {
  if constexpr (COLS == 128) return r & 15;
  else return (r >> 1) & 7;
}

// LDS-DMA a [ROWS][COLS] tile (4 waves).  Row r of the tile is global row min(row0 + r, row_lim-1)
// of a matrix with `stride` elements between rows, starting at `gbase`.
template <int ROWS, int COLS, int NW = 4>
__device__ __forceinline__ void load_tile(char* smem, unsigned tile_off, const bf16_t* gbase,
                                          long stride, int row0, int row_lim, int w, int l) {
  constexpr int CPR = COLS / 8;        // 16-byte chunks per row
  constexpr int RPI = 64 / CPR;        // rows per wave-instruction (1 KiB)
  constexpr int NPIECE = ROWS / RPI;
  constexpr int PPW = (NPIECE + NW - 1) / NW;   // pieces per wave (round-robin over the NW waves)
#pragma unroll
  for (int i = 0; i < PPW; ++i) {
    const int pi = i * NW + w;
    if (NPIECE % NW != 0 && pi >= NPIECE) break;
    const int r = pi * RPI + l / CPR;
    const int cp = l % CPR;
    const int c = cp ^ swz_bits<COLS>(r);
    const int gr = min(row0 + r, row_lim - 1);
    glds16(gbase + (long)gr * stride + c * 8, tile_off + (unsigned)pi * 1024u, smem);
  }
}

__device__ __forceinline__ bf16x8 lds_frag128(const char* smem, unsigned off) {
  return *reinterpret_cast<const bf16x8*>(smem + off);
}
// fragment made of two 8-byte runs (k-slots 0..3 from `off0`, 4..7 from `off1`)
__device__ __forceinline__ bf16x8 lds_frag64x2(const char* smem, unsigned off0, unsigned off1) {
  const u32x2 a = *reinterpret_cast<const u32x2*>(smem + off0);
  const u32x2 b = *reinterpret_cast<const u32x2*>(smem + off1);
  u32x4 v = {a[0], a[1], b[0], b[1]};
  return __builtin_bit_cast(bf16x8, v);
}
__device__ __forceinline__ bf16x8 pack8(const float* p) {
  u32x4 v = {pack_bf2(p[0], p[1]), pack_bf2(p[2], p[3]), pack_bf2(p[4], p[5]), pack_bf2(p[6], p[7])};
  return __builtin_bit_cast(bf16x8, v);
}
__device__ __forceinline__ int crow(int r, int hi) { return (r & 3) + 8 * (r >> 2) + 4 * hi; }

// raw v_exp_f32 (arguments here are <= ~0 or -inf; the libm exp2f wrapper adds a denormal-range rescale:
// v_cmp + v_cndmask + v_ldexp per element) and the lane^32 exchange without an LDS round trip.
__device__ __forceinline__ float fast_exp2(float x) { return __builtin_amdgcn_exp2f(x); }
__device__ __forceinline__ float xlane32(float v) {
  const unsigned u = __float_as_uint(v);
  const auto r = __builtin_amdgcn_permlane32_swap(u, u, false, false);
  return __uint_as_float((threadIdx.x & 32) ? r[0] : r[1]);
}

constexpr float LOG2E = 1.4426950408889634f;
constexpr float LN2 = 0.6931471805599453f;

// =================================================================================================
// forward
// =================================================================================================
template <int DH, bool CAUSAL, int NW>
__global__ __launch_bounds__(NW * 64, 2) void attn_fwd_kernel(const bf16_t* __restrict__ qkv,
                                                       const bf16_t* __restrict__ vt,
                                                       bf16_t* __restrict__ out,
                                                       float* __restrict__ lse, int S, int H, int ld,
                                                       int S_pad, float scale_log2) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr unsigned KT = 64 * DH * 2;   // K tile bytes ; V^T tile has the same size
  constexpr unsigned STAGE = 2 * KT;
  const int tid = threadIdx.x, l = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l31 = l & 31, hi = l >> 5;
  constexpr int QB = NW * 32;            // query rows per workgroup
  const int nqt = (S + QB - 1) / QB;
  const int qt = CAUSAL ? (nqt - 1 - (int)blockIdx.x) : (int)blockIdx.x;  // heavy tiles first
  const int h = blockIdx.y, b = blockIdx.z;
  const int q0 = qt * QB;
  const int qi = q0 + w * 32 + l31;
  const long tok0 = (long)b * S;

  bf16x8 qf[DH / 16];
  {
    const bf16_t* qp = qkv + (tok0 + min(qi, S - 1)) * ld + h * DH + hi * 8;
#pragma unroll
    for (int ks = 0; ks < DH / 16; ++ks) qf[ks] = *reinterpret_cast<const bf16x8*>(qp + ks * 16);
  }
  f32x16 o[DH / 32];
#pragma unroll
  for (int db = 0; db < DH / 32; ++db)
#pragma unroll
    for (int r = 0; r < 16; ++r) o[db][r] = 0.f;
  float m = -INFINITY, lsum = 0.f;

  const int nkt_all = (S + 63) / 64;
  const int nkt = CAUSAL ? min(nkt_all, min(q0 + QB - 1, S - 1) / 64 + 1) : nkt_all;
  const bf16_t* kbase = qkv + tok0 * ld + (long)H * DH + h * DH;
  const bf16_t* vtbase = vt + ((long)b * H + h) * DH * (long)S_pad;

  auto stage = [&](int buf, int kt) {
    const unsigned base = (unsigned)buf * STAGE;
    load_tile<64, DH, NW>(smem, base, kbase, ld, kt * 64, S, w, l);
    load_tile<DH, 64, NW>(smem, base + KT, vtbase + kt * 64, S_pad, 0, DH, w, l);
  };

  stage(0, 0);
  for (int kt = 0; kt < nkt; ++kt) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (kt + 1 < nkt) stage((kt + 1) & 1, kt + 1);
    const int kv0 = kt * 64;
    if (CAUSAL && kv0 > q0 + w * 32 + 31) continue;   // wave-uniform: tile fully above the diagonal
    const unsigned sK = (unsigned)(kt & 1) * STAGE, sV = sK + KT;

    // second 32-key block of the tile entirely past the sequence end (or above the diagonal): skip it
    const bool kb1 = (kv0 + 32 < S) && !(CAUSAL && kv0 + 32 > q0 + w * 32 + 31);
    auto tile_body = [&](auto nkb_tag) {
      constexpr int NKB = decltype(nkb_tag)::value;
      f32x16 s[NKB];
#pragma unroll
      for (int kb = 0; kb < NKB; ++kb)
#pragma unroll
        for (int r = 0; r < 16; ++r) s[kb][r] = 0.f;
#pragma unroll
      for (int ks = 0; ks < DH / 16; ++ks)
#pragma unroll
        for (int kb = 0; kb < NKB; ++kb) {
          const bf16x8 kf = lds_frag128(smem, sK + tile_addr<DH>(kb * 32 + l31, ks * 2 + hi));
          s[kb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf, qf[ks], s[kb], 0, 0, 0);
        }
      const bool need_mask = (kv0 + 32 * NKB > S) || (CAUSAL && kv0 + 32 * NKB - 1 > q0 + w * 32);
      float mx = -INFINITY;
#pragma unroll
      for (int kb = 0; kb < NKB; ++kb)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          float t = s[kb][r] * scale_log2;
          if (need_mask) {
            const int key = kv0 + kb * 32 + crow(r, hi);
            if (key >= S || (CAUSAL && key > qi)) t = -INFINITY;
          }
          s[kb][r] = t;
          mx = fmaxf(mx, t);
        }
      mx = fmaxf(mx, xlane32(mx));
      const float mnew = fmaxf(m, mx);
      const float alpha = fast_exp2(m - mnew);
      m = mnew;
      float psum = 0.f;
#pragma unroll
      for (int kb = 0; kb < NKB; ++kb)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const float p = fast_exp2(s[kb][r] - mnew);
          s[kb][r] = p;
          psum += p;
        }
      lsum = lsum * alpha + psum;
#pragma unroll
      for (int db = 0; db < DH / 32; ++db)
#pragma unroll
        for (int r = 0; r < 16; ++r) o[db][r] *= alpha;
#pragma unroll
      for (int kb = 0; kb < NKB; ++kb)
#pragma unroll
        for (int sp = 0; sp < 2; ++sp) {
          float pv[8];
#pragma unroll
          for (int j = 0; j < 8; ++j) pv[j] = s[kb][8 * sp + j];
          const bf16x8 pf = pack8(pv);
          const int c1 = kb * 4 + 2 * sp;
#pragma unroll
          for (int db = 0; db < DH / 32; ++db) {
            const int r = db * 32 + l31;
            const bf16x8 vf = lds_frag64x2(smem, sV + tile_addr<64>(r, c1) + 8u * hi,
                                           sV + tile_addr<64>(r, c1 + 1) + 8u * hi);
            o[db] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf, pf, o[db], 0, 0, 0);
          }
        }
    };
    if (kb1) tile_body(std::integral_constant<int, 2>{});
    else tile_body(std::integral_constant<int, 1>{});
  }
  const float ltot = lsum + xlane32(lsum);
  if (qi < S) {
    const float inv = 1.f / ltot;
    bf16_t* op = out + (tok0 + qi) * (long)(H * DH) + h * DH;
#pragma unroll
    for (int db = 0; db < DH / 32; ++db)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int d = db * 32 + 8 * g + 4 * hi;
        u32x2 v = {pack_bf2(o[db][4 * g] * inv, o[db][4 * g + 1] * inv),
                   pack_bf2(o[db][4 * g + 2] * inv, o[db][4 * g + 3] * inv)};
        *reinterpret_cast<u32x2*>(op + d) = v;
      }
    if (lse && hi == 0) lse[((long)b * H + h) * S + qi] = m * LN2 + logf(ltot);
  }
}

// =================================================================================================
// head transpose: src[(b*S+s)*ld + col0 + h*DH + d] -> dst[((b*H+h)*DH + d)*S_pad + s], zero for
// s in [S, S_pad).  One block per (64-token tile, h, b).
// =================================================================================================
template <int DH>
__global__ __launch_bounds__(256) void head_transpose_kernel(const bf16_t* __restrict__ src,
                                                             bf16_t* __restrict__ dst, int S, int H,
                                                             int ld, int col0, int S_pad) {
  __shared__ bf16_t tile[64][DH + 2];
  const int s0 = blockIdx.x * 64, h = blockIdx.y, b = blockIdx.z;
  const int t = threadIdx.x;
  constexpr int CPR = DH / 8;
  for (int idx = t; idx < 64 * CPR; idx += 256) {
    const int r = idx / CPR, c = (idx % CPR) * 8;
    u32x4 v = {0u, 0u, 0u, 0u};
    if (s0 + r < S)
      v = *reinterpret_cast<const u32x4*>(src + ((long)b * S + s0 + r) * ld + col0 + h * DH + c);
    unsigned* d = reinterpret_cast<unsigned*>(&tile[r][c]);
    d[0] = v[0]; d[1] = v[1]; d[2] = v[2]; d[3] = v[3];
  }
  __syncthreads();
  for (int idx = t; idx < DH * 8; idx += 256) {
    const int d = idx >> 3, r = (idx & 7) * 8;
    bf16_t e[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) e[j] = tile[r + j][d];
    u32x4 o = {e[0] | ((unsigned)e[1] << 16), e[2] | ((unsigned)e[3] << 16),
               e[4] | ((unsigned)e[5] << 16), e[6] | ((unsigned)e[7] << 16)};
    *reinterpret_cast<u32x4*>(dst + (((long)b * H + h) * DH + d) * S_pad + s0 + r) = o;
  }
}

// delta[b,h,s] = sum_d O[b,s,h,d] * dO[b,s,h,d]   (one thread per 8 elements, dh/8-lane reduce)
template <int DH>
__global__ __launch_bounds__(256) void attn_delta_kernel(const bf16_t* __restrict__ o,
                                                         const bf16_t* __restrict__ dout,
                                                         float* __restrict__ delta, int B, int S,
                                                         int H) {
  constexpr int LPH = DH / 8;  // lanes per (token, head)
  const long total = (long)B * S * H * LPH;
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  float acc = 0.f;
  long tokh = 0;
  if (i < total) {
    tokh = i / LPH;
    const u32x4 a = reinterpret_cast<const u32x4*>(o)[i];
    const u32x4 g = reinterpret_cast<const u32x4*>(dout)[i];
#pragma unroll
    for (int e = 0; e < 4; ++e) acc += bf_lo(a[e]) * bf_lo(g[e]) + bf_hi(a[e]) * bf_hi(g[e]);
  }
#pragma unroll
  for (int off = LPH / 2; off > 0; off >>= 1) acc += __shfl_xor(acc, off, 64);
  if (i < total && (i % LPH) == 0) {
    const long tok = tokh / H;
    const int h = (int)(tokh % H);
    const long b = tok / S;
    const int s = (int)(tok % S);
    delta[(b * H + h) * S + s] = acc;
  }
}

// =================================================================================================
// backward, dQ  (one query per lane)
// =================================================================================================
template <int DH, bool CAUSAL>
__global__ __launch_bounds__(256, 1) void attn_bwd_dq_kernel(
    const bf16_t* __restrict__ qkv, const bf16_t* __restrict__ kt_g /*K^T [B,H,DH,S_pad]*/,
    const bf16_t* __restrict__ dout, const float* __restrict__ lse, const float* __restrict__ delta,
    bf16_t* __restrict__ dqkv, int S, int H, int ld, int S_pad, float scale) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr unsigned KT = 64 * DH * 2;
  const unsigned sK = 0, sV = KT, sKT = 2 * KT;
  const int tid = threadIdx.x, l = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l31 = l & 31, hi = l >> 5;
  const int nqt = (S + 127) / 128;
  const int qt = CAUSAL ? (nqt - 1 - (int)blockIdx.x) : (int)blockIdx.x;
  const int h = blockIdx.y, b = blockIdx.z;
  const int q0 = qt * 128;
  const int qi = q0 + w * 32 + l31;
  const int qc = min(qi, S - 1);
  const long tok0 = (long)b * S;
  const float scale_log2 = scale * LOG2E;

  bf16x8 qf[DH / 16], dof[DH / 16];
  {
    const bf16_t* qp = qkv + (tok0 + qc) * ld + h * DH + hi * 8;
    const bf16_t* dp = dout + (tok0 + qc) * (long)(H * DH) + h * DH + hi * 8;
#pragma unroll
    for (int ks = 0; ks < DH / 16; ++ks) {
      qf[ks] = *reinterpret_cast<const bf16x8*>(qp + ks * 16);
      dof[ks] = *reinterpret_cast<const bf16x8*>(dp + ks * 16);
    }
  }
  const float lse2 = lse[((long)b * H + h) * S + qc] * LOG2E;
  const float dlt = delta[((long)b * H + h) * S + qc];

  f32x16 dq[DH / 32];
#pragma unroll
  for (int db = 0; db < DH / 32; ++db)
#pragma unroll
    for (int r = 0; r < 16; ++r) dq[db][r] = 0.f;

  const int nkt_all = (S + 63) / 64;
  const int nkt = CAUSAL ? min(nkt_all, min(q0 + 127, S - 1) / 64 + 1) : nkt_all;
  const bf16_t* kbase = qkv + tok0 * ld + (long)H * DH + h * DH;
  const bf16_t* vbase = qkv + tok0 * ld + 2L * H * DH + h * DH;
  const bf16_t* ktbase = kt_g + ((long)b * H + h) * DH * (long)S_pad;

  for (int kt = 0; kt < nkt; ++kt) {
    const int kv0 = kt * 64;
    __syncthreads();   // previous tile fully consumed
    load_tile<64, DH>(smem, sK, kbase, ld, kv0, S, w, l);
    load_tile<64, DH>(smem, sV, vbase, ld, kv0, S, w, l);
    load_tile<DH, 64>(smem, sKT, ktbase + kv0, S_pad, 0, DH, w, l);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (CAUSAL && kv0 > q0 + w * 32 + 31) continue;

    f32x16 s[2], dp[2];
#pragma unroll
    for (int kb = 0; kb < 2; ++kb)
#pragma unroll
      for (int r = 0; r < 16; ++r) { s[kb][r] = 0.f; dp[kb][r] = 0.f; }
#pragma unroll
    for (int ks = 0; ks < DH / 16; ++ks)
#pragma unroll
      for (int kb = 0; kb < 2; ++kb) {
        const unsigned a = tile_addr<DH>(kb * 32 + l31, ks * 2 + hi);
        const bf16x8 kf = lds_frag128(smem, sK + a);
        s[kb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf, qf[ks], s[kb], 0, 0, 0);
        const bf16x8 vf = lds_frag128(smem, sV + a);
        dp[kb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf, dof[ks], dp[kb], 0, 0, 0);
      }
#pragma unroll
    for (int kb = 0; kb < 2; ++kb)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int key = kv0 + kb * 32 + crow(r, hi);
        float p = fast_exp2(s[kb][r] * scale_log2 - lse2);
        if (key >= S || (CAUSAL && key > qi)) p = 0.f;
        s[kb][r] = p * (dp[kb][r] - dlt) * scale;   // dS^T
      }
#pragma unroll
    for (int kb = 0; kb < 2; ++kb)
#pragma unroll
      for (int sp = 0; sp < 2; ++sp) {
        float pv[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) pv[j] = s[kb][8 * sp + j];
        const bf16x8 dsf = pack8(pv);
        const int c1 = kb * 4 + 2 * sp;
#pragma unroll
        for (int db = 0; db < DH / 32; ++db) {
          const int r = db * 32 + l31;
          const bf16x8 ktf = lds_frag64x2(smem, sKT + tile_addr<64>(r, c1) + 8u * hi,
                                          sKT + tile_addr<64>(r, c1 + 1) + 8u * hi);
          dq[db] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ktf, dsf, dq[db], 0, 0, 0);
        }
      }
  }
  if (qi < S) {
    bf16_t* op = dqkv + (tok0 + qi) * (long)ld + h * DH;
#pragma unroll
    for (int db = 0; db < DH / 32; ++db)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int d = db * 32 + 8 * g + 4 * hi;
        u32x2 v = {pack_bf2(dq[db][4 * g], dq[db][4 * g + 1]), pack_bf2(dq[db][4 * g + 2], dq[db][4 * g + 3])};
        *reinterpret_cast<u32x2*>(op + d) = v;
      }
  }
}

// =================================================================================================
// backward, dK / dV  (one key per lane)
// =================================================================================================
template <int DH, bool CAUSAL>
__global__ __launch_bounds__(256, 1) void attn_bwd_dkv_kernel(
    const bf16_t* __restrict__ qkv, const bf16_t* __restrict__ qt_g /*Q^T*/,
    const bf16_t* __restrict__ dout, const bf16_t* __restrict__ dot_g /*dO^T*/,
    const float* __restrict__ lse, const float* __restrict__ delta, bf16_t* __restrict__ dqkv, int S,
    int H, int ld, int S_pad, float scale) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr unsigned KT = 64 * DH * 2;
  const unsigned sQ = 0, sDO = KT, sQT = 2 * KT, sDOT = 3 * KT;
  float* s_lse = reinterpret_cast<float*>(smem + 4 * KT);   // [64] lse*log2e
  float* s_dlt = s_lse + 64;                                // [64]
  const int tid = threadIdx.x, l = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l31 = l & 31, hi = l >> 5;
  const int kblk = blockIdx.x, h = blockIdx.y, b = blockIdx.z;
  const int k0 = kblk * 128;
  const int ki = k0 + w * 32 + l31;
  const int kc = min(ki, S - 1);
  const long tok0 = (long)b * S;
  const float scale_log2 = scale * LOG2E;

  bf16x8 kf[DH / 16], vf[DH / 16];
  {
    const bf16_t* kp = qkv + (tok0 + kc) * ld + (long)H * DH + h * DH + hi * 8;
    const bf16_t* vp = qkv + (tok0 + kc) * ld + 2L * H * DH + h * DH + hi * 8;
#pragma unroll
    for (int ks = 0; ks < DH / 16; ++ks) {
      kf[ks] = *reinterpret_cast<const bf16x8*>(kp + ks * 16);
      vf[ks] = *reinterpret_cast<const bf16x8*>(vp + ks * 16);
    }
  }
  f32x16 dk[DH / 32], dv[DH / 32];
#pragma unroll
  for (int db = 0; db < DH / 32; ++db)
#pragma unroll
    for (int r = 0; r < 16; ++r) { dk[db][r] = 0.f; dv[db][r] = 0.f; }

  const int nqt = (S + 63) / 64;
  const int qt_begin = CAUSAL ? (k0 / 64) : 0;
  const bf16_t* qbase = qkv + tok0 * ld + h * DH;
  const bf16_t* dobase = dout + tok0 * (long)(H * DH) + h * DH;
  const bf16_t* qtbase = qt_g + ((long)b * H + h) * DH * (long)S_pad;
  const bf16_t* dotbase = dot_g + ((long)b * H + h) * DH * (long)S_pad;
  const float* lse_row = lse + ((long)b * H + h) * S;
  const float* dlt_row = delta + ((long)b * H + h) * S;

  for (int qt = qt_begin; qt < nqt; ++qt) {
    const int qq0 = qt * 64;
    __syncthreads();
    load_tile<64, DH>(smem, sQ, qbase, ld, qq0, S, w, l);
    load_tile<64, DH>(smem, sDO, dobase, (long)H * DH, qq0, S, w, l);
    load_tile<DH, 64>(smem, sQT, qtbase + qq0, S_pad, 0, DH, w, l);
    load_tile<DH, 64>(smem, sDOT, dotbase + qq0, S_pad, 0, DH, w, l);
    if (tid < 64) {
      const int q = min(qq0 + tid, S - 1);
      s_lse[tid] = lse_row[q] * LOG2E;
      s_dlt[tid] = dlt_row[q];
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    // tile fully masked for this wave: all queries of the tile precede all of the wave's keys
    if (CAUSAL && qq0 + 63 < k0 + w * 32) continue;

#pragma unroll
    for (int qb = 0; qb < 2; ++qb) {
      f32x16 s, dp;
#pragma unroll
      for (int r = 0; r < 16; ++r) { s[r] = 0.f; dp[r] = 0.f; }
#pragma unroll
      for (int ks = 0; ks < DH / 16; ++ks) {
        const unsigned a = tile_addr<DH>(qb * 32 + l31, ks * 2 + hi);
        const bf16x8 qfr = lds_frag128(smem, sQ + a);
        s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(qfr, kf[ks], s, 0, 0, 0);
        const bf16x8 dofr = lds_frag128(smem, sDO + a);
        dp = __builtin_amdgcn_mfma_f32_32x32x16_bf16(dofr, vf[ks], dp, 0, 0, 0);
      }
      float pvals[16], dsvals[16];
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int ql = qb * 32 + crow(r, hi);
        const int q = qq0 + ql;
        float p = fast_exp2(s[r] * scale_log2 - s_lse[ql]);
        if (q >= S || ki >= S || (CAUSAL && ki > q)) p = 0.f;
        pvals[r] = p;
        dsvals[r] = p * (dp[r] - s_dlt[ql]) * scale;
      }
#pragma unroll
      for (int sp = 0; sp < 2; ++sp) {
        const bf16x8 pf = pack8(pvals + 8 * sp);
        const bf16x8 dsf = pack8(dsvals + 8 * sp);
        const int c1 = qb * 4 + 2 * sp;
#pragma unroll
        for (int db = 0; db < DH / 32; ++db) {
          const int r = db * 32 + l31;
          const unsigned a0 = tile_addr<64>(r, c1) + 8u * hi, a1 = tile_addr<64>(r, c1 + 1) + 8u * hi;
          const bf16x8 dotf = lds_frag64x2(smem, sDOT + a0, sDOT + a1);
          dv[db] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(dotf, pf, dv[db], 0, 0, 0);
          const bf16x8 qtf = lds_frag64x2(smem, sQT + a0, sQT + a1);
          dk[db] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(qtf, dsf, dk[db], 0, 0, 0);
        }
      }
    }
  }
  if (ki < S) {
    bf16_t* kp = dqkv + (tok0 + ki) * (long)ld + (long)H * DH + h * DH;
    bf16_t* vp = dqkv + (tok0 + ki) * (long)ld + 2L * H * DH + h * DH;
#pragma unroll
    for (int db = 0; db < DH / 32; ++db)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int d = db * 32 + 8 * g + 4 * hi;
        u32x2 a = {pack_bf2(dk[db][4 * g], dk[db][4 * g + 1]), pack_bf2(dk[db][4 * g + 2], dk[db][4 * g + 3])};
        u32x2 c = {pack_bf2(dv[db][4 * g], dv[db][4 * g + 1]), pack_bf2(dv[db][4 * g + 2], dv[db][4 * g + 3])};
        *reinterpret_cast<u32x2*>(kp + d) = a;
        *reinterpret_cast<u32x2*>(vp + d) = c;
      }
  }
}

static inline int spad(int S) { return (S + 63) / 64 * 64; }

template <typename K>
static int set_lds(K kernel, int bytes, const char* what) {
  hipError_t e = hipFuncSetAttribute((const void*)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
  if (e != hipSuccess) { set_error("%s: hipFuncSetAttribute: %s", what, hipGetErrorString(e)); return HVLM_ERR_LAUNCH; }
  return HVLM_OK;
}

template <int DH>
static int launch_head_transpose(const bf16_t* src, bf16_t* dst, int B, int S, int H, int ld, int col0,
                                 hipStream_t st) {
  dim3 grid(spad(S) / 64, H, B);
  hipLaunchKernelGGL((head_transpose_kernel<DH>), grid, dim3(256), 0, st, src, dst, S, H, ld, col0, spad(S));
  return check_launch("head_transpose");
}

template <int DH, bool CAUSAL>
static int fwd_impl(const bf16_t* qkv, bf16_t* out, float* lse, bf16_t* ws, int B, int S, int H, int ld,
                    float scale, hipStream_t st) {
  int rc = launch_head_transpose<DH>(qkv, ws, B, S, H, ld, 2 * H * DH, st);
  if (rc) return rc;
  constexpr int LDS = 2 * 2 * 64 * DH * 2;
  // rows per workgroup: 128 (4 waves) or 96 (3 waves) — whichever wastes fewer query rows (S=257 -> 96)
  const int waste4 = (S + 127) / 128 * 128 - S, waste3 = (S + 95) / 96 * 96 - S;
  if (false && waste3 < waste4) {   // 96-row workgroups measured slower in situ (649 vs 593 us per CLIP layer): kept for A/B only
    rc = set_lds(attn_fwd_kernel<DH, CAUSAL, 3>, LDS, "attn_fwd");
    if (rc) return rc;
    dim3 grid((S + 95) / 96, H, B);
    hipLaunchKernelGGL((attn_fwd_kernel<DH, CAUSAL, 3>), grid, dim3(192), LDS, st, qkv, (const bf16_t*)ws, out,
                       lse, S, H, ld, spad(S), scale * LOG2E);
  } else {
    rc = set_lds(attn_fwd_kernel<DH, CAUSAL, 4>, LDS, "attn_fwd");
    if (rc) return rc;
    dim3 grid((S + 127) / 128, H, B);
    hipLaunchKernelGGL((attn_fwd_kernel<DH, CAUSAL, 4>), grid, dim3(256), LDS, st, qkv, (const bf16_t*)ws, out,
                       lse, S, H, ld, spad(S), scale * LOG2E);
  }
  return check_launch("attn_fwd");
}

template <int DH, bool CAUSAL>
static int bwd_impl(const bf16_t* qkv, const bf16_t* out, const bf16_t* dout, const float* lse,
                    bf16_t* dqkv, char* ws, int B, int S, int H, int ld, float scale, hipStream_t st) {
  const size_t tsz = (size_t)B * H * DH * spad(S) * 2;
  bf16_t* qT = reinterpret_cast<bf16_t*>(ws);
  bf16_t* kT = reinterpret_cast<bf16_t*>(ws + tsz);
  bf16_t* doT = reinterpret_cast<bf16_t*>(ws + 2 * tsz);
  float* delta = reinterpret_cast<float*>(ws + 3 * tsz);
  int rc;
  if ((rc = launch_head_transpose<DH>(qkv, qT, B, S, H, ld, 0, st))) return rc;
  if ((rc = launch_head_transpose<DH>(qkv, kT, B, S, H, ld, H * DH, st))) return rc;
  if ((rc = launch_head_transpose<DH>(dout, doT, B, S, H, H * DH, 0, st))) return rc;
  {
    const long total = (long)B * S * H * (DH / 8);
    hipLaunchKernelGGL((attn_delta_kernel<DH>), dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, out,
                       dout, delta, B, S, H);
    if ((rc = check_launch("attn_delta"))) return rc;
  }
  {
    constexpr int LDS = 3 * 64 * DH * 2;
    if ((rc = set_lds(attn_bwd_dq_kernel<DH, CAUSAL>, LDS, "attn_bwd_dq"))) return rc;
    dim3 grid((S + 127) / 128, H, B);
    hipLaunchKernelGGL((attn_bwd_dq_kernel<DH, CAUSAL>), grid, dim3(256), LDS, st, qkv, (const bf16_t*)kT,
                       dout, lse, (const float*)delta, dqkv, S, H, ld, spad(S), scale);
    if ((rc = check_launch("attn_bwd_dq"))) return rc;
  }
  {
    constexpr int LDS = 4 * 64 * DH * 2 + 512;
    if ((rc = set_lds(attn_bwd_dkv_kernel<DH, CAUSAL>, LDS, "attn_bwd_dkv"))) return rc;
    dim3 grid((S + 127) / 128, H, B);
    hipLaunchKernelGGL((attn_bwd_dkv_kernel<DH, CAUSAL>), grid, dim3(256), LDS, st, qkv, (const bf16_t*)qT,
                       dout, (const bf16_t*)doT, lse, (const float*)delta, dqkv, S, H, ld, spad(S), scale);
    if ((rc = check_launch("attn_bwd_dkv"))) return rc;
  }
  return HVLM_OK;
}

}  // namespace
}  // namespace hvlm

using namespace hvlm;

extern "C" size_t hvlm_attn_fwd_ws_bytes(int B, int S, int H, int dh) {
  return (size_t)B * H * dh * spad(S) * 2;
}
extern "C" size_t hvlm_attn_bwd_ws_bytes(int B, int S, int H, int dh) {
  return 3 * (size_t)B * H * dh * spad(S) * 2 + (size_t)B * H * S * 4 + 256;
}

extern "C" int hvlm_attn_fwd(const void* qkv, void* out, float* lse, void* ws, int B, int S, int H, int dh,
                             int ld, int causal, float scale, void* stream) {
  HVLM_REQUIRE(qkv && out && ws, "attn_fwd: null pointer");
  HVLM_REQUIRE(B > 0 && S > 0 && H > 0 && ld >= 3 * H * dh && ld % 8 == 0, "attn_fwd: bad shape");
  HVLM_REQUIRE(dh == 64 || dh == 128, "attn_fwd: dh must be 64 or 128 (got %d)", dh);
  hipStream_t st = as_stream(stream);
  const bf16_t* q = (const bf16_t*)qkv;
  if (dh == 128) return causal ? fwd_impl<128, true>(q, (bf16_t*)out, lse, (bf16_t*)ws, B, S, H, ld, scale, st)
                               : fwd_impl<128, false>(q, (bf16_t*)out, lse, (bf16_t*)ws, B, S, H, ld, scale, st);
  return causal ? fwd_impl<64, true>(q, (bf16_t*)out, lse, (bf16_t*)ws, B, S, H, ld, scale, st)
                : fwd_impl<64, false>(q, (bf16_t*)out, lse, (bf16_t*)ws, B, S, H, ld, scale, st);
}

extern "C" int hvlm_attn_bwd(const void* qkv, const void* out, const void* dout, const float* lse,
                             void* dqkv, void* ws, int B, int S, int H, int dh, int ld, int causal,
                             float scale, void* stream) {
  HVLM_REQUIRE(qkv && out && dout && lse && dqkv && ws, "attn_bwd: null pointer");
  HVLM_REQUIRE(B > 0 && S > 0 && H > 0 && ld >= 3 * H * dh && ld % 8 == 0, "attn_bwd: bad shape");
  HVLM_REQUIRE(dh == 64 || dh == 128, "attn_bwd: dh must be 64 or 128 (got %d)", dh);
  HVLM_REQUIRE(((uintptr_t)ws & 15) == 0, "attn_bwd: workspace must be 16-byte aligned");
  hipStream_t st = as_stream(stream);
  const bf16_t* q = (const bf16_t*)qkv;
  const bf16_t* o = (const bf16_t*)out;
  const bf16_t* g = (const bf16_t*)dout;
  if (dh == 128) return causal ? bwd_impl<128, true>(q, o, g, lse, (bf16_t*)dqkv, (char*)ws, B, S, H, ld, scale, st)
                               : bwd_impl<128, false>(q, o, g, lse, (bf16_t*)dqkv, (char*)ws, B, S, H, ld, scale, st);
  return causal ? bwd_impl<64, true>(q, o, g, lse, (bf16_t*)dqkv, (char*)ws, B, S, H, ld, scale, st)
                : bwd_impl<64, false>(q, o, g, lse, (bf16_t*)dqkv, (char*)ws, B, S, H, ld, scale, st);
}
