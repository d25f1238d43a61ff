// Single-token decode kernels for KV-cached generation (batch 1, the reference's eval recipe:
// handsonvlm/evaluation/handsonvlm_inference.py:99-109 — which itself runs cache-less; the cache is exact
// because a generated token's input embedding never changes after it is appended).
// Both kernels are HBM-bound: the GEMV streams the weight matrix once with 16-byte loads (one output row
// per wave, 4 rows in flight per wave for memory-level parallelism); the attention kernel streams the
// K/V cache of one head per workgroup.
#include "common.h"

namespace hvlm {
namespace {

// y[n] = sum_k W[n,k] x[k] (+ res[n]);  one wave per ROWS consecutive rows, x staged in LDS.
template <int ROWS>
__global__ __launch_bounds__(256) void gemv_kernel(const bf16_t* __restrict__ W, const bf16_t* __restrict__ x,
                                                   bf16_t* __restrict__ y, const bf16_t* __restrict__ res, int N,
                                                   int K, int ldw) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  bf16_t* xs = reinterpret_cast<bf16_t*>(smem);
  for (int i = threadIdx.x * 8; i < K; i += 256 * 8)
    *reinterpret_cast<u32x4*>(xs + i) = *reinterpret_cast<const u32x4*>(x + i);
  __syncthreads();
  const int l = threadIdx.x & 63;
  const int wave = blockIdx.x * 4 + (threadIdx.x >> 6);
  const int n0 = wave * ROWS;
  if (n0 >= N) return;
  float acc[ROWS];
#pragma unroll
  for (int r = 0; r < ROWS; ++r) acc[r] = 0.f;
  constexpr int UNR = 4;                         // ROWS*UNR 16-byte loads in flight per lane
  const bf16_t* wrow[ROWS];
#pragma unroll
  for (int r = 0; r < ROWS; ++r) wrow[r] = W + (size_t)min(n0 + r, N - 1) * ldw;
  int k = l * 8;
  for (; k + (UNR - 1) * 512 < K; k += UNR * 512) {
    u32x4 wv[UNR][ROWS];
#pragma unroll
    for (int u = 0; u < UNR; ++u)
#pragma unroll
      for (int r = 0; r < ROWS; ++r)
        wv[u][r] = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(wrow[r] + k + u * 512));
#pragma unroll
    for (int u = 0; u < UNR; ++u) {
      const u32x4 xv = *reinterpret_cast<const u32x4*>(xs + k + u * 512);
#pragma unroll
      for (int r = 0; r < ROWS; ++r)
#pragma unroll
        for (int e = 0; e < 4; ++e)
          acc[r] += bf_lo(wv[u][r][e]) * bf_lo(xv[e]) + bf_hi(wv[u][r][e]) * bf_hi(xv[e]);
    }
  }
  for (; k < K; k += 512) {
    const u32x4 xv = *reinterpret_cast<const u32x4*>(xs + k);
#pragma unroll
    for (int r = 0; r < ROWS; ++r) {
      const u32x4 wv = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(wrow[r] + k));
#pragma unroll
      for (int e = 0; e < 4; ++e) acc[r] += bf_lo(wv[e]) * bf_lo(xv[e]) + bf_hi(wv[e]) * bf_hi(xv[e]);
    }
  }
#pragma unroll
  for (int r = 0; r < ROWS; ++r) acc[r] = wave_sum(acc[r]);
  if (l == 0) {
#pragma unroll
    for (int r = 0; r < ROWS; ++r) {
      const int n = n0 + r;
      if (n < N) y[n] = f2bf(acc[r] + (res ? bf2f(res[n]) : 0.f));
    }
  }
}

// One workgroup (256 threads) per head.  q [H*dh] (one token), K/V cache [S, ld] token-major.
// Phase 1: one key per thread (16-byte loads of its K row, q broadcast from LDS) -> scores in LDS;
// phase 2: block softmax; phase 3: channel pairs x key groups stream V with coalesced 4-byte loads.
// (A first version walked the keys serially per wave and was latency-bound: 7.6 ms per generated token.)
template <int DH>
__global__ __launch_bounds__(256) void attn_decode_kernel(const bf16_t* __restrict__ q,
                                                          const bf16_t* __restrict__ kc,
                                                          const bf16_t* __restrict__ vc, bf16_t* __restrict__ out,
                                                          int S, int ld, float scale) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float* qs = reinterpret_cast<float*>(smem);          // [DH]
  float* red = qs + DH;                                  // [8]
  float* part = red + 8;                                 // [G][DH] partial outputs
  constexpr int CP = DH / 2;                             // channel pairs per key row
  constexpr int G = 256 / CP;                            // key groups in phase 3
  float* sc = part + G * DH;                             // [S] scores / probabilities
  const int h = blockIdx.x, t = threadIdx.x;
  if (t < DH) qs[t] = bf2f(q[h * DH + t]) * scale;
  __syncthreads();
  float mx = -INFINITY;
  for (int s0 = t; s0 < S; s0 += 256) {
    const bf16_t* kp = kc + (size_t)s0 * ld + h * DH;
    float dot = 0.f;
#pragma unroll
    for (int c = 0; c < DH; c += 8) {
      const u32x4 kv = *reinterpret_cast<const u32x4*>(kp + c);
#pragma unroll
      for (int e = 0; e < 4; ++e) dot += bf_lo(kv[e]) * qs[c + 2 * e] + bf_hi(kv[e]) * qs[c + 2 * e + 1];
    }
    sc[s0] = dot;
    mx = fmaxf(mx, dot);
  }
  mx = block_max<4>(mx, red);
  float sum = 0.f;
  for (int s0 = t; s0 < S; s0 += 256) {
    const float p = __expf(sc[s0] - mx);
    sc[s0] = p;
    sum += p;
  }
  sum = block_sum<4>(sum, red + 4);
  __syncthreads();
  const int cp = t % CP, g = t / CP;
  float a0 = 0.f, a1 = 0.f;
  const bf16_t* vbase = vc + h * DH + 2 * cp;
#pragma unroll 8
  for (int s0 = g; s0 < S; s0 += G) {
    const unsigned v = *reinterpret_cast<const unsigned*>(vbase + (size_t)s0 * ld);
    const float p = sc[s0];
    a0 += p * bf_lo(v);
    a1 += p * bf_hi(v);
  }
  part[g * DH + 2 * cp] = a0;
  part[g * DH + 2 * cp + 1] = a1;
  __syncthreads();
  if (t < DH) {
    float acc = 0.f;
#pragma unroll
    for (int i = 0; i < G; ++i) acc += part[i * DH + t];
    out[h * DH + t] = f2bf(acc / sum);
  }
}

}  // namespace
}  // namespace hvlm

using namespace hvlm;

extern "C" int hvlm_gemv_bf16(const void* W, const void* x, void* y, const void* residual, int N, int K, int ldw,
                              void* stream) {
  HVLM_REQUIRE(W && x && y, "gemv: null pointer");
  HVLM_REQUIRE(N > 0 && K > 0 && K % 8 == 0 && ldw >= K && ldw % 8 == 0, "gemv: bad shape (K %% 8, ldw %% 8)");
  HVLM_REQUIRE(K * 2 <= 64 * 1024, "gemv: K too large for the LDS copy of x");
  constexpr int ROWS = 2;
  const int waves = cdiv(N, ROWS);
  hipLaunchKernelGGL((gemv_kernel<ROWS>), dim3(cdiv(waves, 4)), dim3(256), K * 2, as_stream(stream), (const bf16_t*)W,
                     (const bf16_t*)x, (bf16_t*)y, (const bf16_t*)residual, N, K, ldw);
  return check_launch("gemv_bf16");
}

extern "C" int hvlm_attn_decode(const void* q, const void* k_cache, const void* v_cache, void* out, int S, int H, int dh,
                                int ld, float scale, void* stream) {
  HVLM_REQUIRE(q && k_cache && v_cache && out, "attn_decode: null pointer");
  HVLM_REQUIRE(S > 0 && H > 0 && ld >= H * dh, "attn_decode: bad shape");
  HVLM_REQUIRE(dh == 64 || dh == 128, "attn_decode: dh must be 64 or 128 (got %d)", dh);
  const size_t lds = (size_t)(dh + 8 + (256 / (dh / 2)) * dh + S) * 4;
  HVLM_REQUIRE(lds <= 64 * 1024, "attn_decode: S=%d too long for the LDS score buffer", S);
  if (dh == 128)
    hipLaunchKernelGGL((attn_decode_kernel<128>), dim3(H), dim3(256), lds, as_stream(stream), (const bf16_t*)q,
                       (const bf16_t*)k_cache, (const bf16_t*)v_cache, (bf16_t*)out, S, ld, scale);
  else
    hipLaunchKernelGGL((attn_decode_kernel<64>), dim3(H), dim3(256), lds, as_stream(stream), (const bf16_t*)q,
                       (const bf16_t*)k_cache, (const bf16_t*)v_cache, (bf16_t*)out, S, ld, scale);
  return check_launch("attn_decode");
}

// One whole decode step of the decoder stack, launched from native code (no per-kernel host round trip):
// 9 launches per layer on the caller's stream.  Pointer tables are HOST arrays of device pointers.
extern "C" int hvlm_llama_decode_step(const void* const* w_qkv, const void* const* w_o, const void* const* w_gu,
                                      const void* const* w_down, const void* const* ln1, const void* const* ln2,
                                      void* const* kv_cache, const void* norm_w, const void* cos_tab, const void* sin_tab,
                                      const int32_t* pos_dev, const void* x_in, void* hidden_out, void* ws, int L, int D,
                                      int F, int H, int dh, int pos, int cache_ld, float eps, void* stream) {
  HVLM_REQUIRE(w_qkv && w_o && w_gu && w_down && ln1 && ln2 && kv_cache && norm_w && x_in && hidden_out && ws,
               "decode_step: null pointer");
  HVLM_REQUIRE(cache_ld >= 3 * D && D == H * dh, "decode_step: bad shape");
  bf16_t* w = reinterpret_cast<bf16_t*>(ws);
  bf16_t* xn = w;            // [D]
  bf16_t* ao = xn + D;       // [D]
  bf16_t* h = ao + D;        // [D]
  bf16_t* hn = h + D;        // [D]
  bf16_t* xa = hn + D;       // [D]
  bf16_t* xb = xa + D;       // [D]
  bf16_t* gu = xb + D;       // [2F]
  bf16_t* act = gu + 2 * (size_t)F;  // [F]
  const float scale = 1.0f / sqrtf((float)dh);
  const bf16_t* x = reinterpret_cast<const bf16_t*>(x_in);
  int rc;
  for (int l = 0; l < L; ++l) {
    bf16_t* kv = reinterpret_cast<bf16_t*>(kv_cache[l]);
    bf16_t* row = kv + (size_t)pos * cache_ld;
    if ((rc = hvlm_rmsnorm_fwd(x, ln1[l], xn, nullptr, 1, D, eps, stream))) return rc;
    if ((rc = hvlm_gemv_bf16(w_qkv[l], xn, row, nullptr, 3 * D, D, D, stream))) return rc;
    if ((rc = hvlm_rope_qk(row, cos_tab, sin_tab, pos_dev, 1, H, dh, cache_ld, 0, stream))) return rc;
    if ((rc = hvlm_attn_decode(row, kv + D, kv + 2 * D, ao, pos + 1, H, dh, cache_ld, scale, stream))) return rc;
    if ((rc = hvlm_gemv_bf16(w_o[l], ao, h, x, D, D, D, stream))) return rc;
    if ((rc = hvlm_rmsnorm_fwd(h, ln2[l], hn, nullptr, 1, D, eps, stream))) return rc;
    if ((rc = hvlm_gemv_bf16(w_gu[l], hn, gu, nullptr, 2 * F, D, D, stream))) return rc;
    if ((rc = hvlm_swiglu_fwd(gu, act, 1, F, stream))) return rc;
    bf16_t* xo = (l & 1) ? xb : xa;
    if ((rc = hvlm_gemv_bf16(w_down[l], act, xo, h, D, F, F, stream))) return rc;
    x = xo;
  }
  return hvlm_rmsnorm_fwd(x, norm_w, hidden_out, nullptr, 1, D, eps, stream);
}

extern "C" size_t hvlm_llama_decode_ws_bytes(int D, int F) { return (size_t)(6 * D + 3 * (size_t)F) * 2 + 256; }
