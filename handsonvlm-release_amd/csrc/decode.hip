// Single-token decode kernels for KV-cached generation (batch 1, the reference's eval recipe:
// handsonvlm/evaluation/handsonvlm_inference.py:99-109 — which itself runs cache-less; the cache is exact
// because a generated token's input embedding never changes after it is appended).
// Both kernels are HBM-bound: the GEMV streams the weight matrix once with 16-byte loads (one output row
// per wave, 4 rows in flight per wave for memory-level parallelism); the attention kernel streams the
// K/V cache of one head per workgroup.
#include "common.h"

namespace hvlm {
namespace {

// y[n] = sum_k W[n,k] x'[k] (+ res[n]);  one wave per ROWS consecutive rows, x' staged in LDS.
// yrow != null: y is advanced by *yrow * ystride elements first (read on the device).
// PRE selects what x' is (computed by every workgroup while it stages the 8-22 KB vector: cheaper than a kernel launch):
//   0: x' = x;   1: x' = RMSNorm(x) * nw  (same arithmetic as rmsnorm_fwd_kernel: bf16(x*rstd) then * w, bf16);
//   2: x' = SwiGLU(x) with x = [gate(K) | up(K)]  (same arithmetic as swiglu_fwd_kernel).
// (Round 6, measured and not kept — profiles/r06h_decode_kernel_stats_regression.txt: requesting the first two weight blocks of every wave BEFORE
//  the prologue stages x, two register sets: 164 VGPRs instead of ~50, three waves per SIMD instead of eight, and x — which returns behind the
//  weights, loads complete in order — arrives last: q|k|v + gate|up 28.2 vs 26.2 us, o_proj 8.9 vs 8.1 us, the step 3.31 vs 3.18 ms per token.
//  The kernel's speed comes from MANY short waves whose prologues overlap each other's streams, not from one deep pipeline per wave.)
template <int ROWS, int PRE>
__global__ __launch_bounds__(256) void gemv_kernel(const bf16_t* __restrict__ W, const bf16_t* __restrict__ x,
                                                   bf16_t* __restrict__ y, const bf16_t* __restrict__ res, int N,
                                                   int K, int ldw, const bf16_t* __restrict__ nw, float eps,
                                                   const int32_t* __restrict__ yrow, long ystride, int pair_off) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  bf16_t* xs = reinterpret_cast<bf16_t*>(smem);
  if (yrow) y += (long)(*yrow) * ystride;      // graph-replayed decode: the output row (KV-cache position) lives on the device
  if constexpr (PRE == 0) {
    for (int i = threadIdx.x * 8; i < K; i += 256 * 8)
      *reinterpret_cast<u32x4*>(xs + i) = *reinterpret_cast<const u32x4*>(x + i);
  } else if constexpr (PRE == 1) {
    __shared__ float red[4];
    // (round 6: x crosses the L2 once — the first pass parks the raw chunk in xs, where the second pass of the SAME thread finds it)
    float ss = 0.f;
    for (int i = threadIdx.x * 8; i < K; i += 256 * 8) {
      const u32x4 v = *reinterpret_cast<const u32x4*>(x + i);
      *reinterpret_cast<u32x4*>(xs + i) = v;
#pragma unroll
      for (int e = 0; e < 4; ++e) { const float a = bf_lo(v[e]), b = bf_hi(v[e]); ss += a * a + b * b; }
    }
    ss = block_sum<4>(ss, red);
    const float rstd = rsqrtf(ss / (float)K + eps);
    for (int i = threadIdx.x * 8; i < K; i += 256 * 8) {
      const u32x4 v = *reinterpret_cast<const u32x4*>(xs + i);
      const u32x4 wv = *reinterpret_cast<const u32x4*>(nw + i);
      u32x4 o;
#pragma unroll
      for (int e = 0; e < 4; ++e)
        o[e] = pack_bf2(bf2f(f2bf(bf_lo(v[e]) * rstd)) * bf_lo(wv[e]), bf2f(f2bf(bf_hi(v[e]) * rstd)) * bf_hi(wv[e]));
      *reinterpret_cast<u32x4*>(xs + i) = o;
    }
  } else {
    for (int i = threadIdx.x * 8; i < K; i += 256 * 8) {
      const u32x4 g = *reinterpret_cast<const u32x4*>(x + i);
      const u32x4 u = *reinterpret_cast<const u32x4*>(x + K + i);
      u32x4 o;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float g0 = bf_lo(g[e]), g1 = bf_hi(g[e]);
        const float s0 = bf2f(f2bf(g0 * (1.f / (1.f + __expf(-g0))))), s1 = bf2f(f2bf(g1 * (1.f / (1.f + __expf(-g1)))));
        o[e] = pack_bf2(s0 * bf_lo(u[e]), s1 * bf_hi(u[e]));
      }
      *reinterpret_cast<u32x4*>(xs + i) = o;
    }
  }
  __syncthreads();
  const int l = threadIdx.x & 63;
  const int wave = blockIdx.x * 4 + (threadIdx.x >> 6);
  // pair_off != 0 (ROWS == 2; the decode step's gate|up projection, round 6): the wave's two rows are gate row n and up row n + pair_off of the
  // fused [gate; up] weight, N counts PAIRS and y[n] = SwiGLU(bf16(gate . x), bf16(up . x)) — the arithmetic of PRE == 2 below (and of
  // swiglu_fwd_kernel) on the same bf16-rounded pair, so the down projection's input is bit-identical; the 2 F-element gate|up vector is never
  // written, and the down GEMV no longer evaluates F exponentials in EVERY workgroup's prologue (43 per thread at F = 11008)
  const int n0 = pair_off ? wave : wave * ROWS;
  if (n0 >= N) return;
  float acc[ROWS];
#pragma unroll
  for (int r = 0; r < ROWS; ++r) acc[r] = 0.f;
  constexpr int UNR = 4;                         // ROWS*UNR 16-byte loads in flight per lane
  const bf16_t* wrow[ROWS];
#pragma unroll
  for (int r = 0; r < ROWS; ++r) wrow[r] = W + (pair_off ? (size_t)(n0 + r * pair_off) : (size_t)min(n0 + r, N - 1)) * ldw;
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
    if constexpr (ROWS == 2) {
      if (pair_off) {
        const float g0 = bf2f(f2bf(acc[0])), u0 = bf2f(f2bf(acc[1]));
        const float s0 = bf2f(f2bf(g0 * (1.f / (1.f + __expf(-g0)))));
        y[n0] = f2bf(s0 * u0);
        return;
      }
    }
#pragma unroll
    for (int r = 0; r < ROWS; ++r) {
      const int n = n0 + r;
      if (n < N) y[n] = f2bf(acc[r] + (res ? bf2f(res[n]) : 0.f));
    }
  }
}

// gemv_kernel with the first weight block requested BEHIND the loads of x and AHEAD of the prologue's arithmetic (round 6, second attempt —
// the first one, weights before x with two register sets, lost: see the note above gemv_kernel): x's chunks (<= XC per thread: K <= 2048 XC)
// are requested first, so they return first — loads complete in order —, the wave's first UNR x ROWS weight chunks right behind them, and the
// RMSNorm / SwiGLU arithmetic runs while those are in flight.  One register set, the loop and every sum exactly as gemv_kernel: bit-identical.
template <int ROWS, int PRE, int XC>
__global__ __launch_bounds__(256) void gemv_early_kernel(const bf16_t* __restrict__ W, const bf16_t* __restrict__ x,
                                                         bf16_t* __restrict__ y, const bf16_t* __restrict__ res, int N,
                                                         int K, int ldw, const bf16_t* __restrict__ nw, float eps,
                                                         const int32_t* __restrict__ yrow, long ystride, int pair_off) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  bf16_t* xs = reinterpret_cast<bf16_t*>(smem);
  constexpr int UNR = 4;
  const int t = threadIdx.x, l = t & 63;
  const int wave = blockIdx.x * 4 + (t >> 6);
  const int n0 = pair_off ? wave : wave * ROWS;
  // 1. x (and, PRE = 2, the up half) — chunk j of thread t is element (t + 256 j) * 8; chunks past K re-read the last one
  u32x4 xv[XC], uv[PRE == 2 ? XC : 1];
  const int last = K / 8 - 1;
#pragma unroll
  for (int j = 0; j < XC; ++j) {
    const int c = min(t + 256 * j, last);
    xv[j] = *reinterpret_cast<const u32x4*>(x + (size_t)c * 8);
    if constexpr (PRE == 2) uv[j] = *reinterpret_cast<const u32x4*>(x + K + (size_t)c * 8);
  }
  // 2. the first weight block
  const bf16_t* wrow[ROWS];
#pragma unroll
  for (int r = 0; r < ROWS; ++r) wrow[r] = W + (pair_off ? (size_t)(min(n0, N - 1) + r * pair_off) : (size_t)min(n0 + r, N - 1)) * ldw;
  u32x4 wv[UNR][ROWS];
  const bool has0 = (UNR - 1) * 512 + l * 8 < K;
  if (has0) {
#pragma unroll
    for (int u = 0; u < UNR; ++u)
#pragma unroll
      for (int r = 0; r < ROWS; ++r)
        wv[u][r] = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(wrow[r] + l * 8 + u * 512));
  }
  if (yrow) y += (long)(*yrow) * ystride;
  // 3. the prologue's arithmetic (gemv_kernel's, on the chunks in registers)
  if constexpr (PRE == 0) {
#pragma unroll
    for (int j = 0; j < XC; ++j) { const int c = t + 256 * j; if (c <= last) *reinterpret_cast<u32x4*>(xs + (size_t)c * 8) = xv[j]; }
  } else if constexpr (PRE == 1) {
    __shared__ float red[4];
    float ss = 0.f;
#pragma unroll
    for (int j = 0; j < XC; ++j) {
      if (t + 256 * j <= last) {
#pragma unroll
        for (int e = 0; e < 4; ++e) { const float a = bf_lo(xv[j][e]), b = bf_hi(xv[j][e]); ss += a * a + b * b; }
      }
    }
    ss = block_sum<4>(ss, red);
    const float rstd = rsqrtf(ss / (float)K + eps);
#pragma unroll
    for (int j = 0; j < XC; ++j) {
      const int c = t + 256 * j;
      if (c <= last) {
        const u32x4 nv = *reinterpret_cast<const u32x4*>(nw + (size_t)c * 8);
        u32x4 o;
#pragma unroll
        for (int e = 0; e < 4; ++e)
          o[e] = pack_bf2(bf2f(f2bf(bf_lo(xv[j][e]) * rstd)) * bf_lo(nv[e]), bf2f(f2bf(bf_hi(xv[j][e]) * rstd)) * bf_hi(nv[e]));
        *reinterpret_cast<u32x4*>(xs + (size_t)c * 8) = o;
      }
    }
  } else {
#pragma unroll
    for (int j = 0; j < XC; ++j) {
      const int c = t + 256 * j;
      if (c <= last) {
        u32x4 o;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const float g0 = bf_lo(xv[j][e]), g1 = bf_hi(xv[j][e]);
          const float s0 = bf2f(f2bf(g0 * (1.f / (1.f + __expf(-g0))))), s1 = bf2f(f2bf(g1 * (1.f / (1.f + __expf(-g1)))));
          o[e] = pack_bf2(s0 * bf_lo(uv[j][e]), s1 * bf_hi(uv[j][e]));
        }
        *reinterpret_cast<u32x4*>(xs + (size_t)c * 8) = o;
      }
    }
  }
  __syncthreads();
  if (n0 >= N) return;
  float acc[ROWS];
#pragma unroll
  for (int r = 0; r < ROWS; ++r) acc[r] = 0.f;
  int k = l * 8;
  bool have = has0;
  while (have) {
#pragma unroll
    for (int u = 0; u < UNR; ++u) {
      const u32x4 xq = *reinterpret_cast<const u32x4*>(xs + k + u * 512);
#pragma unroll
      for (int r = 0; r < ROWS; ++r)
#pragma unroll
        for (int e = 0; e < 4; ++e)
          acc[r] += bf_lo(wv[u][r][e]) * bf_lo(xq[e]) + bf_hi(wv[u][r][e]) * bf_hi(xq[e]);
    }
    k += UNR * 512;
    have = k + (UNR - 1) * 512 < K;
    if (have) {
#pragma unroll
      for (int u = 0; u < UNR; ++u)
#pragma unroll
        for (int r = 0; r < ROWS; ++r)
          wv[u][r] = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(wrow[r] + k + u * 512));
    }
  }
  for (; k < K; k += 512) {
    const u32x4 xq = *reinterpret_cast<const u32x4*>(xs + k);
#pragma unroll
    for (int r = 0; r < ROWS; ++r) {
      const u32x4 wq = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(wrow[r] + k));
#pragma unroll
      for (int e = 0; e < 4; ++e) acc[r] += bf_lo(wq[e]) * bf_lo(xq[e]) + bf_hi(wq[e]) * bf_hi(xq[e]);
    }
  }
#pragma unroll
  for (int r = 0; r < ROWS; ++r) acc[r] = wave_sum(acc[r]);
  if (l == 0) {
    if constexpr (ROWS == 2) {
      if (pair_off) {
        const float g0 = bf2f(f2bf(acc[0])), u0 = bf2f(f2bf(acc[1]));
        const float s0 = bf2f(f2bf(g0 * (1.f / (1.f + __expf(-g0)))));
        y[n0] = f2bf(s0 * u0);
        return;
      }
    }
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

// Flash-decoding split of the kernel above for the native decode step: grid (H, NS); workgroup (h, j) handles keys
// [j*S/NS, (j+1)*S/NS) and writes {local max, local sum, unnormalised o[DH]} (DH + 2 floats).  One workgroup per head
// used 32 of 256 CUs: 32 us per layer at S = 870.
// Fused into it (each was a 5-8 us kernel boundary per layer of an HBM-bound step that has 70 us of GEMV per layer):
//   * RoPE of the new token (cos_tab != null): q is rotated while it is staged in LDS; the new K row (key S-1, raw in the
//     cache) is rotated by the one thread that scores it, which also writes it back for the tokens to come.  Same arithmetic
//     as rope_kernel: o1 = bf16(bf16(x1*c) + bf16(-x2*s)), o2 = bf16(bf16(x2*c) + bf16(x1*s));
//   * the combine of the NS partials (cnt != null): the LAST workgroup of a head to finish (per-head ticket counter in the
//     workspace, zero before first use, reset by that workgroup) merges them and writes the head's output.
template <int DH>
__global__ __launch_bounds__(256) void attn_decode_partial_kernel(const bf16_t* __restrict__ q, const bf16_t* __restrict__ kc,
                                                                  const bf16_t* __restrict__ vc, float* __restrict__ part_out,
                                                                  int S, int ld, float scale, int NS,
                                                                  const int32_t* __restrict__ pos_dev,
                                                                  const bf16_t* __restrict__ cos_tab,
                                                                  const bf16_t* __restrict__ sin_tab, int* __restrict__ cnt,
                                                                  bf16_t* __restrict__ out) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float* qs = reinterpret_cast<float*>(smem);          // [DH]
  float* red = qs + DH;                                  // [8]
  constexpr int CP = DH / 2;
  constexpr int G = 256 / CP;
  constexpr int HALF = DH / 2;
  float* part = red + 8;                                 // [G][DH]
  float* sc = part + G * DH;                             // [keys of this split]
  const int h = blockIdx.x, j = blockIdx.y, t = threadIdx.x;
  if (pos_dev) {                                         // graph-replayed decode: q is the cache base, the position is on the device
    const int p = *pos_dev;
    S = p + 1;
    q += (size_t)p * ld;
  }
  const int s_begin = (int)((long)S * j / NS), s_end = (int)((long)S * (j + 1) / NS);
  const int n = s_end - s_begin;
  const bool rope = cos_tab != nullptr;
  const bf16_t* ct = rope ? cos_tab + (size_t)(S - 1) * HALF : nullptr;
  const bf16_t* st = rope ? sin_tab + (size_t)(S - 1) * HALF : nullptr;
  if (t < DH) {
    if (rope) {
      const int c = t < HALF ? t : t - HALF;
      const float x1 = bf2f(q[h * DH + c]), x2 = bf2f(q[h * DH + c + HALF]);
      const float cc = bf2f(ct[c]), ss = bf2f(st[c]);
      const float r = t < HALF ? bf2f(f2bf(x1 * cc)) + bf2f(f2bf(-x2 * ss)) : bf2f(f2bf(x2 * cc)) + bf2f(f2bf(x1 * ss));
      qs[t] = bf2f(f2bf(r)) * scale;
    } else {
      qs[t] = bf2f(q[h * DH + t]) * scale;
    }
  }
  // the new token's key (rope mode, last split): raw in the cache.  8 threads rotate it (one 8-pair chunk each), write it
  // back for the tokens to come and park it in LDS; it is then scored like any other key, from LDS.
  const bool has_new = rope && j == NS - 1;
  const int n_old = has_new ? n - 1 : n;
  bf16_t* knew = reinterpret_cast<bf16_t*>(sc + ((n + 3) & ~3));            // [DH] bf16, 16-byte aligned
  if (has_new && t < HALF / 8) {
    const int c = 8 * t;
    bf16_t* kp = const_cast<bf16_t*>(kc) + (size_t)(S - 1) * ld + h * DH;
    const u32x4 x1 = *reinterpret_cast<const u32x4*>(kp + c), x2 = *reinterpret_cast<const u32x4*>(kp + c + HALF);
    const u32x4 cv = *reinterpret_cast<const u32x4*>(ct + c), sv = *reinterpret_cast<const u32x4*>(st + c);
    u32x4 o1, o2;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const float a[2] = {bf_lo(x1[e]), bf_hi(x1[e])}, b[2] = {bf_lo(x2[e]), bf_hi(x2[e])};
      const float cc[2] = {bf_lo(cv[e]), bf_hi(cv[e])}, ss[2] = {bf_lo(sv[e]), bf_hi(sv[e])};
      float r1[2], r2[2];
#pragma unroll
      for (int k = 0; k < 2; ++k) {
        r1[k] = bf2f(f2bf(a[k] * cc[k])) + bf2f(f2bf(-b[k] * ss[k]));
        r2[k] = bf2f(f2bf(b[k] * cc[k])) + bf2f(f2bf(a[k] * ss[k]));
      }
      o1[e] = pack_bf2(r1[0], r1[1]);
      o2[e] = pack_bf2(r2[0], r2[1]);
    }
    *reinterpret_cast<u32x4*>(kp + c) = o1;
    *reinterpret_cast<u32x4*>(kp + c + HALF) = o2;
    *reinterpret_cast<u32x4*>(knew + c) = o1;
    *reinterpret_cast<u32x4*>(knew + c + HALF) = o2;
  }
  __syncthreads();
  float mx = -INFINITY;
  auto score = [&](auto kp) {                  // kp: global or LDS pointer to one key row of this head
    float dot = 0.f;
#pragma unroll
    for (int c = 0; c < DH; c += 8) {
      const u32x4 kv = *reinterpret_cast<const u32x4*>(kp + c);
#pragma unroll
      for (int e = 0; e < 4; ++e) dot += bf_lo(kv[e]) * qs[c + 2 * e] + bf_hi(kv[e]) * qs[c + 2 * e + 1];
    }
    return dot;
  };
  for (int i = t; i < n_old; i += 256) {
    const float dot = score(kc + (size_t)(s_begin + i) * ld + h * DH);
    sc[i] = dot;
    mx = fmaxf(mx, dot);
  }
  if (has_new && t == (n_old & 255)) {
    const float dot = score(static_cast<const bf16_t*>(knew));
    sc[n_old] = dot;
    mx = fmaxf(mx, dot);
  }
  mx = block_max<4>(mx, red);
  float sum = 0.f;
  for (int i = t; i < n; i += 256) {
    const float p = __expf(sc[i] - mx);
    sc[i] = p;
    sum += p;
  }
  sum = block_sum<4>(sum, red + 4);
  __syncthreads();
  const int cp = t % CP, g = t / CP;
  float a0 = 0.f, a1 = 0.f;
  const bf16_t* vbase = vc + (size_t)s_begin * ld + h * DH + 2 * cp;
#pragma unroll 8
  for (int i = g; i < n; i += G) {
    const unsigned v = *reinterpret_cast<const unsigned*>(vbase + (size_t)i * ld);
    const float p = sc[i];
    a0 += p * bf_lo(v);
    a1 += p * bf_hi(v);
  }
  part[g * DH + 2 * cp] = a0;
  part[g * DH + 2 * cp + 1] = a1;
  __syncthreads();
  float* po = part_out + ((size_t)h * NS + j) * (DH + 2);
  // (agent-scope relaxed atomic stores / loads = write-through, L1-bypassing accesses: the partials cross XCDs without the
  // L2 write-back + invalidate of a release / acquire fence, which cost ~20 us per layer here)
  auto st_f = [](float* ptr, float v) {
    __hip_atomic_store(reinterpret_cast<int*>(ptr), __builtin_bit_cast(int, v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  };
  if (t < DH) {
    float acc = 0.f;
#pragma unroll
    for (int i = 0; i < G; ++i) acc += part[i * DH + t];
    st_f(po + 2 + t, acc);
  }
  if (t == 0) { st_f(po, n > 0 ? mx : -INFINITY); st_f(po + 1, n > 0 ? sum : 0.f); }
  if (cnt == nullptr) return;
  // fused combine: the last of the head's NS workgroups merges the partials
  __shared__ int is_last;
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // this thread's partial stores have reached the coherence point
  __syncthreads();
  if (t == 0) is_last = (__hip_atomic_fetch_add(&cnt[h], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == NS - 1);
  __syncthreads();
  if (!is_last) return;
  if (t < DH) {
    auto ld_f = [](const float* ptr) {
      return __builtin_bit_cast(float, __hip_atomic_load(reinterpret_cast<const int*>(ptr), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
    };
    const float* p = part_out + (size_t)h * NS * (DH + 2);
    float m = -INFINITY;
    for (int jj = 0; jj < NS; ++jj) m = fmaxf(m, ld_f(p + jj * (DH + 2)));
    float l = 0.f, o = 0.f;
    for (int jj = 0; jj < NS; ++jj) {
      const float mj = ld_f(p + jj * (DH + 2));
      const float wj = (mj == -INFINITY) ? 0.f : __expf(mj - m);
      l += ld_f(p + jj * (DH + 2) + 1) * wj;
      o += ld_f(p + jj * (DH + 2) + 2 + t) * wj;
    }
    out[h * DH + t] = f2bf(o / l);
  }
  if (t == 0) __hip_atomic_store(&cnt[h], 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// sum over a group of 8 or 16 adjacent lanes (aligned to the group); every lane of the group gets the total
template <int LPR>
__device__ __forceinline__ float group_sum(float v) {
  static_assert(LPR == 8 || LPR == 16, "lane groups of 8 or 16");
  int x = __float_as_int(v);
  v += __int_as_float(__builtin_amdgcn_update_dpp(0, x, 0xB1, 0xF, 0xF, true));            // quad_perm [1,0,3,2]
  x = __float_as_int(v);
  v += __int_as_float(__builtin_amdgcn_update_dpp(0, x, 0x4E, 0xF, 0xF, true));            // quad_perm [2,3,0,1]
  x = __float_as_int(v);
  v += __int_as_float(__builtin_amdgcn_update_dpp(0, x, 0x141, 0xF, 0xF, true));           // row_half_mirror: lane i <-> 7 - i
  if constexpr (LPR == 16) {
    x = __float_as_int(v);
    v += __int_as_float(__builtin_amdgcn_update_dpp(0, x, 0x140, 0xF, 0xF, true));         // row_mirror: lane i <-> 15 - i
  }
  return v;
}

// Round 6: the split kernel above walks its keys one row per THREAD (16 dependent 16-byte loads per key, 108 of 256 threads busy at
// S = 867 / 8 splits) and then the values four bytes per lane — a chain of three HBM round trips that took 16.5 us per layer for 14 MB
// (profiles/r06g_decode_kernel_stats_before.txt), a sixth of the decode step.  This kernel reads a row with DH / 8 ADJACENT lanes x
// 16 B (one wave instruction = 64 * 8 / DH whole rows, as attn_decode_batched_kernel) and requests the K rows AND the V rows of its
// split before anything else — values do not depend on scores — so ONE round trip covers the whole split (<= 128 keys: U loads of K
// and U of V per lane).  No block-wide softmax: every lane group keeps (max, sum, acc[DH]) of its own U rows (all lanes of a group hold
// the group's scores after the DPP sum), and the groups are merged flash-decoding style through LDS — one barrier instead of five.
// Same contract as attn_decode_partial_kernel (RoPE of the new token fused when cos_tab != null, combine of the splits by the head's
// last workgroup when cnt != null, position on the device when pos_dev != null); fused and unfused forms share every instruction that
// touches the data, so they stay bit-identical to each other (tests/test_gpu_model.py).
// ENG = the body runs inside decode_stack_kernel (one launch for the whole stack): the workgroup has more than 256 threads (the others pass
// the barriers and do nothing: `act`), and everything another workgroup wrote EARLIER IN THE SAME LAUNCH — the new token's q | k | v row —
// is read with agent-scope (L1-bypassing, sc1) loads, the output written with agent-scope stores: no kernel boundary makes them coherent.
__device__ __forceinline__ u32x4 load16_agent(const void* ptr) {
  u32x4 v;
  asm volatile("global_load_dwordx4 %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(ptr) : "memory");
  return v;
}
__device__ __forceinline__ bf16_t load2_agent(const bf16_t* ptr) {
  return __hip_atomic_load(ptr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
template <int DH, bool ENG>
__device__ __forceinline__ void attn_decode_split_body(const bf16_t* __restrict__ q, const bf16_t* __restrict__ kc,
                                                       const bf16_t* __restrict__ vc, float* __restrict__ part_out,
                                                       int S, int ld, float scale, int NS,
                                                       const bf16_t* __restrict__ cos_tab, const bf16_t* __restrict__ sin_tab,
                                                       int* __restrict__ cnt, bf16_t* __restrict__ out, const int h, const int j) {
  constexpr int LPR = DH / 8;            // lanes per row (16 B each)
  constexpr int RPI = 64 / LPR;          // rows per wave instruction
  constexpr int NG = 4 * RPI;            // lane groups per workgroup
  constexpr int U = 128 / NG;            // rows per lane group: the split holds at most 128 keys
  constexpr int HALF = DH / 2;
  __shared__ __attribute__((aligned(16))) float qs[DH];
  __shared__ __attribute__((aligned(16))) bf16_t knew[DH];
  // one buffer, two uses: per lane group (acc[DH], max, sum) at stride DH + 4, then — in the head's last workgroup — the NS partials
  constexpr int GS = DH + 4, MAXNS = 32;
  constexpr int BUFN = NG * GS > MAXNS * (DH + 2) ? NG * GS : MAXNS * (DH + 2);
  __shared__ __attribute__((aligned(16))) float buf[BUFN];
  auto gp = [&](int i) -> float* { return buf + i * GS; };
  __shared__ int is_last;
  const int t = threadIdx.x, w = t >> 6, l = t & 63;
  const bool act = !ENG || t < 256;     // (uniform per wave)
  const int c = l % LPR, g = (w & 3) * RPI + l / LPR;
  // (32-bit arithmetic throughout: S * NS and rows * ld are bounded by the launcher; 64-bit division and multiplication are emulated
  //  by long instruction sequences in front of the first load)
  const int s_begin = (int)((unsigned)(S * j) / (unsigned)NS), s_end = (int)((unsigned)(S * (j + 1)) / (unsigned)NS);
  const int n = s_end - s_begin;         // <= 128 (launcher)
  const bool rope = cos_tab != nullptr;
  const bool has_new = rope && j == NS - 1;          // this split holds the new token's key, still un-rotated in the cache
  // 1. every K and V row of the split, in flight at once: unconditional loads (rows past the split re-read its last row — a branch per
  // load made the compiler wait for each one before issuing the next)
  u32x4 kv[U], vv[U];
  if (act) {
  const bf16_t* kb = kc + (long)(s_begin * ld) + h * DH + c * 8;
  const bf16_t* vb = vc + (long)(s_begin * ld) + h * DH + c * 8;
#pragma unroll
  for (int u = 0; u < U; ++u) kv[u] = *reinterpret_cast<const u32x4*>(kb + (long)(max(min(g + u * NG, n - 1), 0) * ld));
#pragma unroll
  for (int u = 0; u < U; ++u) vv[u] = *reinterpret_cast<const u32x4*>(vb + (long)(max(min(g + u * NG, n - 1), 0) * ld));
  }
  // 2. q (rotated here when the RoPE is fused) -> LDS as scaled floats; the new key rotated, written back and parked in LDS
  const bf16_t* ct = rope ? cos_tab + (size_t)(S - 1) * HALF : nullptr;
  const bf16_t* st = rope ? sin_tab + (size_t)(S - 1) * HALF : nullptr;
  if (t < DH) {
    if (rope) {
      const int cc_ = t < HALF ? t : t - HALF;
      const float x1 = bf2f(ENG ? load2_agent(q + h * DH + cc_) : q[h * DH + cc_]);
      const float x2 = bf2f(ENG ? load2_agent(q + h * DH + cc_ + HALF) : q[h * DH + cc_ + HALF]);
      const float cc = bf2f(ct[cc_]), ss = bf2f(st[cc_]);
      const float r = t < HALF ? bf2f(f2bf(x1 * cc)) + bf2f(f2bf(-x2 * ss)) : bf2f(f2bf(x2 * cc)) + bf2f(f2bf(x1 * ss));
      qs[t] = bf2f(f2bf(r)) * scale;
    } else {
      qs[t] = bf2f(q[h * DH + t]) * scale;
    }
  }
  if (has_new && t >= 64 && t < 64 + HALF / 8) {     // (wave 1: wave 0's lanes are busy with q)
    const int c8 = 8 * (t - 64);
    bf16_t* kp = const_cast<bf16_t*>(kc) + (size_t)(S - 1) * ld + h * DH;
    const u32x4 x1 = ENG ? load16_agent(kp + c8) : *reinterpret_cast<const u32x4*>(kp + c8);
    const u32x4 x2 = ENG ? load16_agent(kp + c8 + HALF) : *reinterpret_cast<const u32x4*>(kp + c8 + HALF);
    const u32x4 cv = *reinterpret_cast<const u32x4*>(ct + c8), sv = *reinterpret_cast<const u32x4*>(st + c8);
    u32x4 o1, o2;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const float a[2] = {bf_lo(x1[e]), bf_hi(x1[e])}, b[2] = {bf_lo(x2[e]), bf_hi(x2[e])};
      const float cc[2] = {bf_lo(cv[e]), bf_hi(cv[e])}, ss[2] = {bf_lo(sv[e]), bf_hi(sv[e])};
      float r1[2], r2[2];
#pragma unroll
      for (int k = 0; k < 2; ++k) {
        r1[k] = bf2f(f2bf(a[k] * cc[k])) + bf2f(f2bf(-b[k] * ss[k]));
        r2[k] = bf2f(f2bf(b[k] * cc[k])) + bf2f(f2bf(a[k] * ss[k]));
      }
      o1[e] = pack_bf2(r1[0], r1[1]);
      o2[e] = pack_bf2(r2[0], r2[1]);
    }
    *reinterpret_cast<u32x4*>(kp + c8) = o1;
    *reinterpret_cast<u32x4*>(kp + c8 + HALF) = o2;
    *reinterpret_cast<u32x4*>(knew + c8) = o1;
    *reinterpret_cast<u32x4*>(knew + c8 + HALF) = o2;
  }
  if constexpr (ENG) {                    // the new token's value row: written by other workgroups earlier in this launch
    if (act && j == NS - 1) {
#pragma unroll
      for (int u = 0; u < U; ++u)
        if (g + u * NG == n - 1) vv[u] = load16_agent(vc + (long)((S - 1) * ld) + h * DH + c * 8);
    }
  }
  __syncthreads();
  float qf[8];
  if (act) {
    const f32x4 q0 = *reinterpret_cast<const f32x4*>(qs + c * 8), q1 = *reinterpret_cast<const f32x4*>(qs + c * 8 + 4);
    qf[0] = q0[0]; qf[1] = q0[1]; qf[2] = q0[2]; qf[3] = q0[3]; qf[4] = q1[0]; qf[5] = q1[1]; qf[6] = q1[2]; qf[7] = q1[3];
  // 3. scores of the group's rows (every lane of the group ends up with them), the group's own softmax, P V from registers
  float sc[U];
  float mg = -INFINITY;
#pragma unroll
  for (int u = 0; u < U; ++u) {
    const int row = g + u * NG;
    u32x4 kk = kv[u];
    if (has_new && row == n - 1) kk = *reinterpret_cast<const u32x4*>(knew + c * 8);     // the rotated new key (the load above saw it raw)
    float dot = 0.f;
#pragma unroll
    for (int e = 0; e < 4; ++e) dot += bf_lo(kk[e]) * qf[2 * e] + bf_hi(kk[e]) * qf[2 * e + 1];
    dot = group_sum<LPR>(dot);
    sc[u] = row < n ? dot : -INFINITY;
    mg = fmaxf(mg, sc[u]);
  }
  float lg = 0.f, acc[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) acc[e] = 0.f;
#pragma unroll
  for (int u = 0; u < U; ++u) {
    const float p = (g + u * NG < n) ? __expf(sc[u] - mg) : 0.f;
    lg += p;
#pragma unroll
    for (int e = 0; e < 4; ++e) { acc[2 * e] += p * bf_lo(vv[u][e]); acc[2 * e + 1] += p * bf_hi(vv[u][e]); }
  }
  {
    float* pp = gp(g) + c * 8;
    *reinterpret_cast<f32x4*>(pp) = f32x4{acc[0], acc[1], acc[2], acc[3]};
    *reinterpret_cast<f32x4*>(pp + 4) = f32x4{acc[4], acc[5], acc[6], acc[7]};
    if (c == 0) { gp(g)[DH] = mg; gp(g)[DH + 1] = lg; }
  }
  }   // act
  __syncthreads();
  // 4. merge the lane groups: (max, sum, unnormalised o[DH]) of the split
  float* po = part_out + ((size_t)h * NS + j) * (DH + 2);
  auto st_f = [](float* ptr, float v) {      // agent-scope relaxed atomic store = write-through, L1-bypassing (see attn_decode_partial_kernel)
    __hip_atomic_store(reinterpret_cast<int*>(ptr), __builtin_bit_cast(int, v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  };
  if (t < DH) {
    float m = -INFINITY;
#pragma unroll
    for (int i = 0; i < NG; ++i) m = fmaxf(m, gp(i)[DH]);
    float lsum = 0.f, o = 0.f;
#pragma unroll
    for (int i = 0; i < NG; ++i) {
      const float mi = gp(i)[DH];
      const float wi = (mi == -INFINITY) ? 0.f : __expf(mi - m);
      lsum += gp(i)[DH + 1] * wi;
      o += gp(i)[t] * wi;
    }
    st_f(po + 2 + t, o);
    if (t == 0) { st_f(po, m); st_f(po + 1, lsum); }
  }
  if (cnt == nullptr) return;
  // fused combine: the last of the head's NS workgroups merges the partials (ticket counter, zero at rest)
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // this thread's partial stores have reached the coherence point
  __syncthreads();
  if (t == 0) is_last = (__hip_atomic_fetch_add(&cnt[h], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == NS - 1);
  __syncthreads();
  if (!is_last) return;
  // every float of the head's NS partials in flight at once (the round-3 kernel walked them split by split: 3 NS dependent L2 round trips,
  // most of its 16.5 us), through LDS; then one thread per output channel merges them
  if (act) {
    auto ld_f = [](const float* ptr) {
      return __builtin_bit_cast(float, __hip_atomic_load(reinterpret_cast<const int*>(ptr), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
    };
    const float* p = part_out + (size_t)h * NS * (DH + 2);
    const int total = NS * (DH + 2);
    constexpr int PER = (MAXNS * (DH + 2) + 255) / 256;
    float tmp[PER];
#pragma unroll
    for (int k = 0; k < PER; ++k) { const int i = t + 256 * k; tmp[k] = i < total ? ld_f(p + i) : 0.f; }
#pragma unroll
    for (int k = 0; k < PER; ++k) { const int i = t + 256 * k; if (i < total) buf[i] = tmp[k]; }
  }
  __syncthreads();
  if (t < DH) {
    float m = -INFINITY;
    for (int jj = 0; jj < NS; ++jj) m = fmaxf(m, buf[jj * (DH + 2)]);
    float lsum = 0.f, o = 0.f;
    for (int jj = 0; jj < NS; ++jj) {
      const float mj = buf[jj * (DH + 2)];
      const float wj = (mj == -INFINITY) ? 0.f : __expf(mj - m);
      lsum += buf[jj * (DH + 2) + 1] * wj;
      o += buf[jj * (DH + 2) + 2 + t] * wj;
    }
    if constexpr (ENG) __hip_atomic_store(out + h * DH + t, f2bf(o / lsum), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    else out[h * DH + t] = f2bf(o / lsum);
  }
  if (t == 0) __hip_atomic_store(&cnt[h], 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

template <int DH>
__global__ __launch_bounds__(256) void attn_decode_split_kernel(const bf16_t* __restrict__ q, const bf16_t* __restrict__ kc,
                                                                const bf16_t* __restrict__ vc, float* __restrict__ part_out,
                                                                int S, int ld, float scale, int NS,
                                                                const int32_t* __restrict__ pos_dev,
                                                                const bf16_t* __restrict__ cos_tab,
                                                                const bf16_t* __restrict__ sin_tab, int* __restrict__ cnt,
                                                                bf16_t* __restrict__ out) {
  if (pos_dev) {                         // graph-replayed decode: q is the cache base, the position is on the device
    const int p = *pos_dev;
    S = p + 1;
    q += (size_t)p * ld;
  }
  attn_decode_split_body<DH, false>(q, kc, vc, part_out, S, ld, scale, NS, cos_tab, sin_tab, cnt, out, (int)blockIdx.x, (int)blockIdx.y);
}

template <int DH>
__global__ __launch_bounds__(DH) void attn_decode_combine_kernel(const float* __restrict__ part, bf16_t* __restrict__ out, int NS) {
  const int h = blockIdx.x, t = threadIdx.x;
  const float* p = part + (size_t)h * NS * (DH + 2);
  float m = -INFINITY;
  for (int j = 0; j < NS; ++j) m = fmaxf(m, p[j * (DH + 2)]);
  float l = 0.f, o = 0.f;
  for (int j = 0; j < NS; ++j) {
    const float mj = p[j * (DH + 2)];
    const float wj = (mj == -INFINITY) ? 0.f : __expf(mj - m);
    l += p[j * (DH + 2) + 1] * wj;
    o += p[j * (DH + 2) + 2 + t] * wj;
  }
  out[h * DH + t] = f2bf(o / l);
}

// RoPE of the one new q / k row of the graph-replayed decode step: row = kv + *pos * ld, tables indexed by *pos.  Same
// arithmetic as rope_kernel (elementwise.hip): o1 = bf16(bf16(x1*c) + bf16(-x2*s)), o2 = bf16(bf16(x2*c) + bf16(x1*s)).
__global__ __launch_bounds__(256) void rope_decode_kernel(bf16_t* __restrict__ kv, const bf16_t* __restrict__ cos_tab,
                                                          const bf16_t* __restrict__ sin_tab,
                                                          const int32_t* __restrict__ pos_dev, int H, int dh, int ld) {
  const int half = dh / 2, pv = half / 8;
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= 2 * H * pv) return;
  const int p = *pos_dev;
  const int hh = i / pv, c = (i - hh * pv) * 8;
  bf16_t* base = kv + (size_t)p * ld + (size_t)hh * dh + c;
  const u32x4 cv = *reinterpret_cast<const u32x4*>(cos_tab + (long)p * half + c);
  const u32x4 sv = *reinterpret_cast<const u32x4*>(sin_tab + (long)p * half + c);
  const u32x4 x1 = *reinterpret_cast<const u32x4*>(base);
  const u32x4 x2 = *reinterpret_cast<const u32x4*>(base + half);
  u32x4 o1, o2;
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    const float a[2] = {bf_lo(x1[e]), bf_hi(x1[e])}, b[2] = {bf_lo(x2[e]), bf_hi(x2[e])};
    const float cc[2] = {bf_lo(cv[e]), bf_hi(cv[e])}, ss[2] = {bf_lo(sv[e]), bf_hi(sv[e])};
    float r1[2], r2[2];
#pragma unroll
    for (int k = 0; k < 2; ++k) {
      r1[k] = bf2f(f2bf(a[k] * cc[k])) + bf2f(f2bf(-b[k] * ss[k]));
      r2[k] = bf2f(f2bf(b[k] * cc[k])) + bf2f(f2bf(a[k] * ss[k]));
    }
    o1[e] = pack_bf2(r1[0], r1[1]);
    o2[e] = pack_bf2(r2[0], r2[1]);
  }
  *reinterpret_cast<u32x4*>(base) = o1;
  *reinterpret_cast<u32x4*>(base + half) = o2;
}


// =====================================================================================================================================
// decode_stack_kernel (round 6): ONE persistent launch for the whole decoder stack of a batch-1 decode step — the same five phases per
// layer as the launch sequence of decode_step_impl (q|k|v GEMV with the RMSNorm in its prologue, split attention with the RoPE and the
// combine fused, o_proj GEMV + residual, gate|up GEMV as (gate, up) pairs with the SwiGLU in its epilogue, down GEMV + residual), separated
// by grid barriers instead of kernel boundaries.  What the single launch buys is NOT the boundary (a grid barrier costs more than a kernel
// boundary on this chip: MI355X_MICROARCH.md "Persistent kernels"): it is that a wave knows its rows of the NEXT phase's weight matrix and
// requests them before it arrives at the barrier, so the HBM stream does not stop while a phase drains, the barrier resolves, x is staged
// and normalised — 2.5 us per kernel in the launch sequence, four GEMVs and an 11 us attention per layer.
//   * one workgroup of ENG_NT threads per CU, rows of every phase split evenly over all waves (contiguous per workgroup);
//   * every per-row sum is accumulated in the order of gemv_kernel (16-byte chunks of the row ascending, the same expression, wave_sum), the
//     RMSNorm / SwiGLU arithmetic is gemv_kernel's, the attention IS attn_decode_split_body: the step's results are bit-identical to the
//     launch sequence's (tests/test_gpu_r6.py);
//   * hand-offs between phases: producers store with agent-scope (write-through) stores from ONE wave per workgroup, which waits for
//     them (vmcnt(0)) and then arrives at the barrier; consumers read with agent-scope (L1-bypassing) loads after the barrier
//     (MI355X_MICROARCH.md: "sc1 stores AND sc1 loads" — the per-XCD L2s are not coherent with each other);
//   * the barrier is one monotonic counter (zeroed by the launcher), polled by one lane per workgroup with s_sleep; every spin is BOUNDED:
//     on a timeout the workgroup raises *err and stops waiting (wrong results, reported by the host function) instead of hanging the GPU.
constexpr int ENG_NT = 512, ENG_W = ENG_NT / 64, ENG_MAXL = 48, ENG_UNR = 8;
struct EngineArgs {
  const bf16_t* wqkv[ENG_MAXL]; const bf16_t* wo[ENG_MAXL]; const bf16_t* wgu[ENG_MAXL]; const bf16_t* wdown[ENG_MAXL];
  const bf16_t* ln1[ENG_MAXL]; const bf16_t* ln2[ENG_MAXL]; bf16_t* kv[ENG_MAXL];
  const bf16_t* cos_tab; const bf16_t* sin_tab; const int32_t* pos_dev; const bf16_t* x_in;
  bf16_t* ao; bf16_t* h; bf16_t* xa; bf16_t* xb; bf16_t* act; float* apart; int* cnt; unsigned* bar; int* err;
  int L, D, F, H, pos, cache_ld, ns_cap;
  float eps, scale;
};
static_assert(sizeof(EngineArgs) <= 4096, "kernarg segment");

struct EngPF { u32x4 v[ENG_UNR]; };      // one block (ENG_UNR 16-byte chunks per lane) of a wave's weight stream, in flight

// rows [r0, r1) of wave gw when `n` rows are split evenly over `tw` waves (contiguous, workgroup-major)
__device__ __forceinline__ void eng_rows(int n, int tw, int gw, int& r0, int& r1) {
  r0 = (int)((long)n * gw / tw);
  r1 = (int)((long)n * (gw + 1) / tw);
}
// weight row of stream index rr: plain rows, or (pair mode) gate row rr >> 1 for even rr, up row (rr >> 1) + F for odd rr
template <bool PAIR>
__device__ __forceinline__ const bf16_t* eng_wrow(const bf16_t* W, int rr, int ldw, int pair_off) {
  if constexpr (PAIR) return W + (size_t)((rr >> 1) + (rr & 1) * pair_off) * ldw;
  else return W + (size_t)rr * ldw;
}
// request block `bi` of the wave's stream (rows rr0.., nblk blocks of ENG_UNR chunks per row); chunks past the row's end re-read its
// last chunk (masked at use) — unconditional loads, see attn_decode_split_body
template <bool PAIR>
__device__ __forceinline__ void eng_load(EngPF& b, const bf16_t* W, int rr0, int bi, int nblk, int K, int ldw, int pair_off, int l) {
  const int row = rr0 + bi / nblk, c0 = (bi % nblk) * ENG_UNR;
  const bf16_t* wp = eng_wrow<PAIR>(W, row, ldw, pair_off);
#pragma unroll
  for (int u = 0; u < ENG_UNR; ++u) {
    const int k = min((c0 + u) * 512 + l * 8, K - 8);
    b.v[u] = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(wp + k));
  }
}

__device__ __forceinline__ void eng_fail(const EngineArgs& a) { __hip_atomic_store(a.err, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

// grid barrier number `k` (1-based).  Called by every thread; `stores_by_wave0`: wave 0 has just issued this workgroup's agent-scope
// stores and waits for them before it arrives.
__device__ __forceinline__ void eng_barrier(const EngineArgs& a, unsigned k) {
  __syncthreads();
  if (threadIdx.x == 0) {
    const unsigned target = k * gridDim.x;
    __hip_atomic_fetch_add(a.bar, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    int spins = 0;
    // (once any workgroup has given up, nobody waits any more: the launch ends within one time-out, with wrong results and *err = 1)
    while (__hip_atomic_load(a.bar, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
      __builtin_amdgcn_s_sleep(4);
      if ((++spins & 1023) == 0 && __hip_atomic_load(a.err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) break;
      if (spins > (1 << 20)) { eng_fail(a); break; }               // (~1 s: never hang the GPU)
    }
  }
  __syncthreads();
}

// One GEMV phase.  PRE: 0 = x as is, 1 = RMSNorm(x) * nw.  PAIR: (gate, up) pairs + SwiGLU (n = pairs), else y[n] = W x (+ res[n]).
// `pf` holds block 0 of this wave's stream (requested before the barrier); on return `pf` holds block 0 of the NEXT phase's stream
// (Wn, nn rows / pairs, Kn, pair mode PAIRN) unless Wn is null.  xs: LDS, K bf16; ybuf: LDS, rows of this workgroup.
template <int PRE, bool PAIR, bool PAIRN>
__device__ __forceinline__ void eng_gemv(const EngineArgs& a, EngPF& pf, const bf16_t* W, int n, int K, const bf16_t* x, const bf16_t* nw,
                                         const bf16_t* res, bf16_t* y, const bf16_t* Wn, int nn, int Kn, bf16_t* xs, bf16_t* ybuf,
                                         float* red) {
  const int t = threadIdx.x, w = t >> 6, l = t & 63;
  const int tw = (int)gridDim.x * ENG_W, gw = (int)blockIdx.x * ENG_W + w;
  const int nstream = PAIR ? 2 * n : n;                           // rows of the weight stream
  int r0, r1, g0, g1;
  eng_rows(n, tw, gw, r0, r1);                                    // this wave's outputs
  eng_rows(n, (int)gridDim.x, (int)blockIdx.x, g0, g1);           // this workgroup's outputs (contiguous; r0 >= g0)
  {
    int q0, q1; eng_rows(n, tw, (int)blockIdx.x * ENG_W, q0, q1); g0 = q0;          // (exactly the first wave's first row)
    eng_rows(n, tw, (int)blockIdx.x * ENG_W + ENG_W - 1, q0, q1); g1 = q1;
  }
  (void)nstream;
  // residual values of this wave's rows (agent-scope: written earlier in this launch), one per lane
  float resv = 0.f;
  if (res && r0 + l < r1) resv = bf2f(load2_agent(res + r0 + l));
  // ---- stage x (agent-scope loads) -> xs, with gemv_kernel's prologue arithmetic
  const int nch = K / 8;
  for (int i = t; i < nch; i += ENG_NT) {
    u32x4 v;
    asm volatile("global_load_dwordx4 %0, %1, off sc1" : "=v"(v) : "v"(x + (size_t)i * 8) : "memory");
    asm volatile("s_waitcnt vmcnt(0)" : "+v"(v) : : "memory");
    *reinterpret_cast<u32x4*>(xs + (size_t)i * 8) = v;
  }
  if constexpr (PRE == 1) {
    __syncthreads();
    float ss = 0.f;
    if (t < 256) {                                                 // gemv_kernel's 256 threads, its order, its expression
      for (int i = t * 8; i < K; i += 256 * 8) {
        const u32x4 v = *reinterpret_cast<const u32x4*>(xs + i);
#pragma unroll
        for (int e = 0; e < 4; ++e) { const float p = bf_lo(v[e]), q = bf_hi(v[e]); ss += p * p + q * q; }
      }
      ss = wave_sum(ss);
      if (l == 0) red[w] = ss;
    }
    __syncthreads();
    float tot = 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i) tot += red[i];
    const float rstd = rsqrtf(tot / (float)K + a.eps);
    for (int i = t; i < nch; i += ENG_NT) {
      const u32x4 v = *reinterpret_cast<const u32x4*>(xs + (size_t)i * 8);
      const u32x4 wv = *reinterpret_cast<const u32x4*>(nw + (size_t)i * 8);
      u32x4 o;
#pragma unroll
      for (int e = 0; e < 4; ++e)
        o[e] = pack_bf2(bf2f(f2bf(bf_lo(v[e]) * rstd)) * bf_lo(wv[e]), bf2f(f2bf(bf_hi(v[e]) * rstd)) * bf_hi(wv[e]));
      *reinterpret_cast<u32x4*>(xs + (size_t)i * 8) = o;
    }
  }
  __syncthreads();
  // ---- the wave's stream: blocks of ENG_UNR chunks, block b + 1 requested before block b is multiplied
  const int nchunk = (K + 511) / 512, nblk = (nchunk + ENG_UNR - 1) / ENG_UNR;
  const int rr0 = PAIR ? 2 * r0 : r0, rr1 = PAIR ? 2 * r1 : r1;
  const int TB = (rr1 - rr0) * nblk;
  const int pair_off = PAIR ? a.F : 0;
  float acc = 0.f, gate = 0.f;
  EngPF nx;
  auto fma_blk = [&](const EngPF& b, int bi) {
    const int c0 = (bi % nblk) * ENG_UNR;
#pragma unroll
    for (int u = 0; u < ENG_UNR; ++u) {
      const int k = (c0 + u) * 512 + l * 8;
      if (k < K) {                                                  // (uniform for K % 4096 == 0; the tail block of F = 11008 masks lanes)
        const u32x4 xv = *reinterpret_cast<const u32x4*>(xs + k);
        const u32x4 wv = b.v[u];
#pragma unroll
        for (int e = 0; e < 4; ++e) acc += bf_lo(wv[e]) * bf_lo(xv[e]) + bf_hi(wv[e]) * bf_hi(xv[e]);
      }
    }
    if (bi % nblk == nblk - 1) {                                   // row complete
      const int rr = rr0 + bi / nblk;
      const float v = wave_sum(acc);
      acc = 0.f;
      if constexpr (PAIR) {
        if ((rr & 1) == 0) gate = v;
        else if (l == 0) {
          const float gq = bf2f(f2bf(gate)), uq = bf2f(f2bf(v));
          const float sg = bf2f(f2bf(gq * (1.f / (1.f + __expf(-gq)))));
          ybuf[(rr >> 1) - g0] = f2bf(sg * uq);
        }
      } else {
        const float rv = __shfl(resv, rr - r0, 64);
        if (l == 0) ybuf[rr - g0] = f2bf(v + (res ? rv : 0.f));
      }
    }
  };
  for (int bi = 0; bi < TB; bi += 2) {
    if (bi + 1 < TB) eng_load<PAIR>(nx, W, rr0, bi + 1, nblk, K, K, pair_off, l);
    fma_blk(pf, bi);
    if (bi + 1 >= TB) break;
    if (bi + 2 < TB) eng_load<PAIR>(pf, W, rr0, bi + 2, nblk, K, K, pair_off, l);
    fma_blk(nx, bi + 1);
  }
  // ---- next phase's first block (waves 1..: now; wave 0: behind its stores), results out through wave 0, barrier by the caller
  int nr0 = 0, nr1 = 0;
  if (Wn) eng_rows(nn, tw, gw, nr0, nr1);
  const int nnblk = Wn ? (((Kn + 511) / 512) + ENG_UNR - 1) / ENG_UNR : 1;
  const bool have_next = Wn && nr1 > nr0;
  if (w != 0 && have_next) eng_load<PAIRN>(pf, Wn, PAIRN ? 2 * nr0 : nr0, 0, nnblk, Kn, Kn, PAIRN ? a.F : 0, l);
  __syncthreads();
  if (w == 0) {
    for (int i = l; i < g1 - g0; i += 64) __hip_atomic_store(y + g0 + i, ybuf[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (have_next) eng_load<PAIRN>(pf, Wn, PAIRN ? 2 * nr0 : nr0, 0, nnblk, Kn, Kn, PAIRN ? a.F : 0, l);
  }
}

template <int DH>
__global__ __launch_bounds__(ENG_NT) void decode_stack_kernel(const EngineArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  bf16_t* xs = reinterpret_cast<bf16_t*>(smem);                          // [max(D, F)] bf16
  bf16_t* ybuf = xs + (a.F > a.D ? a.F : a.D);                           // [rows of this workgroup] (<= 3 D / grid + 1)
  __shared__ float red[4];
  const int t = threadIdx.x, w = t >> 6, l = t & 63;
  const int tw = (int)gridDim.x * ENG_W, gw = (int)blockIdx.x * ENG_W + w;
  const int D = a.D, F = a.F, H = a.H;
  int pos = a.pos;
  if (a.pos_dev) pos = *a.pos_dev;                                       // graph-replayed step: the position lives on the device
  const int S = pos + 1;
  // the launch sequence's split count (decode_step_impl), from the same S
  int NS;
  {
    const int c128 = (S + 127) / 128, c64 = (S + 63) / 64, per_cu = 256 / H > 1 ? 256 / H : 1;
    const int want = c64 < per_cu ? c64 : per_cu;
    NS = c128 > want ? c128 : want;
    if (NS > a.ns_cap) NS = a.ns_cap;
  }
  unsigned bk = 0;
  EngPF pf;
  {   // block 0 of layer 0's q|k|v stream
    int r0, r1; eng_rows(3 * D, tw, gw, r0, r1);
    const int nblk = (((D + 511) / 512) + ENG_UNR - 1) / ENG_UNR;
    if (r1 > r0) eng_load<false>(pf, a.wqkv[0], r0, 0, nblk, D, D, 0, l);
  }
  const bf16_t* x = a.x_in;
  for (int layer = 0; layer < a.L; ++layer) {
    bf16_t* kvl = a.kv[layer];
    bf16_t* row = kvl + (size_t)pos * a.cache_ld;
    // 1. q|k|v -> cache row `pos` (raw: the RoPE is the attention's)
    eng_gemv<1, false, false>(a, pf, a.wqkv[layer], 3 * D, D, x, a.ln1[layer], nullptr, row, a.wo[layer], D, D, xs, ybuf, red);
    eng_barrier(a, ++bk);
    // 2. attention: item (head, split) = workgroup index (H * NS <= grid: checked by the launcher)
    {
      const int item = (int)blockIdx.x;
      if (item < H * NS)
        attn_decode_split_body<DH, true>(row, kvl + D, kvl + 2 * D, a.apart, S, a.cache_ld, a.scale, NS, a.cos_tab, a.sin_tab, a.cnt, a.ao,
                                         item / NS, item % NS);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                  // (the head's last workgroup: its agent-scope stores of `ao`)
    }
    eng_barrier(a, ++bk);
    // 3. o_proj + residual -> h
    eng_gemv<0, false, true>(a, pf, a.wo[layer], D, D, a.ao, nullptr, x, a.h, a.wgu[layer], F, D, xs, ybuf, red);
    eng_barrier(a, ++bk);
    // 4. gate|up pairs + SwiGLU -> act
    eng_gemv<1, true, false>(a, pf, a.wgu[layer], F, D, a.h, a.ln2[layer], nullptr, a.act, a.wdown[layer], D, F, xs, ybuf, red);
    eng_barrier(a, ++bk);
    // 5. down + residual -> the next layer's x
    bf16_t* xo = (layer & 1) ? a.xb : a.xa;
    const bool last = layer + 1 == a.L;
    eng_gemv<0, false, false>(a, pf, a.wdown[layer], D, F, a.act, nullptr, a.h, xo, last ? nullptr : a.wqkv[layer + 1], 3 * D, D, xs, ybuf, red);
    eng_barrier(a, ++bk);
    x = xo;
  }
  if (a.pos_dev && blockIdx.x == 0 && t == 0) const_cast<int32_t*>(a.pos_dev)[0] = pos + 1;
}

__global__ void pos_advance_kernel(int32_t* pos_dev) { pos_dev[0] += 1; }

}  // namespace
}  // namespace hvlm

using namespace hvlm;

// GEMV with a fused prologue on x (pre: 0 none, 1 RMSNorm with weights `nw`, 2 SwiGLU of x = [gate | up]); internal to the
// native decode step (hvlm_gemv_bf16 is pre = 0).
static int gemv_pre(const void* W, const void* x, void* y, const void* residual, int N, int K, int ldw, int pre,
                    const void* nw, float eps, void* stream, const int32_t* yrow = nullptr, long ystride = 0, int pair_off = 0) {
  constexpr int ROWS = 2;
  const int waves = pair_off ? N : cdiv(N, ROWS);          // (pair mode: N = number of (gate, up) pairs, one per wave)
  dim3 grid(cdiv(waves, 4)), block(256);
  if (config().gemv_early && pre != 2 && K % 8 == 0 && K <= 2048 * 8) {
    // (x in <= 8 chunks per thread; K <= 4096: 2 — the q|k|v, o_proj and gate|up projections of the 7B model —, else 8)
#define HVLM_GEMV_EARLY(ROWS_, PRE_, XC_, GRID_)                                                                                          \
    hipLaunchKernelGGL((gemv_early_kernel<ROWS_, PRE_, XC_>), GRID_, block, K * 2, as_stream(stream), (const bf16_t*)W, (const bf16_t*)x, \
                       (bf16_t*)y, (const bf16_t*)residual, N, K, ldw, (const bf16_t*)nw, eps, yrow, ystride, pair_off)
    if (pre == 1) { if (K <= 4096) HVLM_GEMV_EARLY(2, 1, 2, grid); else HVLM_GEMV_EARLY(2, 1, 8, grid); }
    // (one row per wave for the narrow projections — twice the workgroups for o_proj / down — measured: 2.745 vs 2.751 ms per token, not kept)
    else { if (K <= 4096) HVLM_GEMV_EARLY(2, 0, 2, grid); else HVLM_GEMV_EARLY(2, 0, 8, grid); }
#undef HVLM_GEMV_EARLY
    return check_launch("gemv_early");
  }
  if (pre == 1)
    hipLaunchKernelGGL((gemv_kernel<ROWS, 1>), grid, block, K * 2, as_stream(stream), (const bf16_t*)W, (const bf16_t*)x,
                       (bf16_t*)y, (const bf16_t*)residual, N, K, ldw, (const bf16_t*)nw, eps, yrow, ystride, pair_off);
  else if (pre == 2)
    hipLaunchKernelGGL((gemv_kernel<ROWS, 2>), grid, block, K * 2, as_stream(stream), (const bf16_t*)W, (const bf16_t*)x,
                       (bf16_t*)y, (const bf16_t*)residual, N, K, ldw, (const bf16_t*)nullptr, 0.f, yrow, ystride, pair_off);
  else
    hipLaunchKernelGGL((gemv_kernel<ROWS, 0>), grid, block, K * 2, as_stream(stream), (const bf16_t*)W, (const bf16_t*)x,
                       (bf16_t*)y, (const bf16_t*)residual, N, K, ldw, (const bf16_t*)nullptr, 0.f, yrow, ystride, pair_off);
  return check_launch("gemv");
}

// Decode attention for lock-step generation of B samples (BASELINE config 5: batch-32 evaluation): grid (H, B); sample b has
// its own query row, its own cache slab (sample_stride elements apart) and its own length lens[b] (device int32: keys
// 0 .. lens[b]-1, the new token's key included).  The kernel is a stream of K rows and then V rows (2 * S * DH * 2 bytes per
// workgroup, nothing re-read), so both phases read a row with DH / 8 ADJACENT lanes x 16 B: one wave instruction covers
// 64 * 8 / DH whole rows (1 KB, every 128-B line touched once), eight such loads are in flight per wave, and the
// per-row dot product is closed with DPP adds inside the lane group.  (The first version read K with one lane per row -
// 64 rows x 16 B per instruction, 24 KB apart - and V four bytes per lane: 3.9 TB/s; profiles/r02v_attn_decode.txt.)
template <int DH>
__global__ __launch_bounds__(256) void attn_decode_batched_kernel(const bf16_t* __restrict__ q, int ldq,
                                                                  const bf16_t* __restrict__ kc, const bf16_t* __restrict__ vc,
                                                                  long sample_stride, const int32_t* __restrict__ lens,
                                                                  bf16_t* __restrict__ out, int ldo, int ld, float scale) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int LPR = DH / 8;            // lanes per row (16 B each)
  constexpr int RPI = 64 / LPR;          // rows per wave instruction
  constexpr int U = 8;                   // loads in flight per lane
  constexpr int RPP = 4 * RPI * U;       // rows per workgroup pass
  float* red = reinterpret_cast<float*>(smem);          // [8]
  float* part = red + 8;                                 // [4 * RPI][DH] partial outputs of every lane group
  float* sc = part + 4 * RPI * DH;                       // [S] scores / probabilities
  const int h = blockIdx.x, b = blockIdx.y, t = threadIdx.x, w = t >> 6, l = t & 63;
  const int c = l % LPR, rg = l / LPR;
  const int S = lens[b];
  if (S <= 0) {                          // no keys: defined output, no reads
    if (t < DH) out[(long)b * ldo + h * DH + t] = 0;
    return;
  }
  kc += (long)b * sample_stride + h * DH + c * 8;
  vc += (long)b * sample_stride + h * DH + c * 8;
  float qf[8];
  {
    const u32x4 qv = *reinterpret_cast<const u32x4*>(q + (long)b * ldq + h * DH + c * 8);
#pragma unroll
    for (int e = 0; e < 4; ++e) { qf[2 * e] = bf_lo(qv[e]) * scale; qf[2 * e + 1] = bf_hi(qv[e]) * scale; }
  }
  // scores: row = pass base + u * (4 * RPI) + w * RPI + rg
  float mx = -INFINITY;
  const int r0 = w * RPI + rg;
  for (int base = 0; base < S; base += RPP) {
    u32x4 kv[U];
#pragma unroll
    for (int u = 0; u < U; ++u) kv[u] = *reinterpret_cast<const u32x4*>(kc + (size_t)min(base + u * 4 * RPI + r0, S - 1) * ld);
#pragma unroll
    for (int u = 0; u < U; ++u) {
      float dot = 0.f;
#pragma unroll
      for (int e = 0; e < 4; ++e) dot += bf_lo(kv[u][e]) * qf[2 * e] + bf_hi(kv[u][e]) * qf[2 * e + 1];
      dot = group_sum<LPR>(dot);
      const int row = base + u * 4 * RPI + r0;
      if (row < S) {
        if (c == 0) sc[row] = dot;
        mx = fmaxf(mx, dot);
      }
    }
  }
  mx = block_max<4>(mx, red);            // (its barriers also publish sc[])
  float sum = 0.f;
  for (int s0 = t; s0 < S; s0 += 256) {
    const float p = __expf(sc[s0] - mx);
    sc[s0] = p;
    sum += p;
  }
  sum = block_sum<4>(sum, red + 4);
  __syncthreads();
  // P V: the lane keeps its 8 columns; rows as above
  float acc[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) acc[e] = 0.f;
  for (int base = 0; base < S; base += RPP) {
    u32x4 vv[U];
#pragma unroll
    for (int u = 0; u < U; ++u) vv[u] = *reinterpret_cast<const u32x4*>(vc + (size_t)min(base + u * 4 * RPI + r0, S - 1) * ld);
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int row = base + u * 4 * RPI + r0;
      const float p = row < S ? sc[row] : 0.f;
#pragma unroll
      for (int e = 0; e < 4; ++e) { acc[2 * e] += p * bf_lo(vv[u][e]); acc[2 * e + 1] += p * bf_hi(vv[u][e]); }
    }
  }
  {
    float* pp = part + (w * RPI + rg) * DH + c * 8;
    *reinterpret_cast<f32x4*>(pp) = f32x4{acc[0], acc[1], acc[2], acc[3]};
    *reinterpret_cast<f32x4*>(pp + 4) = f32x4{acc[4], acc[5], acc[6], acc[7]};
  }
  __syncthreads();
  if (t < DH) {
    float o = 0.f;
#pragma unroll
    for (int i = 0; i < 4 * RPI; ++i) o += part[i * DH + t];
    out[(long)b * ldo + h * DH + t] = f2bf(o / sum);
  }
}

extern "C" int hvlm_attn_decode_batched(const void* q, int ldq, const void* k_cache, const void* v_cache, long sample_stride,
                                        const int32_t* lens, void* out, int ldo, int B, int max_S, int H, int dh, int ld,
                                        float scale, void* stream) {
  HVLM_REQUIRE(q && k_cache && v_cache && lens && out, "attn_decode_batched: null pointer");
  HVLM_REQUIRE(B > 0 && max_S > 0 && H > 0 && ld >= H * dh && ldq >= H * dh && ldo >= H * dh, "attn_decode_batched: bad shape");
  HVLM_REQUIRE(ld % 8 == 0 && ldq % 8 == 0 && sample_stride % 8 == 0 && ((uintptr_t)q & 15) == 0 && ((uintptr_t)k_cache & 15) == 0 &&
               ((uintptr_t)v_cache & 15) == 0, "attn_decode_batched: rows must be 16-byte aligned (ld, ldq, sample_stride %% 8)");
  HVLM_REQUIRE(dh == 64 || dh == 128, "attn_decode_batched: dh must be 64 or 128 (got %d)", dh);
  const size_t lds = (size_t)(8 + 4 * (512 / dh) * dh + max_S) * 4;
  HVLM_REQUIRE(lds <= 64 * 1024, "attn_decode_batched: max_S=%d too long for the LDS score buffer", max_S);
  if (dh == 128)
    hipLaunchKernelGGL((attn_decode_batched_kernel<128>), dim3(H, B), dim3(256), lds, as_stream(stream), (const bf16_t*)q, ldq,
                       (const bf16_t*)k_cache, (const bf16_t*)v_cache, sample_stride, lens, (bf16_t*)out, ldo, ld, scale);
  else
    hipLaunchKernelGGL((attn_decode_batched_kernel<64>), dim3(H, B), dim3(256), lds, as_stream(stream), (const bf16_t*)q, ldq,
                       (const bf16_t*)k_cache, (const bf16_t*)v_cache, sample_stride, lens, (bf16_t*)out, ldo, ld, scale);
  return check_launch("attn_decode_batched");
}

extern "C" int hvlm_gemv_bf16(const void* W, const void* x, void* y, const void* residual, int N, int K, int ldw,
                              void* stream) {
  HVLM_REQUIRE(W && x && y, "gemv: null pointer");
  HVLM_REQUIRE(N > 0 && K > 0 && K % 8 == 0 && ldw >= K && ldw % 8 == 0, "gemv: bad shape (K %% 8, ldw %% 8)");
  HVLM_REQUIRE(K * 2 <= 64 * 1024, "gemv: K too large for the LDS copy of x");
  return gemv_pre(W, x, y, residual, N, K, ldw, 0, nullptr, 0.f, stream);
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
// 5 launches per layer on the caller's stream.  Pointer tables are HOST arrays of device pointers.  The workspace must be
// zero-filled once before its first use (combine tickets).
// pos >= 0: the position is a host value (eager mode).  pos < 0: the position is read on the device from pos_dev[0] by every
// kernel that needs it (the qkv GEMV's output row, RoPE, the attention length) and advanced by one at the end, so the
// launch sequence is identical for every token and can be captured once in a hipGraph and replayed (max_S bounds the
// attention's LDS score buffer and fixes the key split).
static int decode_step_impl(const void* const* w_qkv, const void* const* w_o, const void* const* w_gu,
                            const void* const* w_down, const void* const* ln1, const void* const* ln2,
                            void* const* kv_cache, const void* norm_w, const void* cos_tab, const void* sin_tab,
                            const int32_t* pos_dev, const void* x_in, void* hidden_out, void* ws, int L, int D,
                            int F, int H, int dh, int pos, int max_S, int cache_ld, float eps, void* stream) {
  HVLM_REQUIRE(w_qkv && w_o && w_gu && w_down && ln1 && ln2 && kv_cache && norm_w && x_in && hidden_out && ws && pos_dev,
               "decode_step: null pointer");
  HVLM_REQUIRE(cache_ld >= 3 * D && D == H * dh, "decode_step: bad shape");
  bf16_t* w = reinterpret_cast<bf16_t*>(ws);
  bf16_t* ao = w;            // [D]
  bf16_t* h = ao + D;        // [D]
  bf16_t* xa = h + D;        // [D]
  bf16_t* xb = xa + D;       // [D]
  bf16_t* gu = xb + D;       // [2F]
  float* apart = reinterpret_cast<float*>(gu + 2 * (size_t)F + 64);   // [H][NS][dh + 2] attention partials
  int* cnt = reinterpret_cast<int*>(apart + (size_t)128 * 8 * 130);    // [H] tickets of the fused combine (zero at rest)
  const bool fused = !config().decode_unfused;          // (hvlm_configure("decode_unfused", 1): A/B and the parity test)
  HVLM_REQUIRE(H <= 128, "decode_step: more than 128 heads");
  HVLM_REQUIRE(dh == 64 || dh == 128, "decode_step: dh must be 64 or 128");
  HVLM_REQUIRE(((size_t)F * 2) <= 64 * 1024 && (size_t)D * 2 <= 64 * 1024, "decode_step: D / F too large for the LDS copy of x");
  const float scale = 1.0f / sqrtf((float)dh);
  const bf16_t* x = reinterpret_cast<const bf16_t*>(x_in);
  const bool dev = pos < 0;
  const int S = dev ? max_S : pos + 1;
  HVLM_REQUIRE(S > 0, "decode_step: bad position / max_S");
  // key splits (H * NS workgroups).  Round 6: splits of <= 64 keys (at most 1024 / H of them: the partials' slots in the workspace) for the
  // lane-group kernel attn_decode_split_kernel, which holds a split of <= 128 keys in registers; longer splits (S > 128 * NS) and
  // hvlm_configure("decode_attn_v1", 1) run the round-3 kernel with its 1 / 4 / 8 splits
  int NS = S >= 512 ? 8 : (S >= 128 ? 4 : 1);
  bool v2 = !config().decode_attn_v1;
  if (v2) {
    // splits: 64-key splits, but no more workgroups than CUs when 128-key splits allow it (H = 32, S = 867: 8 splits = 256 workgroups, 2.87
    // against 2.92 ms per token with 14), and never more than the workspace's 1024 / H partial slots
    const int cap = config().decode_attn_ns > 0 ? config().decode_attn_ns : std::max(1, std::min(32, 1024 / H));
    const int ns2 = std::min(std::max(cdiv(S, 128), std::min(cdiv(S, 64), std::max(1, 256 / H))), cap);
    if (cdiv(S, ns2) <= 128 && (size_t)S * (size_t)cache_ld < (1ull << 31)) NS = ns2; else v2 = false;   // (its 32-bit row offsets)
  }
  const size_t lds_attn = (size_t)(dh + 8 + (256 / (dh / 2)) * dh + (S + NS - 1) / NS + 1 + 4) * 4 + (size_t)dh * 2;   // + the new key row
  HVLM_REQUIRE(lds_attn <= 64 * 1024, "decode_step: S=%d too long for the LDS score buffer", S);
  const int32_t* pd = dev ? pos_dev : nullptr;
  int rc;
  // ---- the persistent form (decode_stack_kernel): hvlm_configure("decode_engine", 1) --------------------------------------------------
  if (config().decode_engine && fused && v2) {
    static int cus[64];
    int devi = 0;
    if (hipGetDevice(&devi) == hipSuccess && devi >= 0 && devi < 64) {
      if (!cus[devi]) { hipDeviceProp_t prop; cus[devi] = hipGetDeviceProperties(&prop, devi) == hipSuccess ? prop.multiProcessorCount : 256; }
      const int G = cus[devi];
      const int rows_wg = std::max(3 * D, F) / G + 2;
      const size_t lds = ((size_t)std::max(D, F) + 128) * 2;
      if (L <= ENG_MAXL && H * NS <= G && D % 8 == 0 && F % 8 == 0 && D >= 8 && rows_wg <= 128 && lds <= 48 * 1024) {
        EngineArgs a;
        for (int l = 0; l < L; ++l) {
          a.wqkv[l] = (const bf16_t*)w_qkv[l]; a.wo[l] = (const bf16_t*)w_o[l]; a.wgu[l] = (const bf16_t*)w_gu[l];
          a.wdown[l] = (const bf16_t*)w_down[l]; a.ln1[l] = (const bf16_t*)ln1[l]; a.ln2[l] = (const bf16_t*)ln2[l];
          a.kv[l] = (bf16_t*)kv_cache[l];
        }
        a.cos_tab = (const bf16_t*)cos_tab; a.sin_tab = (const bf16_t*)sin_tab; a.pos_dev = pd; a.x_in = x;
        a.ao = ao; a.h = h; a.xa = xa; a.xb = xb; a.act = gu; a.apart = apart; a.cnt = cnt;
        a.bar = reinterpret_cast<unsigned*>(cnt + 128); a.err = reinterpret_cast<int*>(cnt + 128 + 16);
        a.L = L; a.D = D; a.F = F; a.H = H; a.pos = dev ? 0 : pos; a.cache_ld = cache_ld;
        a.ns_cap = config().decode_attn_ns > 0 ? config().decode_attn_ns : std::max(1, std::min(32, 1024 / H));
        a.eps = eps; a.scale = scale;
        if (hipMemsetAsync(a.bar, 0, 4, as_stream(stream)) != hipSuccess) { set_error("decode_engine: hipMemsetAsync failed"); return HVLM_ERR_LAUNCH; }
        if (dh == 128) hipLaunchKernelGGL((decode_stack_kernel<128>), dim3(G), dim3(ENG_NT), lds, as_stream(stream), a);
        else hipLaunchKernelGGL((decode_stack_kernel<64>), dim3(G), dim3(ENG_NT), lds, as_stream(stream), a);
        if ((rc = check_launch("decode_stack"))) return rc;
        const bf16_t* xl = (L & 1) ? xa : xb;                       // layer L - 1 wrote (L - 1) & 1 ? xb : xa
        if (L == 0) xl = x;
        return hvlm_rmsnorm_fwd(xl, norm_w, hidden_out, nullptr, 1, D, eps, stream);
      }
    }
  }
  // 5 launches per layer: the two RMSNorms and the SwiGLU ride in the prologue of the GEMV that consumes them; attention is
  // split over the keys (flash-decoding) with the new token's RoPE and the combine of the splits fused in
  for (int l = 0; l < L; ++l) {
    bf16_t* kv = reinterpret_cast<bf16_t*>(kv_cache[l]);
    bf16_t* row = dev ? kv : kv + (size_t)pos * cache_ld;
    if ((rc = gemv_pre(w_qkv[l], x, row, nullptr, 3 * D, D, D, 1, ln1[l], eps, stream, pd, cache_ld))) return rc;
    // (Round 6, measured and removed — profiles/r06j_decode_experiments.txt: touching o_proj's and the head of gate|up's weights from a side
    //  stream beside the attention, so that the GEMVs find them in the 256 MB memory-side cache: 3.18 / 3.28 / 3.43 ms per token at 32 / 64 /
    //  128 MB against 2.92 without — the touch kernel (8.3 us per launch) competes with the attention for the same queues, and a GEMV whose
    //  weights sit in the memory-side cache is no faster than one that streams them from HBM.)
    // RoPE of the new q / k row and the combine of the key splits are fused into the attention kernel (HVLM_DECODE_UNFUSED=1
    // keeps them as separate launches: A/B and the parity test)
    if (!fused) {
      if (dev) {
        hipLaunchKernelGGL(rope_decode_kernel, dim3(cdiv(2 * H * (dh / 16), 256)), dim3(256), 0, as_stream(stream), kv,
                           (const bf16_t*)cos_tab, (const bf16_t*)sin_tab, pos_dev, H, dh, cache_ld);
      } else if ((rc = hvlm_rope_qk(row, cos_tab, sin_tab, pos_dev, 1, H, dh, cache_ld, 0, stream))) {
        return rc;
      }
    }
    const bf16_t* fc = fused ? (const bf16_t*)cos_tab : nullptr;
    const bf16_t* fs = fused ? (const bf16_t*)sin_tab : nullptr;
    int* fcnt = fused ? cnt : nullptr;
    if (v2) {
      if (dh == 128)
        hipLaunchKernelGGL((attn_decode_split_kernel<128>), dim3(H, NS), dim3(256), 0, as_stream(stream), (const bf16_t*)row,
                           (const bf16_t*)(kv + D), (const bf16_t*)(kv + 2 * D), apart, S, cache_ld, scale, NS, pd, fc, fs, fcnt, ao);
      else
        hipLaunchKernelGGL((attn_decode_split_kernel<64>), dim3(H, NS), dim3(256), 0, as_stream(stream), (const bf16_t*)row,
                           (const bf16_t*)(kv + D), (const bf16_t*)(kv + 2 * D), apart, S, cache_ld, scale, NS, pd, fc, fs, fcnt, ao);
    } else if (dh == 128) {
      hipLaunchKernelGGL((attn_decode_partial_kernel<128>), dim3(H, NS), dim3(256), lds_attn, as_stream(stream), (const bf16_t*)row,
                         (const bf16_t*)(kv + D), (const bf16_t*)(kv + 2 * D), apart, S, cache_ld, scale, NS, pd, fc, fs, fcnt, ao);
    } else {
      hipLaunchKernelGGL((attn_decode_partial_kernel<64>), dim3(H, NS), dim3(256), lds_attn, as_stream(stream), (const bf16_t*)row,
                         (const bf16_t*)(kv + D), (const bf16_t*)(kv + 2 * D), apart, S, cache_ld, scale, NS, pd, fc, fs, fcnt, ao);
    }
    if (!fused) {
      if (dh == 128) hipLaunchKernelGGL((attn_decode_combine_kernel<128>), dim3(H), dim3(128), 0, as_stream(stream), (const float*)apart, ao, NS);
      else hipLaunchKernelGGL((attn_decode_combine_kernel<64>), dim3(H), dim3(64), 0, as_stream(stream), (const float*)apart, ao, NS);
    }
    if ((rc = check_launch("attn_decode_split"))) return rc;
    if ((rc = gemv_pre(w_o[l], ao, h, x, D, D, D, 0, nullptr, 0.f, stream))) return rc;
    bf16_t* xo = (l & 1) ? xb : xa;
    if (!config().decode_unfused) {
      // gate|up as (gate row, up row) pairs with the SwiGLU in the GEMV's epilogue: `gu` holds the F-element activation
      if ((rc = gemv_pre(w_gu[l], h, gu, nullptr, F, D, D, 1, ln2[l], eps, stream, nullptr, 0, F))) return rc;
      if ((rc = gemv_pre(w_down[l], gu, xo, h, D, F, F, 0, nullptr, 0.f, stream))) return rc;
    } else {
      if ((rc = gemv_pre(w_gu[l], h, gu, nullptr, 2 * F, D, D, 1, ln2[l], eps, stream))) return rc;
      if ((rc = gemv_pre(w_down[l], gu, xo, h, D, F, F, 2, nullptr, 0.f, stream))) return rc;
    }
    x = xo;
  }
  if ((rc = hvlm_rmsnorm_fwd(x, norm_w, hidden_out, nullptr, 1, D, eps, stream))) return rc;
  if (dev) {
    hipLaunchKernelGGL(pos_advance_kernel, dim3(1), dim3(1), 0, as_stream(stream), const_cast<int32_t*>(pos_dev));
    return check_launch("pos_advance");
  }
  return HVLM_OK;
}

extern "C" int hvlm_llama_decode_step(const void* const* w_qkv, const void* const* w_o, const void* const* w_gu,
                                      const void* const* w_down, const void* const* ln1, const void* const* ln2,
                                      void* const* kv_cache, const void* norm_w, const void* cos_tab, const void* sin_tab,
                                      const int32_t* pos_dev, const void* x_in, void* hidden_out, void* ws, int L, int D,
                                      int F, int H, int dh, int pos, int cache_ld, float eps, void* stream) {
  HVLM_REQUIRE(pos >= 0, "decode_step: negative position");
  return decode_step_impl(w_qkv, w_o, w_gu, w_down, ln1, ln2, kv_cache, norm_w, cos_tab, sin_tab, pos_dev, x_in, hidden_out, ws,
                          L, D, F, H, dh, pos, 0, cache_ld, eps, stream);
}

extern "C" int hvlm_llama_decode_step_dev(const void* const* w_qkv, const void* const* w_o, const void* const* w_gu,
                                          const void* const* w_down, const void* const* ln1, const void* const* ln2,
                                          void* const* kv_cache, const void* norm_w, const void* cos_tab,
                                          const void* sin_tab, int32_t* pos_dev, const void* x_in, void* hidden_out,
                                          void* ws, int L, int D, int F, int H, int dh, int max_S, int cache_ld, float eps,
                                          void* stream) {
  return decode_step_impl(w_qkv, w_o, w_gu, w_down, ln1, ln2, kv_cache, norm_w, cos_tab, sin_tab, pos_dev, x_in, hidden_out, ws,
                          L, D, F, H, dh, -1, max_S, cache_ld, eps, stream);
}

extern "C" size_t hvlm_llama_decode_ws_bytes(int D, int F) {
  // activations (4 D + 2 F bf16) + attention partials (up to 128 heads x 8 splits x 130 floats) + 128 combine tickets
  return (size_t)(4 * D + 2 * (size_t)F + 64) * 2 + (size_t)128 * 8 * 130 * 4 + 128 * 4 + 256;
}
