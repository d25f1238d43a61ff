// Cross-entropy over bf16 logits (one workgroup per row, online max/sum in fp32) and the fused
// AdamW step (fp32 master / moments, bf16 gradient in, bf16 weight + transposed weight out).
// Reference: handsonvlm.py:137-145 (CrossEntropyLoss, mean over labels != -100) and
// scripts/finetune.sh:30-33 (AdamW, lr 2e-5, weight decay 0) via HF Trainer.
#include "common.h"

namespace hvlm {
namespace {

constexpr int CE_T = 256;

// online (max, sum) merge
__device__ __forceinline__ void ms_merge(float& m, float& s, float m2, float s2) {
  const float mn = fmaxf(m, m2);
  if (mn == -INFINITY) { s = 0.f; return; }   // both empty: avoid (-inf) - (-inf)
  s = s * __expf(m - mn) + s2 * __expf(m2 - mn);
  m = mn;
}

__global__ __launch_bounds__(CE_T) void ce_fwd_kernel(const bf16_t* __restrict__ logits,
                                                      const int32_t* __restrict__ targets,
                                                      float* __restrict__ loss_sum,
                                                      float* __restrict__ row_lse, int M, int V,
                                                      int ldl) {
  __shared__ float red_m[4], red_s[4];
  const int row = blockIdx.x;
  const int t = threadIdx.x;
  const int tgt = targets[row];
  if (tgt < 0) {  // ignored row (IGNORE_INDEX = -100): no loss, lse unused
    if (row_lse && t == 0) row_lse[row] = 0.f;
    return;
  }
  const bf16_t* lr = logits + (size_t)row * ldl;
  float m = -INFINITY, s = 0.f;
  const int V8 = V & ~7;
  for (int i = t * 8; i < V8; i += CE_T * 8) {
    const u32x4 v = *reinterpret_cast<const u32x4*>(lr + i);
    float x[8];
#pragma unroll
    for (int e = 0; e < 4; ++e) { x[2 * e] = bf_lo(v[e]); x[2 * e + 1] = bf_hi(v[e]); }
    float mx = x[0];
#pragma unroll
    for (int e = 1; e < 8; ++e) mx = fmaxf(mx, x[e]);
    const float mn = fmaxf(m, mx);
    float acc = (m == -INFINITY) ? 0.f : s * __expf(m - mn);
#pragma unroll
    for (int e = 0; e < 8; ++e) acc += __expf(x[e] - mn);
    s = acc; m = mn;
  }
  if (t < V - V8) {
    const float x = bf2f(lr[V8 + t]);
    ms_merge(m, s, x, 1.f);
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    const float m2 = __shfl_xor(m, o, 64), s2 = __shfl_xor(s, o, 64);
    ms_merge(m, s, m2, s2);
  }
  const int w = t >> 6;
  if ((t & 63) == 0) { red_m[w] = m; red_s[w] = s; }
  __syncthreads();
  if (t == 0) {
    float mm = red_m[0], ss = red_s[0];
    for (int i = 1; i < CE_T / 64; ++i) ms_merge(mm, ss, red_m[i], red_s[i]);
    const float lse = mm + logf(ss);
    if (row_lse) {
      row_lse[row] = lse;          // the loss is summed from these in a fixed order by ce_loss_sum_kernel (deterministic)
    } else {
      atomicAdd(loss_sum, lse - bf2f(lr[tgt]));
      atomicAdd(loss_sum + 1, 1.f);
    }
  }
}

// loss_sum[0] += sum over valid rows of (lse - logit[target]), loss_sum[1] += their count — ONE workgroup, fixed order:
// the reported loss is run-to-run bit-identical (fp32 atomics from 13 872 row workgroups are not).
__global__ __launch_bounds__(1024) void ce_loss_sum_kernel(const bf16_t* __restrict__ logits,
                                                           const int32_t* __restrict__ targets,
                                                           const float* __restrict__ row_lse,
                                                           float* __restrict__ loss_sum, int M, int ldl) {
  __shared__ float red[16];
  float s = 0.f, n = 0.f;
  for (int r = threadIdx.x; r < M; r += 1024) {
    const int tgt = targets[r];
    if (tgt >= 0) { s += row_lse[r] - bf2f(logits[(size_t)r * ldl + tgt]); n += 1.f; }
  }
  s = block_sum<16>(s, red);
  n = block_sum<16>(n, red);
  if (threadIdx.x == 0) { loss_sum[0] += s; loss_sum[1] += n; }
}

__global__ __launch_bounds__(CE_T) void ce_bwd_kernel(const bf16_t* __restrict__ logits,
                                                      const int32_t* __restrict__ targets,
                                                      const float* __restrict__ row_lse,
                                                      const float* __restrict__ grad_scale,
                                                      bf16_t* __restrict__ dlogits, int M, int V,
                                                      int ldl) {
  const int row = blockIdx.x;
  const int t = threadIdx.x;
  const int tgt = targets[row];
  const bf16_t* lr = logits + (size_t)row * ldl;
  bf16_t* dr = dlogits + (size_t)row * ldl;
  const float gs = grad_scale[0];
  const float lse = tgt >= 0 ? row_lse[row] : 0.f;
  for (int i = t * 8; i < ldl; i += CE_T * 8) {
    u32x4 o = {0u, 0u, 0u, 0u};
    if (tgt >= 0) {
      const u32x4 v = *reinterpret_cast<const u32x4*>(lr + i);
      float g[8];
#pragma unroll
      for (int e = 0; e < 4; ++e) { g[2 * e] = bf_lo(v[e]); g[2 * e + 1] = bf_hi(v[e]); }
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const int col = i + e;
        float p = (col < V) ? __expf(g[e] - lse) : 0.f;
        if (col == tgt) p -= 1.f;
        g[e] = p * gs;
      }
#pragma unroll
      for (int e = 0; e < 4; ++e) o[e] = pack_bf2(g[2 * e], g[2 * e + 1]);
    }
    *reinterpret_cast<u32x4*>(dr + i) = o;
  }
}

// ---- AdamW, 64x64 tiles (so the transposed bf16 copy can be written with 16-byte stores) -------
struct AdamArgs {
  float lr, beta1, beta2, eps, wd, gscale, bc1, bc2;
};

__global__ __launch_bounds__(256) void adamw_kernel(float* __restrict__ master, float* __restrict__ m,
                                                    float* __restrict__ v,
                                                    const bf16_t* __restrict__ grad,
                                                    bf16_t* __restrict__ w, bf16_t* __restrict__ wt,
                                                    int R, int C, AdamArgs a) {
  __shared__ bf16_t tile[64][66];
  const int r0 = blockIdx.y * 64, c0 = blockIdx.x * 64;
  const int t = threadIdx.x;
  const bool vec_ok = (C % 4) == 0;
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const int idx = t + k * 256;       // 1024 items of 4 elements
    const int r = idx >> 4, c = (idx & 15) * 4;
    bf16_t outv[4] = {0, 0, 0, 0};
    if (r0 + r < R) {
      const int cc0 = c0 + c;
      const size_t off0 = (size_t)(r0 + r) * C + cc0;
      if (vec_ok && cc0 + 3 < C) {
        // 16-byte fp32 accesses, 8-byte bf16 accesses
        const u32x2 gv = *reinterpret_cast<const u32x2*>(grad + off0);
        f32x4 pv = *reinterpret_cast<const f32x4*>(master + off0);
        f32x4 mv = *reinterpret_cast<const f32x4*>(m + off0);
        f32x4 vv = *reinterpret_cast<const f32x4*>(v + off0);
        const float g4[4] = {bf_lo(gv[0]) * a.gscale, bf_hi(gv[0]) * a.gscale, bf_lo(gv[1]) * a.gscale, bf_hi(gv[1]) * a.gscale};
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          float p = pv[e] * (1.f - a.lr * a.wd);
          const float mm = a.beta1 * mv[e] + (1.f - a.beta1) * g4[e];
          const float v2 = a.beta2 * vv[e] + (1.f - a.beta2) * g4[e] * g4[e];
          const float denom = sqrtf(v2) / a.bc2 + a.eps;
          p -= (a.lr / a.bc1) * (mm / denom);
          pv[e] = p; mv[e] = mm; vv[e] = v2;
          outv[e] = f2bf(p);
        }
        *reinterpret_cast<f32x4*>(master + off0) = pv;
        *reinterpret_cast<f32x4*>(m + off0) = mv;
        *reinterpret_cast<f32x4*>(v + off0) = vv;
        u32x2 wv = {outv[0] | ((unsigned)outv[1] << 16), outv[2] | ((unsigned)outv[3] << 16)};
        *reinterpret_cast<u32x2*>(w + off0) = wv;
      } else {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const int cc = cc0 + e;
          if (cc < C) {
            const size_t off = off0 + e;
            const float g = bf2f(grad[off]) * a.gscale;
            float p = master[off];
            p *= (1.f - a.lr * a.wd);
            const float mm = a.beta1 * m[off] + (1.f - a.beta1) * g;
            const float vv = a.beta2 * v[off] + (1.f - a.beta2) * g * g;
            const float denom = sqrtf(vv) / a.bc2 + a.eps;   // bc2 = sqrt(1 - beta2^t)
            p -= (a.lr / a.bc1) * (mm / denom);
            master[off] = p; m[off] = mm; v[off] = vv;
            outv[e] = f2bf(p);
            w[off] = outv[e];
          }
        }
      }
    }
    if (wt) {
#pragma unroll
      for (int e = 0; e < 4; ++e) tile[r][c + e] = outv[e];
    }
  }
  if (!wt) return;
  __syncthreads();
#pragma unroll
  for (int k = 0; k < 2; ++k) {
    const int idx = t + k * 256;
    const int c = idx >> 3, r = (idx & 7) * 8;
    if (c0 + c >= C) continue;
    bf16_t* op = wt + (size_t)(c0 + c) * R + r0 + r;
    if (r0 + r + 7 < R && (R % 8) == 0) {
      bf16_t e[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) e[j] = tile[r + j][c];
      u32x4 o = {e[0] | ((unsigned)e[1] << 16), e[2] | ((unsigned)e[3] << 16),
                 e[4] | ((unsigned)e[5] << 16), e[6] | ((unsigned)e[7] << 16)};
      *reinterpret_cast<u32x4*>(op) = o;
    } else {
      for (int j = 0; j < 8; ++j) if (r0 + r + j < R) op[j] = tile[r + j][c];
    }
  }
}

}  // namespace
}  // namespace hvlm

using namespace hvlm;

extern "C" int hvlm_ce_fwd(const void* logits, const int32_t* targets, float* loss_sum, float* row_lse,
                           int M, int V, int ldl, void* stream) {
  HVLM_REQUIRE(logits && targets && loss_sum, "ce_fwd: null pointer");
  HVLM_REQUIRE(M > 0 && V > 0 && ldl >= V && ldl % 8 == 0, "ce_fwd: bad shape (ldl %% 8)");
  hipLaunchKernelGGL(ce_fwd_kernel, dim3(M), dim3(CE_T), 0, as_stream(stream), (const bf16_t*)logits,
                     targets, loss_sum, row_lse, M, V, ldl);
  if (row_lse)
    hipLaunchKernelGGL(ce_loss_sum_kernel, dim3(1), dim3(1024), 0, as_stream(stream), (const bf16_t*)logits, targets,
                       (const float*)row_lse, loss_sum, M, ldl);
  return check_launch("ce_fwd");
}

extern "C" int hvlm_ce_bwd(const void* logits, const int32_t* targets, const float* row_lse,
                           const float* grad_scale, void* dlogits, int M, int V, int ldl, void* stream) {
  HVLM_REQUIRE(logits && targets && row_lse && grad_scale && dlogits, "ce_bwd: null pointer");
  HVLM_REQUIRE(M > 0 && V > 0 && ldl >= V && ldl % 8 == 0, "ce_bwd: bad shape (ldl %% 8)");
  hipLaunchKernelGGL(ce_bwd_kernel, dim3(M), dim3(CE_T), 0, as_stream(stream), (const bf16_t*)logits,
                     targets, row_lse, grad_scale, (bf16_t*)dlogits, M, V, ldl);
  return check_launch("ce_bwd");
}

extern "C" int hvlm_adamw_step(float* master, float* m, float* v, const void* grad, void* w, void* wt,
                               int R, int C, float lr, float beta1, float beta2, float eps,
                               float weight_decay, float grad_scale, int step, void* stream) {
  HVLM_REQUIRE(master && m && v && grad && w, "adamw: null pointer");
  HVLM_REQUIRE(R > 0 && C > 0 && step >= 1, "adamw: bad args");
  AdamArgs a;
  a.lr = lr; a.beta1 = beta1; a.beta2 = beta2; a.eps = eps; a.wd = weight_decay; a.gscale = grad_scale;
  a.bc1 = 1.f - powf(beta1, (float)step);
  a.bc2 = sqrtf(1.f - powf(beta2, (float)step));
  dim3 grid(cdiv(C, 64), cdiv(R, 64));
  hipLaunchKernelGGL(adamw_kernel, grid, dim3(256), 0, as_stream(stream), master, m, v,
                     (const bf16_t*)grad, (bf16_t*)w, (bf16_t*)wt, R, C, a);
  return check_launch("adamw_step");
}
