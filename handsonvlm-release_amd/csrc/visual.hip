// Visual-token pipeline kernels (all HBM-bound): CLIP patch im2col + embedding assembly,
// temporal/spatial pooling, multimodal splice with the <hand_traj> sinusoidal encoding, and the
// hand-token gather.  Reference: hoi_forecast/model/visual_to_tokens.py:252-284 and
// handsonvlm/model/language_model/handsonvlm.py:146-187,212-451.
#include "common.h"

namespace hvlm {
namespace {

// patches[(n*G*G + gy*G + gx), k] = images[n, c, gy*P+py, gx*P+px], k = c*P*P + py*P + px
__global__ __launch_bounds__(256) void im2col_kernel(const bf16_t* __restrict__ img,
                                                     bf16_t* __restrict__ out, int N, int HW, int P,
                                                     int Kpad) {
  const int G = HW / P;
  const int K = 3 * P * P;
  const long total = (long)N * G * G * Kpad;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
    const int k = (int)(i % Kpad);
    const long row = i / Kpad;
    bf16_t v = 0;
    if (k < K) {
      const int gx = (int)(row % G);
      const int gy = (int)((row / G) % G);
      const long n = row / ((long)G * G);
      const int c = k / (P * P);
      const int rem = k - c * P * P;
      const int py = rem / P, px = rem - py * P;
      v = img[((n * 3 + c) * HW + (gy * P + py)) * (long)HW + gx * P + px];
    }
    out[i] = v;
  }
}

// The same gather, one workgroup per (frame, patch row): the P image rows of the strip x 3 channels are read ROW-CONTIGUOUS into LDS
// (two pixels per lane and load: whole 448-B image rows at HW = 224) and leave as the G patch rows of `out` (Kpad contiguous
// elements each) — both sides of the copy coalesced.  The element-wise kernel above (2 bytes and five integer divisions per
// thread) took 1.07 ms for the SFT step's 1600 frames (1 GB of traffic: ~1 TB/s).  TO_F16: the frames are bf16 and the patches
// leave as IEEE half — images.to(dtype=self.dtype) of clip_encoder.py:48 applied on the way through (bit-identical to casting the
// frames first: the same conversion of the same values), which also removes the tower's separate cast pass over the frames.
// Needs P and HW even (pairs of pixels), LDS = 3 * P * HW * 2 bytes (18.4 KiB at 224 px, 27.6 KiB at 336 px).
template <bool TO_F16>
__global__ __launch_bounds__(256) void im2col_strip_kernel(const bf16_t* __restrict__ img, bf16_t* __restrict__ out, int HW, int P,
                                                           int Kpad) {
  extern __shared__ __attribute__((aligned(16))) char strip_raw[];
  unsigned* strip = reinterpret_cast<unsigned*>(strip_raw);          // [3 * P][HW / 2] pixel pairs
  const int G = HW / P, K = 3 * P * P, HW2 = HW / 2;
  const int gy = (int)blockIdx.x % G;
  const long n = (long)blockIdx.x / G;
  const int nload = 3 * P * HW2;
  for (int i = threadIdx.x; i < nload; i += 256) {
    const int x2 = i % HW2, rp = i / HW2;                             // rp = c * P + py
    const int c = rp / P, py = rp - c * P;
    unsigned v = *reinterpret_cast<const unsigned*>(img + ((n * 3 + c) * HW + (gy * P + py)) * (long)HW + 2 * x2);
    if (TO_F16) v = pack_h2(bf_lo(v), bf_hi(v));
    strip[i] = v;
  }
  __syncthreads();
  const int K2 = Kpad / 2, PP = P * P;
  const int nstore = G * K2;
  unsigned* orow = reinterpret_cast<unsigned*>(out + (n * G * G + (long)gy * G) * Kpad);
  for (int i = threadIdx.x; i < nstore; i += 256) {
    const int k = 2 * (i % K2), gx = i / K2;
    unsigned v = 0u;
    if (k < K) {                                                      // (K even: a pair never straddles the padding)
      const int c = k / PP, rem = k - c * PP;
      const int py = rem / P, px = rem - py * P;                      // px even (P even, k even): px + 1 is in the same patch row
      v = strip[(c * P + py) * HW2 + (gx * P + px) / 2];
    }
    orow[(long)gx * K2 + (k >> 1)] = v;
  }
}

// tokens[n, 0] = cls + pos[0]; tokens[n, 1+p] = patch_out[n*NP+p] + pos[1+p]
template <bool F16>
__global__ __launch_bounds__(256) void clip_embed_kernel(const bf16_t* __restrict__ patch_out,
                                                         const bf16_t* __restrict__ cls,
                                                         const bf16_t* __restrict__ pos,
                                                         bf16_t* __restrict__ tokens, long N, int NP,
                                                         int D) {
  const int dv = D / 8;
  const long total = N * (NP + 1) * dv;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
    const int c = (int)(i % dv) * 8;
    const long tokrow = i / dv;
    const int t = (int)(tokrow % (NP + 1));
    const long n = tokrow / (NP + 1);
    const bf16_t* src = (t == 0) ? cls + c : patch_out + (n * NP + (t - 1)) * (long)D + c;
    const u32x4 a = *reinterpret_cast<const u32x4*>(src);
    const u32x4 b = *reinterpret_cast<const u32x4*>(pos + (long)t * D + c);
    u32x4 o;
#pragma unroll
    for (int e = 0; e < 4; ++e) o[e] = E16<F16>::pack2(E16<F16>::lo(a[e]) + E16<F16>::lo(b[e]), E16<F16>::hi(a[e]) + E16<F16>::hi(b[e]));
    *reinterpret_cast<u32x4*>(tokens + tokrow * (long)D + c) = o;
  }
}

__device__ __forceinline__ int sel_frame(int k, int T) {
  // np.round(np.linspace(0, T-1, 4)) : round-half-to-even on k*(T-1)/3
  return (int)rintf((float)k * (float)(T - 1) / 3.0f);
}

// One work item = 8 channels of one output row.  grid-strided.  G = patches per side (16 at the reference's 224 px,
// 24 at 336 px), G2 = G*G, H = G/2, P2 = H*H pooled cells per selected frame.
//   out row r < T     : mean over the G2 patches of frame r
//   out row r >= T    : j = r-T; frame sel(j/P2); (oy,ox) = ((j%P2)/H, j%H); mean of the 2x2 patches
// F16: the tokens / the output are IEEE half (the CLIP tower's own dtype: pooling INSIDE the tower's last layer, see
// hvlm/clip.py::_last_layer_pooled) instead of bfloat16; accumulation is fp32 either way.
// fidx (optional, int32 [B*T]): frame (b, t) of the pooled layout is token block fidx[b*T + t] of `tok` — the tower ran on the
// DISTINCT frames only (config.dedup_frames) and the pooling gathers them back.
template <bool F16>
__global__ __launch_bounds__(256) void pool_fwd_kernel(const bf16_t* __restrict__ tok,
                                                       bf16_t* __restrict__ out, int B, int T, int D, int G,
                                                       int skip /*1 if cls is dropped*/, const int32_t* __restrict__ fidx) {
  using EL = E16<F16>;
  const int dv = D / 8;
  const int G2 = G * G, H = G / 2, P2 = H * H;
  const int tpf = G2 + skip;
  const int rows = T + 4 * P2;
  const long total = (long)B * rows * dv;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
    const int c = (int)(i % dv) * 8;
    const long rr = i / dv;
    const int r = (int)(rr % rows);
    const long b = rr / rows;
    float acc[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) acc[e] = 0.f;
    float scale;
    if (r < T) {
      const long fr = fidx ? (long)fidx[b * T + r] : b * T + r;
      const bf16_t* base = tok + (fr * (long)tpf + skip) * D + c;
      for (int p = 0; p < G2; ++p) {
        const u32x4 v = *reinterpret_cast<const u32x4*>(base + (long)p * D);
#pragma unroll
        for (int e = 0; e < 4; ++e) { acc[2 * e] += EL::lo(v[e]); acc[2 * e + 1] += EL::hi(v[e]); }
      }
      scale = 1.f / (float)G2;
    } else {
      const int j = r - T;
      const int f = sel_frame(j / P2, T);
      const int jj = j % P2;
      const int oy = jj / H, ox = jj % H;
      const long fr = fidx ? (long)fidx[b * T + f] : b * T + f;
      const bf16_t* base = tok + (fr * (long)tpf + skip) * D + c;
#pragma unroll
      for (int dy = 0; dy < 2; ++dy)
#pragma unroll
        for (int dx = 0; dx < 2; ++dx) {
          const int p = (2 * oy + dy) * G + 2 * ox + dx;
          const u32x4 v = *reinterpret_cast<const u32x4*>(base + (long)p * D);
#pragma unroll
          for (int e = 0; e < 4; ++e) { acc[2 * e] += EL::lo(v[e]); acc[2 * e + 1] += EL::hi(v[e]); }
        }
      scale = 0.25f;
    }
    u32x4 o;
#pragma unroll
    for (int e = 0; e < 4; ++e) o[e] = EL::pack2(acc[2 * e] * scale, acc[2 * e + 1] * scale);
    *reinterpret_cast<u32x4*>(out + rr * (long)D + c) = o;
  }
}

__global__ __launch_bounds__(256) void pool_bwd_kernel(const bf16_t* __restrict__ dout,
                                                       bf16_t* __restrict__ dtok, int B, int T, int D, int G) {
  const int dv = D / 8;
  const int G2 = G * G, H = G / 2, P2 = H * H;
  const int rows = T + 4 * P2;
  const float inv = 1.f / (float)G2;
  const long total = (long)B * T * G2 * dv;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
    const int c = (int)(i % dv) * 8;
    const long rr = i / dv;
    const int p = (int)(rr % G2);
    const long bt = rr / G2;
    const int t = (int)(bt % T);
    const long b = bt / T;
    float acc[8];
    {
      const u32x4 v = *reinterpret_cast<const u32x4*>(dout + (b * rows + t) * (long)D + c);
#pragma unroll
      for (int e = 0; e < 4; ++e) { acc[2 * e] = bf_lo(v[e]) * inv; acc[2 * e + 1] = bf_hi(v[e]) * inv; }
    }
    const int py = p / G, px = p % G;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      if (sel_frame(k, T) == t) {
        const int j = k * P2 + (py >> 1) * H + (px >> 1);
        const u32x4 v = *reinterpret_cast<const u32x4*>(dout + (b * rows + T + j) * (long)D + c);
#pragma unroll
        for (int e = 0; e < 4; ++e) { acc[2 * e] += bf_lo(v[e]) * 0.25f; acc[2 * e + 1] += bf_hi(v[e]) * 0.25f; }
      }
    }
    u32x4 o;
#pragma unroll
    for (int e = 0; e < 4; ++e) o[e] = pack_bf2(acc[2 * e], acc[2 * e + 1]);
    *reinterpret_cast<u32x4*>(dtok + rr * (long)D + c) = o;
  }
}

// ---- multimodal splice + <hand_traj> positional encoding -------------------------------------
// channel ch of a hand row: hand = ch & 1, cc = ch >> 1 in [0, D/2); quarter q = cc / (D/8),
// i = cc % (D/8); inv_freq_i = 10000^(-(2i)/(D/4)); q even: sin(x*f) ; q odd: cos(y*f).
__global__ __launch_bounds__(256) void splice_fwd_kernel(
    const bf16_t* __restrict__ embed, const bf16_t* __restrict__ visual,
    const int32_t* __restrict__ src, const int32_t* __restrict__ hand_slot,
    const float* __restrict__ hands, const float* __restrict__ hand_scale, bf16_t* __restrict__ out,
    int B, int S, int D, int NV, int NH, const int32_t* __restrict__ row_sample) {
  const int dv = D / 8;
  const long total = (long)B * S * dv;
  const int D8 = D / 8;
  const float lg = -logf(10000.f) * 2.f / (float)(D / 4);
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
    const int c = (int)(i % dv) * 8;
    const long row = i / dv;
    const long b = row_sample ? (long)row_sample[row] : row / S;      // packed rows carry their sample index
    const int sidx = src[row];
    u32x4 v = {0u, 0u, 0u, 0u};
    if (sidx >= 0) v = *reinterpret_cast<const u32x4*>(embed + (long)sidx * D + c);
    else if (sidx <= -2) v = *reinterpret_cast<const u32x4*>(visual + (b * NV + (-2 - sidx)) * (long)D + c);
    const int slot = hand_slot ? hand_slot[row] : -1;
    if (slot >= 0) {
      const float sc = hand_scale[b];
      float r[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const int ch = c + e;
        const int hand = ch & 1, cc = ch >> 1;
        const int q = cc / D8, fi = cc - q * D8;
        const float* hp = hands + ((b * 2 + hand) * NH + slot) * 2L;
        const float f = __expf(lg * (float)fi);
        r[e] = (q & 1) ? cosf(hp[1] * f) : sinf(hp[0] * f);
        r[e] *= sc;
      }
#pragma unroll
      for (int e = 0; e < 4; ++e) v[e] = pack_bf2(bf_lo(v[e]) + r[2 * e], bf_hi(v[e]) + r[2 * e + 1]);
    }
    *reinterpret_cast<u32x4*>(out + row * (long)D + c) = v;
  }
}

__global__ __launch_bounds__(256) void splice_bwd_kernel(const bf16_t* __restrict__ dx,
                                                         const int32_t* __restrict__ src,
                                                         float* __restrict__ dembed,
                                                         bf16_t* __restrict__ dvisual, int B, int S,
                                                         int D, int NV, const int32_t* __restrict__ row_sample) {
  const int dv = D / 8;
  const long total = (long)B * S * dv;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
    const int c = (int)(i % dv) * 8;
    const long row = i / dv;
    const long b = row_sample ? (long)row_sample[row] : row / S;
    const int sidx = src[row];
    if (sidx == -1) continue;
    const u32x4 v = *reinterpret_cast<const u32x4*>(dx + row * (long)D + c);
    if (sidx >= 0) {
      if (dembed) {
        float* dp = dembed + (long)sidx * D + c;
#pragma unroll
        for (int e = 0; e < 4; ++e) { atomicAdd(dp + 2 * e, bf_lo(v[e])); atomicAdd(dp + 2 * e + 1, bf_hi(v[e])); }
      }
    } else if (dvisual) {
      *reinterpret_cast<u32x4*>(dvisual + (b * NV + (-2 - sidx)) * (long)D + c) = v;
    }
  }
}

// out[b, hand, j, c] = hidden[rows[b*4+j], 2c+hand]
template <typename OutT>
__global__ __launch_bounds__(256) void hand_gather_fwd_kernel(const bf16_t* __restrict__ hidden,
                                                              const int32_t* __restrict__ rows,
                                                              OutT* __restrict__ out, int B, int D) {
  const int half = D / 2;
  const long total = (long)B * 2 * 4 * half;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
    const int c = (int)(i % half);
    const int j = (int)((i / half) & 3);
    const int hand = (int)((i / (half * 4L)) & 1);
    const long b = i / (half * 8L);
    const int row = rows[b * 4 + j];
    const float v = row >= 0 ? bf2f(hidden[(long)row * D + 2 * c + hand]) : 0.f;
    if constexpr (sizeof(OutT) == 4) out[i] = v; else out[i] = f2bf(v);
  }
}

template <typename InT>
__global__ __launch_bounds__(256) void hand_gather_bwd_kernel(const InT* __restrict__ dout,
                                                              const int32_t* __restrict__ rows,
                                                              bf16_t* __restrict__ dhidden, int B, int D) {
  const int half = D / 2;
  const long total = (long)B * 2 * 4 * half;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
    const int c = (int)(i % half);
    const int j = (int)((i / half) & 3);
    const int hand = (int)((i / (half * 4L)) & 1);
    const long b = i / (half * 8L);
    const int row = rows[b * 4 + j];
    if (row < 0) continue;
    float g;
    if constexpr (sizeof(InT) == 4) g = dout[i]; else g = bf2f(dout[i]);
    bf16_t* p = dhidden + (long)row * D + 2 * c + hand;
    *p = f2bf(bf2f(*p) + g);
  }
}

static inline int grid_for(long items) {
  long g = (items + 255) / 256;
  return (int)(g < 1 ? 1 : (g > 4096 ? 4096 : g));
}

}  // namespace
}  // namespace hvlm

using namespace hvlm;

template <bool TO_F16>
static int im2col_strip(const void* images, void* patches, int N, int HW, int P, int Kpad, void* stream, const char* what) {
  const int lds = 3 * P * HW * 2;
  static bool attr_done = false;
  if (!attr_done && lds > 48 * 1024) {
    if (hipFuncSetAttribute((const void*)im2col_strip_kernel<TO_F16>, hipFuncAttributeMaxDynamicSharedMemorySize, lds) != hipSuccess) {
      set_error("%s: hipFuncSetAttribute failed", what);
      return HVLM_ERR_LAUNCH;
    }
    attr_done = true;
  }
  hipLaunchKernelGGL(im2col_strip_kernel<TO_F16>, dim3((unsigned)((long)N * (HW / P))), dim3(256), lds, as_stream(stream),
                     (const bf16_t*)images, (bf16_t*)patches, HW, P, Kpad);
  return check_launch(what);
}

extern "C" int hvlm_patch_im2col(const void* images, void* patches, int N, int HW, int P, int Kpad,
                                 void* stream) {
  HVLM_REQUIRE(images && patches && N > 0 && HW % P == 0 && Kpad >= 3 * P * P, "im2col: bad args");
  if (P % 2 == 0 && HW % 2 == 0 && Kpad % 2 == 0 && 3 * P * HW * 2 <= 64 * 1024)      // the coalesced strip kernel
    return im2col_strip<false>(images, patches, N, HW, P, Kpad, stream, "patch_im2col");
  const long total = (long)N * (HW / P) * (HW / P) * Kpad;
  hipLaunchKernelGGL(im2col_kernel, dim3(grid_for(total)), dim3(256), 0, as_stream(stream),
                     (const bf16_t*)images, (bf16_t*)patches, N, HW, P, Kpad);
  return check_launch("patch_im2col");
}

extern "C" int hvlm_patch_im2col_f16(const void* images, void* patches, int N, int HW, int P, int Kpad, void* stream) {
  HVLM_REQUIRE(images && patches && N > 0 && HW % P == 0 && Kpad >= 3 * P * P, "im2col_f16: bad args");
  HVLM_REQUIRE(P % 2 == 0 && HW % 2 == 0 && Kpad % 2 == 0 && 3 * P * HW * 2 <= 64 * 1024,
               "im2col_f16: needs even P / HW / Kpad and a strip of at most 64 KiB (P = %d, HW = %d)", P, HW);
  return im2col_strip<true>(images, patches, N, HW, P, Kpad, stream, "patch_im2col_f16");
}

extern "C" int hvlm_clip_embed(const void* patch_out, const void* cls, const void* pos, void* tokens,
                               int N, int NP, int D, void* stream) {
  HVLM_REQUIRE(patch_out && cls && pos && tokens && N > 0 && NP > 0 && D % 8 == 0, "clip_embed: bad args");
  const long total = (long)N * (NP + 1) * (D / 8);
  hipLaunchKernelGGL(clip_embed_kernel<false>, dim3(grid_for(total)), dim3(256), 0, as_stream(stream),
                     (const bf16_t*)patch_out, (const bf16_t*)cls, (const bf16_t*)pos, (bf16_t*)tokens,
                     (long)N, NP, D);
  return check_launch("clip_embed");
}

extern "C" int hvlm_clip_embed_f16(const void* patch_out, const void* cls, const void* pos, void* tokens,
                                   int N, int NP, int D, void* stream) {
  HVLM_REQUIRE(patch_out && cls && pos && tokens && N > 0 && NP > 0 && D % 8 == 0, "clip_embed_f16: bad args");
  const long total = (long)N * (NP + 1) * (D / 8);
  hipLaunchKernelGGL(clip_embed_kernel<true>, dim3(grid_for(total)), dim3(256), 0, as_stream(stream),
                     (const bf16_t*)patch_out, (const bf16_t*)cls, (const bf16_t*)pos, (bf16_t*)tokens,
                     (long)N, NP, D);
  return check_launch("clip_embed_f16");
}

extern "C" int hvlm_temporal_spatial_pool_fwd(const void* tokens, void* out, int B, int T, int D,
                                              int drop_cls, int grid, void* stream) {
  HVLM_REQUIRE(tokens && out && B > 0 && T >= 1 && D % 8 == 0, "pool_fwd: bad args");
  HVLM_REQUIRE(grid >= 2 && grid % 2 == 0, "pool_fwd: grid=%d must be even (16 = reference geometry)", grid);
  const long total = (long)B * (T + grid * grid) * (D / 8);
  hipLaunchKernelGGL(pool_fwd_kernel<false>, dim3(grid_for(total)), dim3(256), 0, as_stream(stream),
                     (const bf16_t*)tokens, (bf16_t*)out, B, T, D, grid, drop_cls ? 1 : 0, (const int32_t*)nullptr);
  return check_launch("temporal_spatial_pool_fwd");
}

extern "C" int hvlm_temporal_spatial_pool_fwd_f16(const void* tokens, void* out, int B, int T, int D,
                                                  int drop_cls, int grid, const int32_t* frame_index, void* stream) {
  HVLM_REQUIRE(tokens && out && B > 0 && T >= 1 && D % 8 == 0, "pool_fwd_f16: bad args");
  HVLM_REQUIRE(grid >= 2 && grid % 2 == 0, "pool_fwd_f16: grid=%d must be even (16 = reference geometry)", grid);
  const long total = (long)B * (T + grid * grid) * (D / 8);
  hipLaunchKernelGGL(pool_fwd_kernel<true>, dim3(grid_for(total)), dim3(256), 0, as_stream(stream),
                     (const bf16_t*)tokens, (bf16_t*)out, B, T, D, grid, drop_cls ? 1 : 0, frame_index);
  return check_launch("temporal_spatial_pool_fwd_f16");
}

extern "C" int hvlm_temporal_spatial_pool_bwd(const void* dout, void* dtokens, int B, int T, int D,
                                              int grid, void* stream) {
  HVLM_REQUIRE(dout && dtokens && B > 0 && T >= 1 && D % 8 == 0, "pool_bwd: bad args");
  HVLM_REQUIRE(grid >= 2 && grid % 2 == 0, "pool_bwd: grid=%d must be even", grid);
  const long total = (long)B * T * grid * grid * (D / 8);
  hipLaunchKernelGGL(pool_bwd_kernel, dim3(grid_for(total)), dim3(256), 0, as_stream(stream),
                     (const bf16_t*)dout, (bf16_t*)dtokens, B, T, D, grid);
  return check_launch("temporal_spatial_pool_bwd");
}

extern "C" int hvlm_embed_splice_fwd(const void* embed, const void* visual, const int32_t* src,
                                     const int32_t* hand_slot, const float* hands,
                                     const float* hand_scale, void* out, int B, int S, int D, int NV,
                                     int NH, void* stream) {
  HVLM_REQUIRE(embed && src && out && B > 0 && S > 0 && D % 8 == 0, "splice_fwd: bad args");
  HVLM_REQUIRE(!hand_slot || (hands && hand_scale && NH > 0), "splice_fwd: hand_slot needs hands/scale");
  const long total = (long)B * S * (D / 8);
  hipLaunchKernelGGL(splice_fwd_kernel, dim3(grid_for(total)), dim3(256), 0, as_stream(stream),
                     (const bf16_t*)embed, (const bf16_t*)visual, src, hand_slot, hands, hand_scale,
                     (bf16_t*)out, B, S, D, NV, NH, (const int32_t*)nullptr);
  return check_launch("embed_splice_fwd");
}

extern "C" int hvlm_embed_splice_fwd_packed(const void* embed, const void* visual, const int32_t* src,
                                            const int32_t* hand_slot, const float* hands, const float* hand_scale,
                                            const int32_t* row_sample, void* out, int rows, int D, int NV, int NH,
                                            void* stream) {
  HVLM_REQUIRE(embed && src && row_sample && out && rows > 0 && D % 8 == 0, "splice_fwd_packed: bad args");
  const long total = (long)rows * (D / 8);
  hipLaunchKernelGGL(splice_fwd_kernel, dim3(grid_for(total)), dim3(256), 0, as_stream(stream),
                     (const bf16_t*)embed, (const bf16_t*)visual, src, hand_slot, hands, hand_scale,
                     (bf16_t*)out, 1, rows, D, NV, NH, row_sample);
  return check_launch("embed_splice_fwd_packed");
}

extern "C" int hvlm_embed_splice_bwd_packed(const void* dx, const int32_t* src, const int32_t* row_sample, float* dembed,
                                            void* dvisual, int rows, int D, int NV, void* stream) {
  HVLM_REQUIRE(dx && src && row_sample && rows > 0 && D % 8 == 0, "splice_bwd_packed: bad args");
  const long total = (long)rows * (D / 8);
  hipLaunchKernelGGL(splice_bwd_kernel, dim3(grid_for(total)), dim3(256), 0, as_stream(stream),
                     (const bf16_t*)dx, src, dembed, (bf16_t*)dvisual, 1, rows, D, NV, row_sample);
  return check_launch("embed_splice_bwd_packed");
}

extern "C" int hvlm_embed_splice_bwd(const void* dx, const int32_t* src, float* dembed, void* dvisual,
                                     int B, int S, int D, int NV, void* stream) {
  HVLM_REQUIRE(dx && src && B > 0 && S > 0 && D % 8 == 0, "splice_bwd: bad args");
  const long total = (long)B * S * (D / 8);
  hipLaunchKernelGGL(splice_bwd_kernel, dim3(grid_for(total)), dim3(256), 0, as_stream(stream),
                     (const bf16_t*)dx, src, dembed, (bf16_t*)dvisual, B, S, D, NV, (const int32_t*)nullptr);
  return check_launch("embed_splice_bwd");
}

namespace hvlm {
namespace {
// Deterministic embedding gradient: one (token, 8-channel chunk) per thread, rows of the segment summed in the fixed,
// host-sorted order (fp32), written once as bf16.  No atomics: run-to-run bit-identical.
__global__ __launch_bounds__(256) void embed_grad_segments_kernel(const bf16_t* __restrict__ dx,
                                                                  const int32_t* __restrict__ seg_rows,
                                                                  const int32_t* __restrict__ seg_start,
                                                                  const int32_t* __restrict__ seg_token,
                                                                  bf16_t* __restrict__ dembed, int n_seg, int D) {
  const int dv = D / 8;
  const long total = (long)n_seg * dv;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
    const int c = (int)(i % dv) * 8;
    const int sg = (int)(i / dv);
    const int r0 = seg_start[sg], r1 = seg_start[sg + 1];
    float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    for (int r = r0; r < r1; ++r) {
      const u32x4 v = *reinterpret_cast<const u32x4*>(dx + (long)seg_rows[r] * D + c);
#pragma unroll
      for (int e = 0; e < 4; ++e) { acc[2 * e] += bf_lo(v[e]); acc[2 * e + 1] += bf_hi(v[e]); }
    }
    u32x4 o;
#pragma unroll
    for (int e = 0; e < 4; ++e) o[e] = pack_bf2(acc[2 * e], acc[2 * e + 1]);
    *reinterpret_cast<u32x4*>(dembed + (long)seg_token[sg] * D + c) = o;
  }
}

}  // namespace
}  // namespace hvlm

extern "C" int hvlm_embed_grad_segments(const void* dx, const int32_t* seg_rows, const int32_t* seg_start,
                                        const int32_t* seg_token, void* dembed, int n_seg, int D, void* stream) {
  HVLM_REQUIRE(dx && seg_rows && seg_start && seg_token && dembed && n_seg >= 0 && D % 8 == 0, "embed_grad_segments: bad args");
  if (n_seg == 0) return HVLM_OK;
  const long total = (long)n_seg * (D / 8);
  hipLaunchKernelGGL(embed_grad_segments_kernel, dim3(cdiv(total, 256)), dim3(256), 0, as_stream(stream),
                     (const bf16_t*)dx, seg_rows, seg_start, seg_token, (bf16_t*)dembed, n_seg, D);
  return check_launch("embed_grad_segments");
}

extern "C" int hvlm_hand_gather_fwd(const void* hidden, const int32_t* rows, void* out, int B, int D,
                                    int out_f32, void* stream) {
  HVLM_REQUIRE(hidden && rows && out && B > 0 && D % 2 == 0, "hand_gather_fwd: bad args");
  const long total = (long)B * 4 * D;
  if (out_f32)
    hipLaunchKernelGGL((hand_gather_fwd_kernel<float>), dim3(grid_for(total)), dim3(256), 0,
                       as_stream(stream), (const bf16_t*)hidden, rows, (float*)out, B, D);
  else
    hipLaunchKernelGGL((hand_gather_fwd_kernel<bf16_t>), dim3(grid_for(total)), dim3(256), 0,
                       as_stream(stream), (const bf16_t*)hidden, rows, (bf16_t*)out, B, D);
  return check_launch("hand_gather_fwd");
}

extern "C" int hvlm_hand_gather_bwd(const void* dout, const int32_t* rows, void* dhidden, int B, int D,
                                    int dout_f32, void* stream) {
  HVLM_REQUIRE(dout && rows && dhidden && B > 0 && D % 2 == 0, "hand_gather_bwd: bad args");
  const long total = (long)B * 4 * D;
  if (dout_f32)
    hipLaunchKernelGGL((hand_gather_bwd_kernel<float>), dim3(grid_for(total)), dim3(256), 0,
                       as_stream(stream), (const float*)dout, rows, (bf16_t*)dhidden, B, D);
  else
    hipLaunchKernelGGL((hand_gather_bwd_kernel<bf16_t>), dim3(grid_for(total)), dim3(256), 0,
                       as_stream(stream), (const bf16_t*)dout, rows, (bf16_t*)dhidden, B, D);
  return check_launch("hand_gather_bwd");
}
