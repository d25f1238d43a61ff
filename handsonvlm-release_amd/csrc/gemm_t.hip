// bf16 GEMM with K-STRIDED ("transposed") operands, for the backward GEMMs without transposed copies:
//     C[M,N] = act(opA(A) @ opB(B)^T + bias) + residual
//     opA: A is [M,K] (K contiguous)  or, trans_a, [K,M] (M contiguous)
//     opB: B is [N,K] (K contiguous)  or, trans_b, [K,N] (N contiguous)
//   dX = dY @ W        -> trans_b  (W [N_out=K, K_in=N] used as stored: no resident W^T)
//   dW = dY^T @ X      -> trans_a + trans_b (dY [M_tok=K, N_out=M], X [M_tok=K, K_in=N] used as stored)
//
// Same skeleton as gemm.hip (persistent 256x256 tiles, 8 waves, 2-stage LDS ring, swapped-operand
// v_mfma_f32_32x32x16_bf16, fused epilogue); what differs:
//   * a K-strided operand tile is staged as [64 k][256 x] (512-B rows) and its MFMA fragments are built with
//     ds_read_b64_tr_b16 (LDS transpose read: a 16-lane group reads a 4(k) x 16(x) block, lane q receives
//     column q — semantics probed on hardware, tools/probe_tr16.hip).  The 64-B segment index of a row is
//     XOR-ed with (k & 3) so the four k-rows of one read land in four different bank quarters;
//   * all operand tiles are fetched with `buffer_load_dwordx4 ... lds` through a buffer descriptor, so rows /
//     k-rows past the end of the matrix read as ZERO: K need not be a multiple of 64 when both operands are
//     K-strided (K = number of tokens), and no zero-padded copies are needed.
#include "common.h"
#include "gemm_args.h"
#include <stdlib.h>

namespace hvlm {
namespace {

constexpr int BM = 256, BN = 256, BK = 64;
constexpr int TILE_BYTES = BM * BK * 2;
constexpr int STAGE_BYTES = 2 * TILE_BYTES;
constexpr int LDS_BYTES = 2 * STAGE_BYTES;
constexpr int GROUP_M = 4;

typedef __attribute__((ext_vector_type(4))) short s16x4;
typedef __attribute__((address_space(3))) void* lds_ptr_t;

struct GemmTArgs {
  const bf16_t* A;
  const bf16_t* B;
  void* C;
  const bf16_t* bias;
  const bf16_t* res;
  unsigned a_bytes, b_bytes;
  int M, N, K, lda, ldb, ldc, ldr, act, out_f32, tiles_m, tiles_n;
};

__device__ __forceinline__ float quick_gelu_t(float x) { return x / (1.f + __expf(-1.702f * x)); }

template <bool TA, bool TB>
__global__ __launch_bounds__(512, 2) void gemm_bf16_t_kernel(GemmTArgs p) {
  constexpr int FM = 4, FN = 2, WN = 4;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x;
  const int l = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l31 = l & 31, hi = l >> 5;
  const int wm = w / WN, wn = w % WN;

  const int nwg = p.tiles_m * p.tiles_n;
  auto tile_coords = [&](int vb, int& tm_, int& tn_) {
    const int q = nwg >> 3, r = nwg & 7, xcd = vb & 7, idx = vb >> 3;
    const int swz = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    const int gsz = GROUP_M * p.tiles_n;
    const int gid = swz / gsz;
    const int first_m = gid * GROUP_M;
    const int gm = min(p.tiles_m - first_m, GROUP_M);
    const int in = swz - gid * gsz;
    tm_ = first_m + in % gm;
    tn_ = in / gm;
  };
  int tm, tn;
  tile_coords(blockIdx.x, tm, tn);
  int m0 = tm * BM, n0 = tn * BN;

  const auto rsA = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(p.A), 0, p.a_bytes, 0x00020000);
  const auto rsB = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(p.B), 0, p.b_bytes, 0x00020000);

  // per-lane byte offsets of the 4 DMA pieces of each operand (K-tile 0); the K advance goes in soffset
  unsigned va[4], vb_[4];
  auto set_offsets = [&](int m0_, int n0_) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int pi = w * 4 + i;
      if constexpr (TA) {
        const int kr = pi * 2 + (l >> 5);
        const int c = (l & 31) ^ ((kr & 3) << 2);
        va[i] = (unsigned)(((long)kr * p.lda + m0_ + c * 8) * 2);
      } else {
        const int row = pi * 8 + (l >> 3);
        const int c = (l & 7) ^ ((row >> 1) & 7);
        va[i] = (unsigned)(((long)(m0_ + row) * p.lda + c * 8) * 2);
      }
      if constexpr (TB) {
        const int kr = pi * 2 + (l >> 5);
        const int c = (l & 31) ^ ((kr & 3) << 2);
        vb_[i] = (unsigned)(((long)kr * p.ldb + n0_ + c * 8) * 2);
      } else {
        const int row = pi * 8 + (l >> 3);
        const int c = (l & 7) ^ ((row >> 1) & 7);
        vb_[i] = (unsigned)(((long)(n0_ + row) * p.ldb + c * 8) * 2);
      }
    }
  };
  set_offsets(m0, n0);
  const unsigned dma_off = (unsigned)(w * 4) * 1024u;
  const unsigned ka_step = TA ? (unsigned)p.lda * (BK * 2u) : (BK * 2u);   // bytes per K-tile
  const unsigned kb_step = TB ? (unsigned)p.ldb * (BK * 2u) : (BK * 2u);

  auto stage_part = [&](int buf, int kt, int part) {
    const unsigned base = (unsigned)buf * STAGE_BYTES + dma_off + part * 1024u;
    // the K advance is added to the VGPR offset: only voffset takes part in the descriptor's range check
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsA, (lds_ptr_t)(smem + base), 16, va[part] + (unsigned)kt * ka_step, 0, 0, 0);
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsB, (lds_ptr_t)(smem + base + TILE_BYTES), 16,
                                             vb_[part] + (unsigned)kt * kb_step, 0, 0, 0);
  };

  // ---- fragment addressing ----------------------------------------------------------------------
  // normal tile: [256 rows][64 k], 128-B rows, chunk ^ ((row>>1)&7)
  const int sw = (l31 >> 1) & 7;
  unsigned koff[4];
#pragma unroll
  for (int ks = 0; ks < 4; ++ks) koff[ks] = (unsigned)(((ks * 2 + hi) ^ sw) << 4);
  const unsigned a_base_n = (unsigned)(wm * 128 + l31) * 128u;
  const unsigned b_base_n = (unsigned)(wn * 64 + l31) * 128u;
  // K-strided tile: [64 k][256 x], 512-B rows, 16-B chunk ^ ((k&3)<<2); transpose-read per 16-lane group
  const int p16 = l & 15, g16 = (l >> 4) & 1;
  const unsigned t_row = (unsigned)(hi * 8 + (p16 >> 2)) * 512u;
  const int t_swz = ((p16 >> 2) & 3) << 2;
  unsigned ta_off[FM], tb_off[FN];
#pragma unroll
  for (int f = 0; f < FM; ++f) {
    const int chunk = ((wm * 128 + f * 32 + g16 * 16 + (p16 & 3) * 4) >> 3);
    ta_off[f] = t_row + (unsigned)((chunk ^ t_swz) << 4) + (unsigned)((p16 & 1) * 8);
  }
#pragma unroll
  for (int f = 0; f < FN; ++f) {
    const int chunk = ((wn * 64 + f * 32 + g16 * 16 + (p16 & 3) * 4) >> 3);
    tb_off[f] = t_row + (unsigned)((chunk ^ t_swz) << 4) + (unsigned)((p16 & 1) * 8);
  }
  auto tr_frag = [&](unsigned off) -> bf16x8 {
    const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(smem + off));
    const s16x4 hi4 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(smem + off + 4 * 512));
    typedef __attribute__((ext_vector_type(8))) short s16x8;
    const s16x8 v = {lo[0], lo[1], lo[2], lo[3], hi4[0], hi4[1], hi4[2], hi4[3]};
    return __builtin_bit_cast(bf16x8, v);
  };
  auto load_frags = [&](unsigned sb, int ks, bf16x8 (&af)[FM], bf16x8 (&bfr)[FN]) {
#pragma unroll
    for (int i = 0; i < FM; ++i) {
      if constexpr (TA) af[i] = tr_frag(sb + ta_off[i] + (unsigned)ks * (16 * 512u));
      else af[i] = *reinterpret_cast<const bf16x8*>(smem + sb + a_base_n + i * 4096u + koff[ks]);
    }
#pragma unroll
    for (int j = 0; j < FN; ++j) {
      if constexpr (TB) bfr[j] = tr_frag(sb + TILE_BYTES + tb_off[j] + (unsigned)ks * (16 * 512u));
      else bfr[j] = *reinterpret_cast<const bf16x8*>(smem + sb + TILE_BYTES + b_base_n + j * 4096u + koff[ks]);
    }
  };

  f32x16 acc[FN][FM];

  auto epilogue = [&](int m0, int n0) {
    const int ncol0 = n0 + wn * 64;
    const bool fast = !p.out_f32 && ncol0 + 64 <= p.N;
#pragma unroll
    for (int i = 0; i < FM; ++i) {
      const int m = m0 + wm * 128 + i * 32 + l31;
      if (m >= p.M) continue;
#pragma unroll
      for (int j = 0; j < FN; ++j)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const int nb = ncol0 + j * 32 + 8 * g + 4 * hi;
          if (nb >= p.N) continue;
          float v[4];
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] = acc[j][i][4 * g + e];
          const bool full = fast || (nb + 3 < p.N);
          if (p.bias) {
            for (int e = 0; e < 4; ++e) if (nb + e < p.N) v[e] += bf2f(p.bias[nb + e]);
          }
          if (p.act == HVLM_ACT_QUICK_GELU) {
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = quick_gelu_t(v[e]);
          }
          if (p.res) {
            const bf16_t* rp = p.res + (size_t)m * p.ldr + nb;
            for (int e = 0; e < 4; ++e) if (nb + e < p.N) v[e] += bf2f(rp[e]);
          }
          if (p.out_f32) {
            float* cp = reinterpret_cast<float*>(p.C) + (size_t)m * p.ldc + nb;
            if (full) { f32x4 o = {v[0], v[1], v[2], v[3]}; *reinterpret_cast<f32x4*>(cp) = o; }
            else { for (int e = 0; e < 4; ++e) if (nb + e < p.N) cp[e] = v[e]; }
          } else {
            bf16_t* cp = reinterpret_cast<bf16_t*>(p.C) + (size_t)m * p.ldc + nb;
            if (full) { u32x2 o = {pack_bf2(v[0], v[1]), pack_bf2(v[2], v[3])}; *reinterpret_cast<u32x2*>(cp) = o; }
            else { for (int e = 0; e < 4; ++e) if (nb + e < p.N) cp[e] = f2bf(v[e]); }
          }
        }
    }
  };

  const int nk = (p.K + BK - 1) / BK;
  int vb = blockIdx.x;
  unsigned g = 0;
#pragma unroll
  for (int part = 0; part < 4; ++part) stage_part(0, 0, part);
  bf16x8 afA[FM], bfA[FN], afB[FM], bfB[FN];
  while (true) {
#pragma unroll
    for (int j = 0; j < FN; ++j)
#pragma unroll
      for (int i = 0; i < FM; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[j][i][r] = 0.f;
    const int cm0 = m0, cn0 = n0;
    const int nvb = vb + (int)gridDim.x;
    const bool has_next = nvb < nwg;
    for (int kt = 0; kt < nk; ++kt, ++g) {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
      const unsigned sb = (g & 1u) * STAGE_BYTES;
      const bool last = kt + 1 == nk;
      if (last && has_next) {
        tile_coords(nvb, tm, tn);
        m0 = tm * BM; n0 = tn * BN;
        set_offsets(m0, n0);
      }
      const bool more = !last || has_next;
      const int nkt = last ? 0 : kt + 1;
      load_frags(sb, 0, afA, bfA);
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) {
        if (more) stage_part((int)((g + 1) & 1u), nkt, ks);
        if (ks < 3) {
          if (ks & 1) load_frags(sb, ks + 1, afA, bfA); else load_frags(sb, ks + 1, afB, bfB);
        }
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int j = 0; j < FN; ++j)
#pragma unroll
          for (int i = 0; i < FM; ++i) {
            if (ks & 1) acc[j][i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bfB[j], afB[i], acc[j][i], 0, 0, 0);
            else acc[j][i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bfA[j], afA[i], acc[j][i], 0, 0, 0);
          }
        __builtin_amdgcn_s_setprio(0);
      }
    }
    epilogue(cm0, cn0);
    if (!has_next) break;
    vb = nvb;
  }
}

template <bool TA, bool TB>
static int launch(const GemmTArgs& p, int grid, hipStream_t st) {
  static bool done = false;
  if (!done) {
    hipError_t e = hipFuncSetAttribute((const void*)gemm_bf16_t_kernel<TA, TB>,
                                       hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
    if (e != hipSuccess) { set_error("gemm_t: hipFuncSetAttribute: %s", hipGetErrorString(e)); return HVLM_ERR_LAUNCH; }
    done = true;
  }
  hipLaunchKernelGGL((gemm_bf16_t_kernel<TA, TB>), dim3(grid), dim3(512), LDS_BYTES, st, p);
  return check_launch("gemm_bf16");
}

}  // namespace
}  // namespace hvlm

using namespace hvlm;

extern "C" int hvlm_gemm_bf16(const void* A, const void* B, void* C, int M, int N, int K, int lda, int ldb, int ldc,
                              const void* bias, const void* residual, int ldr, int act, int out_f32, int trans_a,
                              int trans_b, void* stream) {
  if (!trans_a && !trans_b)
    return hvlm_gemm_bf16_nt(A, B, C, M, N, K, lda, ldb, ldc, bias, residual, ldr, act, out_f32, stream);
  HVLM_REQUIRE(A && B && C, "gemm: null operand");
  HVLM_REQUIRE(M > 0 && N > 0 && K > 0, "gemm: bad shape M=%d N=%d K=%d", M, N, K);
  HVLM_REQUIRE((trans_a && trans_b) || K % BK == 0, "gemm: K=%d must be a multiple of %d unless both operands are K-strided", K, BK);
  HVLM_REQUIRE(lda >= (trans_a ? M : K) && ldb >= (trans_b ? N : K) && ldc >= N, "gemm: leading dimension too small");
  HVLM_REQUIRE(lda % 8 == 0 && ldb % 8 == 0 && ldc % 4 == 0, "gemm: lda/ldb %% 8, ldc %% 4 required");
  HVLM_REQUIRE(((uintptr_t)A & 15) == 0 && ((uintptr_t)B & 15) == 0 && ((uintptr_t)C & 15) == 0, "gemm: 16-byte alignment");
  HVLM_REQUIRE(act == HVLM_ACT_NONE || act == HVLM_ACT_QUICK_GELU, "gemm: unknown act %d", act);
  const size_t ab = (size_t)(trans_a ? K : M) * lda * 2, bb = (size_t)(trans_b ? K : N) * ldb * 2;
  HVLM_REQUIRE(ab < (1ull << 32) && bb < (1ull << 32), "gemm: K-strided operands are limited to 4 GiB (buffer addressing)");
  GemmTArgs p;
  p.A = (const bf16_t*)A; p.B = (const bf16_t*)B; p.C = C; p.bias = (const bf16_t*)bias; p.res = (const bf16_t*)residual;
  p.a_bytes = (unsigned)ab; p.b_bytes = (unsigned)bb;
  p.M = M; p.N = N; p.K = K; p.lda = lda; p.ldb = ldb; p.ldc = ldc; p.ldr = ldr; p.act = act; p.out_f32 = out_f32;
  p.tiles_m = cdiv(M, BM); p.tiles_n = cdiv(N, BN);
  static int pgrid = 0;
  if (!pgrid) {
    int dev = 0; hipDeviceProp_t prop;
    pgrid = (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess) ? prop.multiProcessorCount : 256;
    pgrid = (pgrid / 8) * 8;
  }
  int grid = p.tiles_m * p.tiles_n;
  if (grid > pgrid) grid = pgrid;
  hipStream_t st = as_stream(stream);
  if (trans_a && trans_b) {
    // both operands K-strided (dW = dY^T X): the 8-phase ping-pong kernel with transpose reads (gemm_pp.hip)
    GemmArgs q;
    q.A = p.A; q.B = p.B; q.C = C; q.bias = p.bias; q.res = p.res;
    q.M = M; q.N = N; q.K = K; q.lda = lda; q.ldb = ldb; q.ldc = ldc; q.ldr = ldr; q.act = act; q.out_f32 = out_f32;
    q.tiles_m = p.tiles_m; q.tiles_n = p.tiles_n; q.group_m = 4;
    q.splits = 1; q.k_split = K; q.slab = 0; q.alpha = 1.f; q.dyn = 0; q.dbg = 0; q.epi_swap = config().gemm_epi_swap; q.tail_half = 0; q.sched = pp_sched_ws();
    q.dyn = ((K + BK - 1) / BK >= 3 && pgrid <= 512 && q.tiles_m * q.tiles_n >= 3 * pgrid && !config().gemm_static) ? 1 + pp_sched_slot(st) : 0;
    const int rc = launch_gemm_pp_tn(q, grid, st);
    if (rc) return rc;
    return check_launch("gemm_pp_tn");
  }
  if (trans_a) return launch<true, false>(p, grid, st);
  return launch<false, true>(p, grid, st);
}
