// bf16 GEMM on MFMA for gfx950:  C[M,N] = act(A[M,K] @ B[N,K]^T + bias) + residual.
//
// Structure (256x256 output tiles, 8 waves as 2(M) x 4(N), wave tile 128x64):
//   * persistent workgroups (one per CU): each walks output tiles in an XCD-aware order and runs ONE
//     flat K-tile pipeline across tile boundaries — the LDS-DMA of the next tile's first K-tile is
//     issued during the last K-tile of the current one, so the epilogue overlaps loads in flight;
//   * K is walked in 64-wide tiles; both operand tiles (256 rows x 128 B) are copied HBM -> LDS by
//     LDS-DMA (`global_load_lds_dwordx4`, 1 KiB per wave-instruction, no VGPR round trip) into a
//     2-deep ring; the 8 DMA pieces per wave are spread over the 4 k-steps of the running tile;
//   * fragments are register double-buffered: the ds_read_b128 of k-step s+1 fly under the MFMAs of s;
//   * LDS image: row-major 128-B rows, 16-B chunk c of row r stored at chunk position
//     c ^ ((r>>1)&7).  LDS-DMA writes lane-linear, so the permutation is applied to the per-lane
//     SOURCE address and again on the fragment read; with it every ds_read_b128 lane group of a
//     32-row fragment touches 16 distinct 16-B slots of the 256-B bank row (SQ_LDS_BANK_CONFLICT = 0);
//   * v_mfma_f32_32x32x16_bf16 with the operands SWAPPED (weight-tile rows as the MFMA "A"):
//     the accumulator then holds, per lane, one output row m and 4 consecutive output columns per
//     register quad, so the epilogue (bias, QuickGELU, residual, bf16 pack) stores 8 contiguous
//     bytes per lane without an LDS transpose;
//   * tile order is XCD-aware (block b runs on XCD b%8): each XCD gets a contiguous range of a
//     grouped (GROUP_M row-panels x all column-panels) ordering so the tiles resident on one XCD
//     share operand panels in that XCD's private L2.
// Other template variants (plain 2-stage loop, 16-wave layout, register staging, ablations) are kept
// for A/B measurements (tools/gemm_bench.py, HVLM_GEMM_VARIANT); numbers in DESIGN.md.
//
// Reference arithmetic replaced: every nn.Linear on the path (see include/hvlm.h).
#include "common.h"
#include "gemm_args.h"
#include <stdlib.h>

namespace hvlm {

namespace {

constexpr int BM = 256, BN = 256, BK = 64;
constexpr int TILE_BYTES = BM * BK * 2;       // 32 KiB per operand tile
constexpr int STAGE_BYTES = 2 * TILE_BYTES;   // A + B
constexpr int LDS_BYTES = 2 * STAGE_BYTES;    // 128 KiB, 2-deep ring
constexpr int GROUP_M = 4;

typedef __attribute__((address_space(3))) void* lds_ptr_t;
typedef const __attribute__((address_space(1))) void* gbl_ptr_t;

__device__ __forceinline__ void glds16(const bf16_t* g, unsigned lds_off, char* smem) {
  __builtin_amdgcn_global_load_lds((gbl_ptr_t)g, (lds_ptr_t)(smem + lds_off), 16, 0, 0);
}

__device__ __forceinline__ float quick_gelu(float x) { return x / (1.f + __expf(-1.702f * x)); }

template <int VARIANT, int WM, int WN>
__global__ __launch_bounds__(WM * WN * 64, (WM * WN) >= 8 ? (WM * WN) / 4 : 1) void gemm_bf16_nt_kernel(GemmArgs p) {
  constexpr int NW = WM * WN;            // waves per workgroup
  constexpr int FM = BM / WM / 32;       // 32-row A fragments per wave
  constexpr int FN = BN / WN / 32;       // 32-col B fragments per wave
  constexpr int PPW = 32 / NW;           // LDS-DMA pieces per wave per operand tile (32 pieces of 8 rows)
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x;
  const int l = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l31 = l & 31, hi = l >> 5;

  // ---- XCD-aware, grouped tile mapping ---------------------------------------------------
  const int ntile = p.tiles_m * p.tiles_n;
  const int nwg = ntile * p.splits;
  int ksl = 0;                           // K-slice of the tile being set up (split-K)
  auto tile_coords = [&](int vb, int& tm_, int& tn_) {
    const int q = nwg >> 3, r = nwg & 7, xcd = vb & 7, idx = vb >> 3;
    int swz = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    ksl = swz / ntile;
    swz -= ksl * ntile;
    const int gsz = p.group_m * p.tiles_n;
    const int gid = swz / gsz;
    const int first_m = gid * p.group_m;
    const int gm = min(p.tiles_m - first_m, p.group_m);
    const int in = swz - gid * gsz;
    tm_ = first_m + in % gm;
    tn_ = in / gm;
  };
  int tm, tn;
  tile_coords(blockIdx.x, tm, tn);
  int m0 = tm * BM, n0 = tn * BN;

  // ---- per-lane LDS-DMA source pointers (4 A pieces + 4 B pieces per wave per K-tile) ----
  const bf16_t* ga[PPW];
  const bf16_t* gb[PPW];
  auto set_pointers = [&](int m0_, int n0_) {
#pragma unroll
    for (int i = 0; i < PPW; ++i) {
      const int row = (w * PPW + i) * 8 + (l >> 3);        // tile row this lane feeds
      const int c = (l & 7) ^ ((row >> 1) & 7);            // source chunk for chunk position l&7
      const int ra = min(m0_ + row, p.M - 1);
      const int rb = min(n0_ + row, p.N - 1);
      ga[i] = p.A + (size_t)ra * p.lda + c * 8 + ksl * p.k_split;
      gb[i] = p.B + (size_t)rb * p.ldb + c * 8 + ksl * p.k_split;
    }
  };
  set_pointers(m0, n0);
  const unsigned dma_off = (unsigned)(w * PPW) * 1024u;  // + i*1024 ; wave-uniform

  auto stage = [&](int buf, int kt) {
    const unsigned base = (unsigned)buf * STAGE_BYTES + dma_off;
    const int k0 = kt * BK;
#pragma unroll
    for (int i = 0; i < PPW; ++i) glds16(ga[i] + k0, base + i * 1024u, smem);
#pragma unroll
    for (int i = 0; i < PPW; ++i) glds16(gb[i] + k0, base + TILE_BYTES + i * 1024u, smem);
  };

  // ---- fragment read offsets ----------------------------------------------------------------
  const int wm = w / WN, wn = w % WN;
  const int sw = (l31 >> 1) & 7;
  unsigned koff[4];
#pragma unroll
  for (int ks = 0; ks < 4; ++ks) koff[ks] = (unsigned)(((ks * 2 + hi) ^ sw) << 4);
  const unsigned a_base = (unsigned)(wm * (FM * 32) + l31) * 128u;
  const unsigned b_base = (unsigned)TILE_BYTES + (unsigned)(wn * (FN * 32) + l31) * 128u;

  f32x16 acc[FN][FM];
#pragma unroll
  for (int j = 0; j < FN; ++j)
#pragma unroll
    for (int i = 0; i < FM; ++i)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[j][i][r] = 0.f;

  // ---- epilogue: lane holds row m, 4 consecutive columns per register quad ----------------
  size_t c_slab = 0;                     // fp32 slab offset of the tile being written (split-K)
  auto epilogue_generic = [&](int m0, int n0) {
  #pragma unroll
    for (int i = 0; i < FM; ++i) {
      const int m = m0 + wm * (FM * 32) + i * 32 + l31;
      if (m >= p.M) continue;
  #pragma unroll
      for (int j = 0; j < FN; ++j) {
  #pragma unroll
        for (int g = 0; g < 4; ++g) {
          const int nb = n0 + wn * (FN * 32) + j * 32 + 8 * g + 4 * hi;
          if (nb >= p.N) continue;
          float v[4];
  #pragma unroll
          for (int e = 0; e < 4; ++e) v[e] = acc[j][i][4 * g + e];
          const bool full = (nb + 3 < p.N);
          if (p.bias) {
            if (full) {
              const u32x2 bb = *reinterpret_cast<const u32x2*>(p.bias + nb);
              v[0] += bf_lo(bb[0]); v[1] += bf_hi(bb[0]); v[2] += bf_lo(bb[1]); v[3] += bf_hi(bb[1]);
            } else {
              for (int e = 0; e < 4; ++e) if (nb + e < p.N) v[e] += bf2f(p.bias[nb + e]);
            }
          }
          if (p.act == HVLM_ACT_QUICK_GELU) {
  #pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = quick_gelu(v[e]);
          }
          if (p.res) {
            const bf16_t* rp = p.res + (size_t)m * p.ldr + nb;
            if (full) {
              const u32x2 rr = *reinterpret_cast<const u32x2*>(rp);
              v[0] += bf_lo(rr[0]); v[1] += bf_hi(rr[0]); v[2] += bf_lo(rr[1]); v[3] += bf_hi(rr[1]);
            } else {
              for (int e = 0; e < 4; ++e) if (nb + e < p.N) v[e] += bf2f(rp[e]);
            }
          }
          if (p.out_f32) {
            float* cp = reinterpret_cast<float*>(p.C) + c_slab + (size_t)m * p.ldc + nb;
            if (full) {
              f32x4 o = {v[0], v[1], v[2], v[3]};
              *reinterpret_cast<f32x4*>(cp) = o;
            } else {
              for (int e = 0; e < 4; ++e) if (nb + e < p.N) cp[e] = v[e];
            }
          } else {
            bf16_t* cp = reinterpret_cast<bf16_t*>(p.C) + (size_t)m * p.ldc + nb;
            if (full) {
              u32x2 o = {pack_bf2(v[0], v[1]), pack_bf2(v[2], v[3])};
              *reinterpret_cast<u32x2*>(cp) = o;
            } else {
              for (int e = 0; e < 4; ++e) if (nb + e < p.N) cp[e] = f2bf(v[e]);
            }
          }
        }
      }
    }
  };

  // Fast path (whole wave tile inside N, bf16 output): bias preloaded once, the residual of one 32-row
  // fragment fetched as ONE batch of 8-byte loads before it is consumed.  The generic path issues a
  // dependent load + wait per register quad (32 serial round trips per lane), which cost ~20 us per
  // 256x256 tile and dominated the K = 1024 CLIP GEMMs.
  auto epilogue = [&](int m0, int n0) {
    const int ncol0 = n0 + wn * (FN * 32);
    if (p.out_f32 || ncol0 + FN * 32 > p.N) { epilogue_generic(m0, n0); return; }
    u32x2 bias_r[FN][4];
    if (p.bias) {
#pragma unroll
      for (int j = 0; j < FN; ++j)
#pragma unroll
        for (int g = 0; g < 4; ++g)
          bias_r[j][g] = *reinterpret_cast<const u32x2*>(p.bias + ncol0 + j * 32 + 8 * g + 4 * hi);
    }
#pragma unroll
    for (int i = 0; i < FM; ++i) {
      const int m = m0 + wm * (FM * 32) + i * 32 + l31;
      const int mc = min(m, p.M - 1);
      u32x2 res_r[FN][4];
      if (p.res) {
        const bf16_t* rp = p.res + (size_t)mc * p.ldr + ncol0 + 4 * hi;
#pragma unroll
        for (int j = 0; j < FN; ++j)
#pragma unroll
          for (int g = 0; g < 4; ++g) res_r[j][g] = *reinterpret_cast<const u32x2*>(rp + j * 32 + 8 * g);
      }
      bf16_t* cp = reinterpret_cast<bf16_t*>(p.C) + (size_t)mc * p.ldc + ncol0 + 4 * hi;
#pragma unroll
      for (int j = 0; j < FN; ++j)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          float v[4];
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] = acc[j][i][4 * g + e];
          if (p.bias) {
            const u32x2 bb = bias_r[j][g];
            v[0] += bf_lo(bb[0]); v[1] += bf_hi(bb[0]); v[2] += bf_lo(bb[1]); v[3] += bf_hi(bb[1]);
          }
          if (p.act == HVLM_ACT_QUICK_GELU) {
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = quick_gelu(v[e]);
          }
          if (p.res) {
            const u32x2 rr = res_r[j][g];
            v[0] += bf_lo(rr[0]); v[1] += bf_hi(rr[0]); v[2] += bf_lo(rr[1]); v[3] += bf_hi(rr[1]);
          }
          if (m < p.M) {
            u32x2 o = {pack_bf2(v[0], v[1]), pack_bf2(v[2], v[3])};
            *reinterpret_cast<u32x2*>(cp + j * 32 + 8 * g) = o;
          }
        }
    }
  };

  const int nk = (p.splits > 1 ? p.k_split : p.K) / BK;
  if constexpr (VARIANT == 6) {
    // VARIANT 6: persistent workgroups.  Each workgroup walks output tiles vb = blockIdx.x + k*gridDim.x
    // (same XCD-aware order), running ONE flat K-tile pipeline across tile boundaries: the LDS-DMA of the
    // next output tile's first K-tile is issued during the last K-tile of the current one, so the
    // epilogue (and the first-tile latency) overlaps with loads already in flight.
    auto stage_part = [&](int buf, int kt, int part) {
      const unsigned base = (unsigned)buf * STAGE_BYTES + dma_off;
      const int k0 = kt * BK;
      glds16(ga[part] + k0, base + part * 1024u, smem);
      glds16(gb[part] + k0, base + TILE_BYTES + part * 1024u, smem);
    };
    auto load_frags = [&](unsigned sb, int ks, bf16x8 (&af)[FM], bf16x8 (&bfr)[FN]) {
#pragma unroll
      for (int i = 0; i < FM; ++i)
        af[i] = *reinterpret_cast<const bf16x8*>(smem + sb + a_base + i * 4096u + koff[ks]);
#pragma unroll
      for (int j = 0; j < FN; ++j)
        bfr[j] = *reinterpret_cast<const bf16x8*>(smem + sb + b_base + j * 4096u + koff[ks]);
    };
    int vb = blockIdx.x;
    unsigned g = 0;                       // running K-tile counter: K-tile uses LDS stage g & 1
    stage(0, 0);
    bf16x8 afA[FM], bfA[FN], afB[FM], bfB[FN];
    while (true) {
#pragma unroll
      for (int j = 0; j < FN; ++j)
#pragma unroll
        for (int i = 0; i < FM; ++i)
#pragma unroll
          for (int r = 0; r < 16; ++r) acc[j][i][r] = 0.f;
      const int cm0 = m0, cn0 = n0;
      const size_t cslab = (size_t)ksl * p.slab;
      const int nvb = vb + (int)gridDim.x;
      const bool has_next = nvb < nwg;
      for (int kt = 0; kt < nk; ++kt, ++g) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        const unsigned sb = (g & 1u) * STAGE_BYTES;
        const bool last = kt + 1 == nk;
        if (last && has_next) {           // all loads of the current tile are issued: retarget the pointers
          tile_coords(nvb, tm, tn);
          m0 = tm * BM; n0 = tn * BN;
          set_pointers(m0, n0);
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
      c_slab = cslab;
      epilogue(cm0, cn0);
      if (!has_next) break;
      vb = nvb;
    }
    return;
  }
  if constexpr (VARIANT == 0) {
    stage(0, 0);
    for (int kt = 0; kt < nk; ++kt) {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // this wave's LDS-DMA pieces of tile kt landed
      __syncthreads();
      if (kt + 1 < nk) stage((kt + 1) & 1, kt + 1);
      const unsigned sb = (unsigned)(kt & 1) * STAGE_BYTES;
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) {
        bf16x8 af[FM], bfr[FN];
#pragma unroll
        for (int i = 0; i < FM; ++i)
          af[i] = *reinterpret_cast<const bf16x8*>(smem + sb + a_base + i * 4096u + koff[ks]);
#pragma unroll
        for (int j = 0; j < FN; ++j)
          bfr[j] = *reinterpret_cast<const bf16x8*>(smem + sb + b_base + j * 4096u + koff[ks]);
#pragma unroll
        for (int j = 0; j < FN; ++j)
#pragma unroll
          for (int i = 0; i < FM; ++i)
            acc[j][i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bfr[j], af[i], acc[j][i], 0, 0, 0);
      }
    }
  } else if constexpr (VARIANT != 4 && VARIANT != 5 && VARIANT != 6) {
    // VARIANT 1: register double-buffered fragments (reads of k-step s+1 fly under the MFMAs of s),
    // LDS-DMA of the next tile spread over the four k-steps (2 pieces each), s_setprio around MFMA.
    auto stage_part = [&](int buf, int kt, int part) {
      const unsigned base = (unsigned)buf * STAGE_BYTES + dma_off;
      const int k0 = kt * BK;
      if constexpr (PPW == 8) {
        glds16(ga[2 * part] + k0, base + (2 * part) * 1024u, smem);
        glds16(ga[2 * part + 1] + k0, base + (2 * part + 1) * 1024u, smem);
        glds16(gb[2 * part] + k0, base + TILE_BYTES + (2 * part) * 1024u, smem);
        glds16(gb[2 * part + 1] + k0, base + TILE_BYTES + (2 * part + 1) * 1024u, smem);
      } else if constexpr (PPW == 4) {
        glds16(ga[part] + k0, base + part * 1024u, smem);
        glds16(gb[part] + k0, base + TILE_BYTES + part * 1024u, smem);
      } else {   // PPW == 2: A pieces on even k-steps, B pieces on odd
        const int i = part >> 1;
        if (part & 1) glds16(gb[i] + k0, base + TILE_BYTES + i * 1024u, smem);
        else glds16(ga[i] + k0, base + i * 1024u, smem);
      }
    };
    auto load_frags = [&](unsigned sb, int ks, bf16x8 (&af)[FM], bf16x8 (&bfr)[FN]) {
#pragma unroll
      for (int i = 0; i < FM; ++i)
        af[i] = *reinterpret_cast<const bf16x8*>(smem + sb + a_base + i * 4096u + koff[ks]);
#pragma unroll
      for (int j = 0; j < FN; ++j)
        bfr[j] = *reinterpret_cast<const bf16x8*>(smem + sb + b_base + j * 4096u + koff[ks]);
    };
    // ablation bits (development only; results are wrong when set): VARIANT = 1 + 16*bits
    constexpr bool NO_GLDS = (VARIANT >> 4) & 1;     // no LDS-DMA inside the loop
    constexpr bool NO_READ = (VARIANT >> 4) & 2;     // fragments loaded once
    constexpr bool NO_SYNC = (VARIANT >> 4) & 4;     // no vmcnt wait / barrier inside the loop
    constexpr bool HOT_GLDS = (VARIANT >> 4) & 8;    // LDS-DMA always re-reads K-tile 0 (cache-hot source)
    stage(0, 0);
    bf16x8 afA[FM], bfA[FN], afB[FM], bfB[FN];
    if constexpr (NO_READ) {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
      load_frags(0, 0, afA, bfA);
      load_frags(0, 1, afB, bfB);
    }
    for (int kt = 0; kt < nk; ++kt) {
      if constexpr (!NO_SYNC) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
      }
      const unsigned sb = (unsigned)(kt & 1) * STAGE_BYTES;
      const bool more = kt + 1 < nk;
      if constexpr (!NO_READ) load_frags(sb, 0, afA, bfA);
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) {
        if constexpr (!NO_GLDS) {
          if (more) stage_part((kt + 1) & 1, HOT_GLDS ? 0 : kt + 1, ks);
        }
        if constexpr (!NO_READ) {
          if (ks < 3) {
            if (ks & 1) load_frags(sb, ks + 1, afA, bfA); else load_frags(sb, ks + 1, afB, bfB);
          }
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
  }

  if constexpr (VARIANT == 4 || VARIANT == 5) {
    // VARIANT 4/5: register-staged operand tiles.  Tile t+1 is fetched with plain 16-byte global loads
    // at the top of tile t (cheap to issue, fully asynchronous) and written to the other LDS stage with
    // ds_write_b128 late in tile t, instead of 8 LDS-DMA instructions per wave per tile whose issue cost
    // (~100 cycles each) was measured to take 20 % of the kernel (ablation in DESIGN.md).
    u32x4 ra[PPW], rb[PPW];
    auto gload = [&](int kt) {
      const int k0 = kt * BK;
#pragma unroll
      for (int i = 0; i < PPW; ++i) ra[i] = *reinterpret_cast<const u32x4*>(ga[i] + k0);
#pragma unroll
      for (int i = 0; i < PPW; ++i) rb[i] = *reinterpret_cast<const u32x4*>(gb[i] + k0);
    };
    const unsigned wr_off = dma_off + (unsigned)l * 16u;
    auto lwrite_a = [&](int buf) {
#pragma unroll
      for (int i = 0; i < PPW; ++i)
        *reinterpret_cast<u32x4*>(smem + (unsigned)buf * STAGE_BYTES + wr_off + i * 1024u) = ra[i];
    };
    auto lwrite_b = [&](int buf) {
#pragma unroll
      for (int i = 0; i < PPW; ++i)
        *reinterpret_cast<u32x4*>(smem + (unsigned)buf * STAGE_BYTES + TILE_BYTES + wr_off + i * 1024u) = rb[i];
    };
    auto load_frags = [&](unsigned sb, int ks, bf16x8 (&af)[FM], bf16x8 (&bfr)[FN]) {
#pragma unroll
      for (int i = 0; i < FM; ++i)
        af[i] = *reinterpret_cast<const bf16x8*>(smem + sb + a_base + i * 4096u + koff[ks]);
#pragma unroll
      for (int j = 0; j < FN; ++j)
        bfr[j] = *reinterpret_cast<const bf16x8*>(smem + sb + b_base + j * 4096u + koff[ks]);
    };
    gload(0);
    lwrite_a(0);
    lwrite_b(0);
    bf16x8 afA[FM], bfA[FN], afB[FM], bfB[FN];
    for (int kt = 0; kt < nk; ++kt) {
      __syncthreads();
      const unsigned sb = (unsigned)(kt & 1) * STAGE_BYTES;
      const bool more = kt + 1 < nk;
      if (more) gload(kt + 1);
      load_frags(sb, 0, afA, bfA);
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) {
        if (ks < 3) {
          if (ks & 1) load_frags(sb, ks + 1, afA, bfA); else load_frags(sb, ks + 1, afB, bfB);
        }
        if (more) {
          if constexpr (VARIANT == 4) {
            if (ks == 2) lwrite_a((kt + 1) & 1);
            if (ks == 3) lwrite_b((kt + 1) & 1);
          } else {
            if (ks == 3) { lwrite_a((kt + 1) & 1); lwrite_b((kt + 1) & 1); }
          }
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
  }

  epilogue(m0, n0);
}

}  // namespace

}  // namespace hvlm

using namespace hvlm;

// development ablations of the ping-pong kernel's K loop (wrong results by construction; timing / PMC only):
// 1 = no LDS-DMA in load segment B, 2 = none in load segment A, 4 = staging cursors never advance (cache-hot source),
// 8 = no fragment reads in load segment B (8 of the 24 ds_read_b128 per K-tile and wave)
static int gemm_dbg() {
  static const int v = getenv("HVLM_GEMM_DBG") ? atoi(getenv("HVLM_GEMM_DBG")) : 0;
  return v;
}

extern "C" int hvlm_gemm_bf16_nt(const void* A, const void* B, void* C, int M, int N, int K, int lda,
                                 int ldb, int ldc, const void* bias, const void* residual, int ldr,
                                 int act, int out_f32, void* stream) {
  HVLM_REQUIRE(A && B && C, "gemm: null operand");
  HVLM_REQUIRE(M > 0 && N > 0 && K > 0, "gemm: bad shape M=%d N=%d K=%d", M, N, K);
  HVLM_REQUIRE(K % BK == 0, "gemm: K=%d must be a multiple of %d", K, BK);
  HVLM_REQUIRE(lda >= K && ldb >= K && ldc >= N, "gemm: leading dimension too small");
  HVLM_REQUIRE(lda % 8 == 0 && ldb % 8 == 0 && ldc % 4 == 0, "gemm: lda/ldb %% 8, ldc %% 4 required");
  HVLM_REQUIRE(((uintptr_t)A & 15) == 0 && ((uintptr_t)B & 15) == 0 && ((uintptr_t)C & 15) == 0,
               "gemm: operands must be 16-byte aligned");
  HVLM_REQUIRE(!residual || (ldr >= N && ldr % 4 == 0 && ((uintptr_t)residual & 7) == 0),
               "gemm: bad residual");
  HVLM_REQUIRE(!bias || ((uintptr_t)bias & 7) == 0, "gemm: bias must be 8-byte aligned");
  HVLM_REQUIRE(act == HVLM_ACT_NONE || act == HVLM_ACT_QUICK_GELU, "gemm: unknown act %d", act);
  GemmArgs p;
  p.A = (const bf16_t*)A; p.B = (const bf16_t*)B; p.C = C;
  p.bias = (const bf16_t*)bias; p.res = (const bf16_t*)residual;
  p.M = M; p.N = N; p.K = K; p.lda = lda; p.ldb = ldb; p.ldc = ldc; p.ldr = ldr;
  p.act = act; p.out_f32 = out_f32;
  p.tiles_m = cdiv(M, BM); p.tiles_n = cdiv(N, BN);
  p.splits = 1; p.k_split = K; p.slab = 0; p.alpha = 1.f; p.dyn = 0; p.dbg = gemm_dbg();
  { const char* ge = getenv("HVLM_GEMM_GROUP_M"); p.group_m = ge ? atoi(ge) : GROUP_M; if (p.group_m < 1) p.group_m = 1; }
  const char* ve = getenv("HVLM_GEMM_VARIANT");   // development switch; default = tuned variant
  const int variant = ve ? atoi(ve) : 9;
  int grid = p.tiles_m * p.tiles_n;
  static int pgrid = 0;
  if (!pgrid) {
    int dev = 0; hipDeviceProp_t prop;
    if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess) pgrid = prop.multiProcessorCount;
    else pgrid = 256;
    pgrid = (pgrid / 8) * 8;
  }
#define HVLM_LAUNCH_GEMM(V, WM_, WN_)                                                                 \
  do {                                                                                                \
    static bool attr_done = false;                                                                    \
    if (!attr_done) {                                                                                 \
      hipError_t e = hipFuncSetAttribute((const void*)gemm_bf16_nt_kernel<V, WM_, WN_>,               \
                                         hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);      \
      if (e != hipSuccess) {                                                                          \
        set_error("gemm: hipFuncSetAttribute failed: %s", hipGetErrorString(e));                      \
        return HVLM_ERR_LAUNCH;                                                                       \
      }                                                                                               \
      attr_done = true;                                                                               \
    }                                                                                                 \
    hipLaunchKernelGGL((gemm_bf16_nt_kernel<V, WM_, WN_>), dim3(grid), dim3(WM_ * WN_ * 64), LDS_BYTES, \
                       as_stream(stream), p);                                                         \
  } while (0)
  switch (variant) {
    case 0: HVLM_LAUNCH_GEMM(0, 2, 4); break;
    case 2: HVLM_LAUNCH_GEMM(0, 4, 4); break;
    case 3: HVLM_LAUNCH_GEMM(1, 4, 4); break;
    case 7: HVLM_LAUNCH_GEMM(1, 2, 2); break;
    case 8: HVLM_LAUNCH_GEMM(0, 2, 2); break;
    case 4: HVLM_LAUNCH_GEMM(4, 2, 4); break;
    case 6: { const int full = grid; grid = grid < pgrid ? grid : pgrid; (void)full; HVLM_LAUNCH_GEMM(6, 2, 4); break; }
    case 5: HVLM_LAUNCH_GEMM(5, 2, 4); break;
    case 9: {
      // dynamic tile scheduling needs >= 3 units per workgroup: every workgroup commits to three units when it starts,
      // so with fewer the first arrivals would take everything (256 tiles on 256 CUs: 1383 -> 917 TFLOP/s)
      p.dyn = (K / BK >= 3 && pgrid <= 512 && grid >= 3 * pgrid && !getenv("HVLM_GEMM_STATIC")) ? 1 + pp_sched_slot(as_stream(stream)) : 0;
      const int rc = launch_gemm_pp(p, grid < pgrid ? grid : pgrid, as_stream(stream));
      if (rc) return rc;
      break;
    }
    case 17: HVLM_LAUNCH_GEMM(17, 2, 4); break;
    case 129: HVLM_LAUNCH_GEMM(129, 2, 4); break;
    case 33: HVLM_LAUNCH_GEMM(33, 2, 4); break;
    case 49: HVLM_LAUNCH_GEMM(49, 2, 4); break;
    case 113: HVLM_LAUNCH_GEMM(113, 2, 4); break;
    case 65: HVLM_LAUNCH_GEMM(65, 2, 4); break;
    default: HVLM_LAUNCH_GEMM(1, 2, 4); break;
  }
#undef HVLM_LAUNCH_GEMM
  return check_launch("gemm_bf16_nt");
}

// ---- split-K for skinny problems (few output tiles, e.g. the batch-1 prefill M = 867) --------------------------
namespace hvlm {
namespace {
__global__ __launch_bounds__(256) void splitk_finalize_kernel(const float* __restrict__ ws, int splits, size_t slab,
                                                              const bf16_t* __restrict__ bias,
                                                              const bf16_t* __restrict__ res, int ldr, int act,
                                                              bf16_t* __restrict__ C, int ldc, int M, int N) {
  const long total = (long)M * (N / 4);
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
    const long m = i / (N / 4);
    const int n = (int)(i - m * (N / 4)) * 4;
    f32x4 v = *reinterpret_cast<const f32x4*>(ws + m * (long)N + n);
    for (int s = 1; s < splits; ++s) {
      const f32x4 t = *reinterpret_cast<const f32x4*>(ws + s * slab + m * (long)N + n);
      v += t;
    }
    float o[4] = {v[0], v[1], v[2], v[3]};
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      if (bias) o[e] += bf2f(bias[n + e]);
      if (act == HVLM_ACT_QUICK_GELU) o[e] = quick_gelu(o[e]);
      if (res) o[e] += bf2f(res[m * (long)ldr + n + e]);
    }
    u32x2 pk = {pack_bf2(o[0], o[1]), pack_bf2(o[2], o[3])};
    *reinterpret_cast<u32x2*>(C + m * (long)ldc + n) = pk;
  }
}
}  // namespace
}  // namespace hvlm

extern "C" size_t hvlm_gemm_splitk_ws_bytes(int M, int N, int splits) { return (size_t)splits * M * N * 4; }

extern "C" int hvlm_gemm_bf16_nt_splitk(const void* A, const void* B, void* C, int M, int N, int K, int lda, int ldb,
                                        int ldc, const void* bias, const void* residual, int ldr, int act, void* ws,
                                        int splits, void* stream) {
  HVLM_REQUIRE(A && B && C && ws, "gemm_splitk: null pointer");
  HVLM_REQUIRE(M > 0 && N > 0 && K > 0 && splits >= 2, "gemm_splitk: bad shape / splits");
  HVLM_REQUIRE(K % (BK * splits) == 0, "gemm_splitk: K=%d must be a multiple of %d*splits", K, BK);
  HVLM_REQUIRE(N % 4 == 0 && lda >= K && ldb >= K && ldc >= N && lda % 8 == 0 && ldb % 8 == 0 && ldc % 4 == 0,
               "gemm_splitk: bad leading dimensions (N %% 4)");
  HVLM_REQUIRE(((uintptr_t)A & 15) == 0 && ((uintptr_t)B & 15) == 0 && ((uintptr_t)C & 7) == 0 && ((uintptr_t)ws & 15) == 0,
               "gemm_splitk: alignment");
  GemmArgs p;
  p.A = (const bf16_t*)A; p.B = (const bf16_t*)B; p.C = ws; p.bias = nullptr; p.res = nullptr;
  p.M = M; p.N = N; p.K = K; p.lda = lda; p.ldb = ldb; p.ldc = N; p.ldr = 0; p.act = HVLM_ACT_NONE; p.out_f32 = 1;
  p.tiles_m = cdiv(M, BM); p.tiles_n = cdiv(N, BN); p.group_m = GROUP_M;
  p.splits = splits; p.k_split = K / splits; p.slab = (size_t)M * N; p.alpha = 1.f; p.dyn = 0; p.dbg = gemm_dbg();
  static int cus = 0;
  if (!cus) {
    int dev = 0; hipDeviceProp_t prop;
    cus = (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess) ? prop.multiProcessorCount : 256;
    cus = (cus / 8) * 8;
  }
  int grid = p.tiles_m * p.tiles_n * splits;
  if (grid > cus) grid = cus;
  p.dyn = (p.k_split / BK >= 3 && cus <= 512 && p.tiles_m * p.tiles_n * splits >= 3 * cus && !getenv("HVLM_GEMM_STATIC")) ? 1 + pp_sched_slot(as_stream(stream)) : 0;
  int rc = launch_gemm_pp(p, grid, as_stream(stream));
  if (rc) return rc;
  rc = check_launch("gemm_splitk");
  if (rc) return rc;
  const long total = (long)M * (N / 4);
  int fg = (int)((total + 255) / 256); if (fg > 2048) fg = 2048;
  hipLaunchKernelGGL(splitk_finalize_kernel, dim3(fg), dim3(256), 0, as_stream(stream), (const float*)ws, splits,
                     (size_t)M * N, (const bf16_t*)bias, (const bf16_t*)residual, ldr, act, (bf16_t*)C, ldc, M, N);
  return check_launch("gemm_splitk_finalize");
}

// ---- fp8 (OCP e4m3) operands: CDNA4 v_mfma_scale_f32_32x32x64_f8f6f4 ------------------------------------------------
namespace hvlm {
namespace {
// out[i] = e4m3(clamp(x[i] * scale, -448, 448)); 8 elements per thread
__global__ __launch_bounds__(256) void quantize_fp8_kernel(const bf16_t* __restrict__ x, unsigned char* __restrict__ out,
                                                           const float* __restrict__ scale_ptr, float scale_imm, long n8) {
  const float sc = scale_ptr ? *scale_ptr : scale_imm;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n8; i += (long)gridDim.x * 256) {
    const u32x4 v = reinterpret_cast<const u32x4*>(x)[i];
    float f[8];
#pragma unroll
    for (int e = 0; e < 4; ++e) { f[2 * e] = bf_lo(v[e]) * sc; f[2 * e + 1] = bf_hi(v[e]) * sc; }
#pragma unroll
    for (int e = 0; e < 8; ++e) f[e] = fminf(fmaxf(f[e], -448.f), 448.f);
    unsigned lo = 0, hi = 0;
    lo = __builtin_amdgcn_cvt_pk_fp8_f32(f[0], f[1], lo, false);
    lo = __builtin_amdgcn_cvt_pk_fp8_f32(f[2], f[3], lo, true);
    hi = __builtin_amdgcn_cvt_pk_fp8_f32(f[4], f[5], hi, false);
    hi = __builtin_amdgcn_cvt_pk_fp8_f32(f[6], f[7], hi, true);
    const u32x2 o = {lo, hi};
    reinterpret_cast<u32x2*>(out)[i] = o;
  }
}
}  // namespace
}  // namespace hvlm

extern "C" int hvlm_quantize_fp8(const void* x, void* out, const float* scale_dev, float scale, long n, void* stream) {
  HVLM_REQUIRE(x && out && n > 0 && n % 8 == 0, "quantize_fp8: n must be a positive multiple of 8");
  HVLM_REQUIRE(((uintptr_t)x & 15) == 0 && ((uintptr_t)out & 7) == 0, "quantize_fp8: alignment");
  const long n8 = n / 8;
  long grid = (n8 + 255) / 256; if (grid > 4096) grid = 4096;
  hipLaunchKernelGGL(quantize_fp8_kernel, dim3((unsigned)grid), dim3(256), 0, as_stream(stream), (const bf16_t*)x,
                     (unsigned char*)out, scale_dev, scale, n8);
  return check_launch("quantize_fp8");
}

namespace hvlm {
namespace {
// Per-row (per-token / per-output-channel) quantisation: s = 448 / amax(row); out = e4m3(x * s); inv_scale[row] = amax / 448.
// One workgroup per row; the row is read twice (amax, then convert) — the second read hits L2.
__global__ __launch_bounds__(256) void quantize_fp8_rowwise_kernel(const bf16_t* __restrict__ x, unsigned char* __restrict__ out,
                                                                   float* __restrict__ inv_scale, int K, long ldx, long ldo) {
  __shared__ float red[4];
  const long row = blockIdx.x;
  const bf16_t* xr = x + row * ldx;
  float mx = 0.f;
  for (int i = threadIdx.x * 8; i < K; i += 256 * 8) {
    const u32x4 v = *reinterpret_cast<const u32x4*>(xr + i);
#pragma unroll
    for (int e = 0; e < 4; ++e) mx = fmaxf(mx, fmaxf(fabsf(bf_lo(v[e])), fabsf(bf_hi(v[e]))));
  }
  mx = block_max<4>(mx, red);
  const float amax = fmaxf(mx, 1e-12f);
  const float sc = 448.f / amax;
  if (threadIdx.x == 0) inv_scale[row] = amax / 448.f;
  unsigned char* orow = out + row * ldo;
  for (int i = threadIdx.x * 8; i < K; i += 256 * 8) {
    const u32x4 v = *reinterpret_cast<const u32x4*>(xr + i);
    float f[8];
#pragma unroll
    for (int e = 0; e < 4; ++e) { f[2 * e] = bf_lo(v[e]) * sc; f[2 * e + 1] = bf_hi(v[e]) * sc; }
#pragma unroll
    for (int e = 0; e < 8; ++e) f[e] = fminf(fmaxf(f[e], -448.f), 448.f);
    unsigned lo = 0, hi = 0;
    lo = __builtin_amdgcn_cvt_pk_fp8_f32(f[0], f[1], lo, false);
    lo = __builtin_amdgcn_cvt_pk_fp8_f32(f[2], f[3], lo, true);
    hi = __builtin_amdgcn_cvt_pk_fp8_f32(f[4], f[5], hi, false);
    hi = __builtin_amdgcn_cvt_pk_fp8_f32(f[6], f[7], hi, true);
    const u32x2 o = {lo, hi};
    *reinterpret_cast<u32x2*>(orow + i) = o;
  }
}
}  // namespace
}  // namespace hvlm

extern "C" int hvlm_quantize_fp8_rowwise(const void* x, void* out, float* inv_scale, int rows, int K, int ldx, int ldo,
                                         void* stream) {
  HVLM_REQUIRE(x && out && inv_scale && rows > 0 && K > 0 && K % 8 == 0 && ldx % 8 == 0 && ldo % 8 == 0 && ldx >= K && ldo >= K,
               "quantize_fp8_rowwise: bad args (K, ldx, ldo multiples of 8)");
  HVLM_REQUIRE(((uintptr_t)x & 15) == 0 && ((uintptr_t)out & 7) == 0, "quantize_fp8_rowwise: alignment");
  hipLaunchKernelGGL(quantize_fp8_rowwise_kernel, dim3(rows), dim3(256), 0, as_stream(stream), (const bf16_t*)x,
                     (unsigned char*)out, inv_scale, K, (long)ldx, (long)ldo);
  return check_launch("quantize_fp8_rowwise");
}

static int gemm_fp8_impl(const void* A, const void* B, void* C, int M, int N, int K, int lda, int ldb, int ldc,
                         float alpha, const float* row_scale, const float* col_scale, const void* bias,
                         const void* residual, int ldr, int act, int out_f32, void* stream);

extern "C" int hvlm_gemm_fp8_nt(const void* A, const void* B, void* C, int M, int N, int K, int lda, int ldb, int ldc,
                                float alpha, const void* bias, const void* residual, int ldr, int act, int out_f32,
                                void* stream) {
  return gemm_fp8_impl(A, B, C, M, N, K, lda, ldb, ldc, alpha, nullptr, nullptr, bias, residual, ldr, act, out_f32, stream);
}

extern "C" int hvlm_gemm_fp8_nt_scaled(const void* A, const void* B, void* C, int M, int N, int K, int lda, int ldb, int ldc,
                                       float alpha, const float* row_scale, const float* col_scale, const void* bias,
                                       const void* residual, int ldr, int act, int out_f32, void* stream) {
  return gemm_fp8_impl(A, B, C, M, N, K, lda, ldb, ldc, alpha, row_scale, col_scale, bias, residual, ldr, act, out_f32, stream);
}

static int gemm_fp8_impl(const void* A, const void* B, void* C, int M, int N, int K, int lda, int ldb, int ldc,
                         float alpha, const float* row_scale, const float* col_scale, const void* bias,
                         const void* residual, int ldr, int act, int out_f32, void* stream) {
  HVLM_REQUIRE(A && B && C, "gemm_fp8: null operand");
  HVLM_REQUIRE(M > 0 && N > 0 && K > 0 && K % 128 == 0, "gemm_fp8: K=%d must be a positive multiple of 128", K);
  HVLM_REQUIRE(lda >= K && ldb >= K && ldc >= N && lda % 16 == 0 && ldb % 16 == 0 && ldc % 4 == 0, "gemm_fp8: bad leading dimensions");
  HVLM_REQUIRE(((uintptr_t)A & 15) == 0 && ((uintptr_t)B & 15) == 0 && ((uintptr_t)C & 15) == 0, "gemm_fp8: 16-byte alignment");
  HVLM_REQUIRE(!residual || (ldr >= N && ldr % 4 == 0 && ((uintptr_t)residual & 7) == 0), "gemm_fp8: bad residual");
  HVLM_REQUIRE(!bias || ((uintptr_t)bias & 7) == 0, "gemm_fp8: bias must be 8-byte aligned");
  HVLM_REQUIRE(act == HVLM_ACT_NONE || act == HVLM_ACT_QUICK_GELU, "gemm_fp8: unknown act %d", act);
  GemmArgs p;
  p.A = (const bf16_t*)A; p.B = (const bf16_t*)B; p.C = C; p.bias = (const bf16_t*)bias; p.res = (const bf16_t*)residual;
  p.M = M; p.N = N; p.K = K; p.lda = lda; p.ldb = ldb; p.ldc = ldc; p.ldr = ldr; p.act = act; p.out_f32 = out_f32;
  p.tiles_m = cdiv(M, BM); p.tiles_n = cdiv(N, BN); p.group_m = GROUP_M;
  p.splits = 1; p.k_split = K; p.slab = 0; p.alpha = alpha; p.dyn = 0; p.dbg = 0;
  p.row_scale = row_scale; p.col_scale = col_scale;
  static int cus = 0;
  if (!cus) {
    int dev = 0; hipDeviceProp_t prop;
    cus = (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess) ? prop.multiProcessorCount : 256;
    cus = (cus / 8) * 8;
  }
  int grid = p.tiles_m * p.tiles_n;
  if (grid > cus) grid = cus;
  p.dyn = (K / 128 >= 3 && cus <= 512 && p.tiles_m * p.tiles_n >= 3 * cus && !getenv("HVLM_GEMM_STATIC")) ? 1 + pp_sched_slot(as_stream(stream)) : 0;
  const int rc = launch_gemm_pp_fp8(p, grid, as_stream(stream));
  if (rc) return rc;
  return check_launch("gemm_fp8_nt");
}
