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
//              S = Q K^T, dP = dO V^T, dV^T = dO^T P, dK^T = Q^T dS — in ONE pass by role-split waves (attn_bwd_dkv_fused:
//              four waves own S -> P -> dV, four own dP -> dS -> dK, P is handed over through LDS).  No atomics, deterministic.
//              The gradient of the rotary embedding can ride in the dQ / dK epilogues (hvlm_attn_bwd_rope).
//   Workgroups are mapped XCD-aware (xcd_block): the tiles of one (sample, head) share one L2.  Outputs leave through a
//   wave-private LDS block as 16-byte row-contiguous stores (store_block_rows).
//   The "transposed" operand fragments (V^T, K^T, Q^T, dO^T) are gathered from the row-major K / V / Q / dO tiles in
//   LDS with transpose reads (ds_read_b64_tr_b16): no transposed tiles, no pre-passes, no workspace beyond delta.
//   K/V/Q/dO tiles go HBM -> LDS by LDS-DMA (buffer_load_dwordx4 ... lds) with the XOR swizzle
//   applied to the per-lane source address (conflict-free ds_read_b128 fragment reads).
//   Sequences are padded [B*S] or packed (cu_seqlens: the reference's flash_attn_varlen contract); the forward also has
//   an fp16 instantiation for the CLIP tower and a resident-K/V persistent variant for S <= 288 (CLIP, S = 257).
//
// Reference arithmetic replaced: LlamaAttention softmax(QK^T/sqrt(dh) + causal mask) V
// (transformers modeling_llama; flash path llava/train/llama_flash_attn_monkey_patch.py:81-94) and
// CLIPAttention (non-causal, dh = 64).
#include "common.h"
#include <type_traits>
#include <stdlib.h>

namespace hvlm {
// attention_asm.hip
int attn_fwd_d128_causal_asm(const bf16_t* qkv, bf16_t* out, float* lse, void* ws, int B, int S_max, int H, int ld, float scale_log2,
                             const int32_t* cu, hipStream_t st, int variant = 0, void* prof = nullptr);
int attn_fwd_clip_s257_asm(const bf16_t* qkv, bf16_t* out, int B, int H, int ld, float scale_log2, hipStream_t st, int variant = 0,
                           void* prof = nullptr);
bool attn_fwd_d128_causal_asm_ok(const void* ws, int B, int S_max, int H);   // preconditions of the two assembly kernels: a call
bool attn_fwd_clip_s257_asm_ok(int B, int H, int ld);                        // that does not meet them runs on the C++ kernel below
namespace {

typedef __attribute__((address_space(3))) void* lds_ptr_t;
typedef const __attribute__((address_space(1))) void* gbl_ptr_t;
typedef __attribute__((ext_vector_type(4))) short s16x4_t;
typedef __attribute__((ext_vector_type(8))) short s16x8_t;

// LDS-DMA of 16 bytes per lane through a BUFFER load (base = wave-uniform pointer, per-lane 32-bit byte offset).
// Not `global_load_lds`: that one is a FLAT-class instruction with an LDS memory operand, and while one is pending
// the compiler's wait-count pass turns EVERY later lgkmcnt wait (i.e. every ds_read result) into `vmcnt(0) lgkmcnt(0)`
// — the prefetch of the next tile, issued before the compute of the current one, was silently serialised.
__device__ __forceinline__ void glds16(const bf16_t* base, unsigned byte_off, unsigned lds_off, char* smem) {
  const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(base), 0, 0xffffffffu, 0x00020000);
  __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_ptr_t)(smem + lds_off), 16, byte_off, 0, 0, 0);
}

// LDS tile image: row-major [rows][COLS] bf16, the 16-byte chunk c of row r stored at chunk position c ^ swz(r).  The
// permutation has to serve TWO read patterns of the same tile without bank conflicts:
//   * ds_read_b128 fragment reads: a 16-lane service group = 16 rows (distinct mod 16), one chunk each -> swz must be a
//     bijection of the row bits it looks at (any bit permutation is);
//   * ds_read_b64_tr_b16 transpose reads: a 32-lane group = rows r0..r0+3 (r0 % 4 == 0) x one aligned group of four chunks
//     (64 B) x two 8-B halves -> the FOUR ROWS must land in four different 64-B bank quarters.
// Round 1 used swz = r & 15 (256-B rows) / (r >> 1) & 7 (128-B rows): conflict-free for the first pattern, but the four
// rows of a transpose read only permuted chunks INSIDE one quarter — 4-way (2-way) conflicts, 46-60 % of the LDS cycles of
// the attention kernels (SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE).  Now the low row bits select the quarter:
//   256-B rows (dh = 128): swz = ((r & 3) << 2) | ((r >> 2) & 3)          (rows r0..r0+3 -> quarters 0..3)
//   128-B rows (dh = 64):  swz = (((r >> 1) & 1) << 2) | ((r >> 2) & 3)   (with the bank half r & 1: four distinct 64-B groups)
template <int COLS>
__device__ __forceinline__ int swz_bits(int r) {
  if constexpr (COLS == 128) return ((r & 3) << 2) | ((r >> 2) & 3);
  else return (((r >> 1) & 1) << 2) | ((r >> 2) & 3);
}
// byte offset of 16-byte chunk `c` (logical) of row `r` inside a swizzled [rows][COLS] bf16 tile
template <int COLS>
__device__ __forceinline__ unsigned tile_addr(int r, int c) {
  if constexpr (COLS == 128) return (unsigned)r * 256u + (unsigned)((c ^ swz_bits<128>(r)) << 4);
  else return (unsigned)r * 128u + (unsigned)((c ^ swz_bits<64>(r)) << 4);
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
    glds16(gbase, (unsigned)(((long)gr * stride + c * 8) * 2), tile_off + (unsigned)pi * 1024u, smem);
  }
}

// LDS-DMA of 64-row x COLS tiles with everything tile-independent hoisted out of the loop: the per-lane byte offset of each
// of the wave's pieces (row inside the tile, swizzled source chunk) is computed ONCE (`init`), the tile enters as one scalar
// byte offset, and the sample's row range is enforced by the buffer descriptor (rows >= S read as zeros) instead of a
// per-lane clamp.  The first version recomputed row, swizzle, clamp and a 64-bit address for every piece of every tile:
// measured 580-920 cycles per 64-key step in the forward kernel, 11-17 % of the step (in-kernel cycle stamps).
template <int COLS, int NW = 4>
struct TileDma {
  static constexpr int CPR = COLS / 8, RPI = 64 / CPR, NPIECE = 64 / RPI, PPW = (NPIECE + NW - 1) / NW;
  unsigned voff[PPW];
  int w_;
  __device__ __forceinline__ void init(long stride, int w, int l) {
    w_ = w;
#pragma unroll
    for (int i = 0; i < PPW; ++i) {
      const int pi = i * NW + w;
      const int r = pi * RPI + l / CPR;
      const int c = (l % CPR) ^ swz_bits<COLS>(r);
      voff[i] = (unsigned)(((long)r * stride + c * 8) * 2);
    }
  }
  // rs: descriptor over the sample's rows (base = row 0 of the sample, num_records = S * stride * 2 bytes); soff = byte
  // offset of the tile's first row + the operand's column offset; tile_off = LDS byte offset of the tile
  __device__ __forceinline__ void issue(__amdgpu_buffer_rsrc_t rs, unsigned soff, unsigned tile_off, char* smem) const {
#pragma unroll
    for (int i = 0; i < PPW; ++i) {
      const int pi = i * NW + w_;
      if (NPIECE % NW != 0 && pi >= NPIECE) break;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_ptr_t)(smem + tile_off + (unsigned)pi * 1024u), 16, voff[i], soff, 0, 0);
    }
  }
};
__device__ __forceinline__ __amdgpu_buffer_rsrc_t sample_rsrc(const bf16_t* row0, long rows, long stride) {
  const unsigned long bytes = (unsigned long)rows * (unsigned long)stride * 2ul;
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(row0), 0, bytes > 0xffffffffull ? 0xffffffffu : (unsigned)bytes, 0x00020000);
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
template <bool F16 = false>
__device__ __forceinline__ bf16x8 pack8(const float* p) {
  u32x4 v = {E16<F16>::pack2(p[0], p[1]), E16<F16>::pack2(p[2], p[3]), E16<F16>::pack2(p[4], p[5]), E16<F16>::pack2(p[6], p[7])};
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

// XCD-aware workgroup -> (tile, head, sample) mapping.  The hardware deals consecutive workgroups round-robin over the 8 XCDs
// (each with its own 4 MiB L2), so with the plain (tile = blockIdx.x, head = .y, sample = .z) order the q-tiles of ONE
// (sample, head) — which all stream the same K / V rows — land on 8 different XCDs and every XCD fetches that K / V from
// the fabric itself.  Here XCD x takes a CONTIGUOUS chunk of the (sample, head, tile) order: the tiles of one (sample, head)
// run on one XCD at about the same time and share its L2.
__device__ __forceinline__ void xcd_block(int& bx, int& by, int& bz) {
  const int gx = (int)gridDim.x, gy = (int)gridDim.y;
  const int total = gx * gy * (int)gridDim.z;
  const int lin = (int)blockIdx.x + gx * ((int)blockIdx.y + gy * (int)blockIdx.z);
  const int x = lin & 7, q = total >> 3, r = total & 7;
  const int v = x * q + min(x, r) + (lin >> 3);
  bx = v % gx;
  const int rest = v / gx;
  by = rest % gy;
  bz = rest / gy;
}

// Epilogue helper: a wave's 32 x DH block in the "one row per lane" accumulator layout (registers = head-dim quads, see the
// design comment at the top) goes through a wave-private LDS block and leaves with 16-byte row-contiguous stores: 8 (4 at
// dh = 64) store instructions of 4 (8) full rows each instead of 16 (8) instructions that touch 32 rows with 16 bytes apiece.
// `blk` = 32 * DH * 2 bytes of LDS nobody else uses; chunk c of row r sits at c ^ (r & (DH/8 - 1)).
template <int DH, bool F16>
__device__ __forceinline__ void store_block_rows(char* blk, const f32x16 (&o)[DH / 32], float mul, bf16_t* grow0, long gstride,
                                                 int rows_valid, int l) {
  constexpr int ROWB = DH * 2, CPR = DH / 8;
  const int l31 = l & 31, hi = l >> 5;
#pragma unroll
  for (int db = 0; db < DH / 32; ++db)
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const int c = db * 4 + g;
      const u32x2 v = {E16<F16>::pack2(o[db][4 * g] * mul, o[db][4 * g + 1] * mul),
                       E16<F16>::pack2(o[db][4 * g + 2] * mul, o[db][4 * g + 3] * mul)};
      *reinterpret_cast<u32x2*>(blk + l31 * ROWB + ((c ^ (l31 & (CPR - 1))) << 4) + hi * 8) = v;
    }
  constexpr int RPI = 64 / CPR;            // rows per store instruction
  const int c = l % CPR;
#pragma unroll
  for (int it = 0; it < 32 / RPI; ++it) {
    const int r = it * RPI + l / CPR;
    const u32x4 v = *reinterpret_cast<const u32x4*>(blk + r * ROWB + ((c ^ (r & (CPR - 1))) << 4));
    if (r < rows_valid) *reinterpret_cast<u32x4*>(grow0 + (long)r * gstride + c * 8) = v;
  }
}

// Inverse RoPE (the gradient of apply_rotary_pos_emb, half-split pairs (d, d + DH/2)) applied to a 32 x DH gradient block in
// the "one row per lane" accumulator layout BEFORE it is stored — the separate rope pass over dqkv (hvlm_rope_qk, inverse)
// re-read and re-wrote two thirds of the buffer.  Bit-identical to that pass: the values are first scaled and rounded to
// bf16 exactly as they used to be stored, then rotated with rope_kernel's op-by-op bf16 arithmetic
// (o1 = bf16(bf16(x1*c) + bf16(x2*s)), o2 = bf16(bf16(x2*c) + bf16(-x1*s)) for the inverse).
// cos_tab / sin_tab: bf16 [max_pos, DH/2]; p = position of this lane's row.
template <int DH>
struct RopeRow {                       // cos / sin of this lane's row for its DH/4 rotation pairs (16 registers each at dh = 128)
  u32x2 cv[DH / 64][4], sv[DH / 64][4];
  __device__ __forceinline__ void load(const bf16_t* __restrict__ cos_tab, const bf16_t* __restrict__ sin_tab, int p, int hi) {
#pragma unroll
    for (int db = 0; db < DH / 64; ++db)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int d = db * 32 + 8 * g + 4 * hi;          // 4 consecutive rotation pairs d .. d+3 (< DH/2)
        cv[db][g] = *reinterpret_cast<const u32x2*>(cos_tab + (long)p * (DH / 2) + d);
        sv[db][g] = *reinterpret_cast<const u32x2*>(sin_tab + (long)p * (DH / 2) + d);
      }
  }
};
template <int DH>
__device__ __forceinline__ void rope_inverse_block(f32x16 (&o)[DH / 32], float mul, const RopeRow<DH>& rr) {
  static_assert(DH / 32 % 2 == 0, "rope pairs (d, d + DH/2) sit in accumulator blocks db and db + DH/64");
#pragma unroll
  for (int db = 0; db < DH / 64; ++db)
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const u32x2 cv = rr.cv[db][g], sv = rr.sv[db][g];
      const float cc[4] = {bf_lo(cv[0]), bf_hi(cv[0]), bf_lo(cv[1]), bf_hi(cv[1])};
      const float ss[4] = {bf_lo(sv[0]), bf_hi(sv[0]), bf_lo(sv[1]), bf_hi(sv[1])};
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float a = bf2f(f2bf(o[db][4 * g + e] * mul));
        const float b = bf2f(f2bf(o[db + DH / 64][4 * g + e] * mul));
        const float sneg = -ss[e];
        o[db][4 * g + e] = bf2f(f2bf(a * cc[e])) + bf2f(f2bf(-b * sneg));
        o[db + DH / 64][4 * g + e] = bf2f(f2bf(b * cc[e])) + bf2f(f2bf(a * sneg));
      }
    }
}

constexpr float LOG2E = 1.4426950408889634f;
constexpr float LN2 = 0.6931471805599453f;

// =================================================================================================
// forward
// =================================================================================================
// (the body lives in a __device__ template: a __global__ TEMPLATE whose body names device-only types — the buffer
// resource — or holds inline asm with AMDGPU register constraints is dropped silently by the host pass, leaving its launch
// stub undefined)
template <int DH, bool CAUSAL, int NW, bool F16>
__device__ __forceinline__ void attn_fwd_body(const bf16_t* __restrict__ qkv, const bf16_t* __restrict__ vt,
                                              bf16_t* __restrict__ out, float* __restrict__ lse, int S_max, int H, int ld,
                                              int S_pad, float scale_log2, const int32_t* __restrict__ cu_seqlens) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr unsigned KT = 64 * DH * 2;   // K tile bytes ; V^T tile has the same size
  constexpr unsigned STAGE = 2 * KT;
  const int tid = threadIdx.x, l = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l31 = l & 31, hi = l >> 5;
  constexpr int QB = NW * 32;            // query rows per workgroup
  int bx, h, b;
  xcd_block(bx, h, b);
  // varlen (cu_seqlens != NULL, llama_flash_attn_monkey_patch.py:81-94): sample b owns the packed rows
  // [cu[b], cu[b+1]); the grid covers max_seqlen and the workgroups past this sample's length leave at once.
  const long tok0 = cu_seqlens ? (long)cu_seqlens[b] : (long)b * S_max;
  const int S = cu_seqlens ? (cu_seqlens[b + 1] - cu_seqlens[b]) : S_max;
  const int nqt = (S + QB - 1) / QB;
  if (bx >= nqt) return;
  const int qt = CAUSAL ? (nqt - 1 - bx) : bx;  // heavy tiles first
  const int q0 = qt * QB;
  const int qi = q0 + w * 32 + l31;

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

  // a stage = the K and V tiles of 64 keys, both row-major; the V^T fragments of O^T = V^T P^T are gathered from the V
  // tile with `ds_read_b64_tr_b16` (inline asm, see the backward kernels) — no V^T pre-pass, no workspace
  TileDma<DH, NW> dma;
  dma.init(ld, w, l);
  const __amdgpu_buffer_rsrc_t rs_kv = sample_rsrc(qkv + tok0 * ld, S, ld);
  const unsigned k_col = (unsigned)(((long)H * DH + h * DH) * 2), v_col = k_col + (unsigned)((long)H * DH * 2);
  auto stage = [&](int buf, int kt) {
    const unsigned base = (unsigned)buf * STAGE;
    const unsigned row_off = (unsigned)((long)kt * 64 * ld * 2);
    dma.issue(rs_kv, row_off + k_col, base, smem);
    dma.issue(rs_kv, row_off + v_col, base + KT, smem);
  };
  const int p16 = l & 15, g16 = l31 >> 4;
  const unsigned lds0 = (unsigned)(unsigned long)(lds_ptr_t)smem;
  unsigned v_off[DH / 32][2];
#pragma unroll
  for (int db = 0; db < DH / 32; ++db)
#pragma unroll
    for (int hf = 0; hf < 2; ++hf)
      v_off[db][hf] = lds0 + tile_addr<DH>(4 * hi + (p16 >> 2) + 8 * hf, db * 4 + g16 * 2 + ((p16 & 3) >> 1)) + (unsigned)((p16 & 1) * 8);

  const f32x16 zero16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
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
    // The softmax VALU work is kept as small as in the resident kernel below: the score scale is folded into the exponent
    // (p = exp2(fma(s, scale, -m)), the running max is taken on raw scores), O and the row sum are rescaled only when some
    // lane's maximum moved (wave-uniform branch), and the mask code exists only in the variant taken by boundary tiles
    // (sequence tail / causal diagonal) — the first version masked per element behind 32 wave-uniform branches per tile.
    // ONE body (separate instantiations per case make the O accumulators meet at a join and spill): the second key block
    // and the mask are small wave-uniform branches that only touch the score registers.
    f32x16 s[2];
    if (kb1) {
      // every K fragment of the step is requested before the first MFMA (64 registers at dh = 128): the read -> wait ->
      // MFMA chains of the first version exposed one LDS latency per MFMA (QK^T: 1250 cycles per step for 512 of matrix pipe)
      bf16x8 kfr[DH / 16][2];
#pragma unroll
      for (int ks = 0; ks < DH / 16; ++ks)
#pragma unroll
        for (int kb = 0; kb < 2; ++kb) kfr[ks][kb] = lds_frag128(smem, sK + tile_addr<DH>(kb * 32 + l31, ks * 2 + hi));
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int ks = 0; ks < DH / 16; ++ks)
#pragma unroll
        for (int kb = 0; kb < 2; ++kb) s[kb] = E16<F16>::mfma32(kfr[ks][kb], qf[ks], ks == 0 ? zero16 : s[kb]);
    } else {
#pragma unroll
      for (int ks = 0; ks < DH / 16; ++ks) {
        const bf16x8 kf = lds_frag128(smem, sK + tile_addr<DH>(l31, ks * 2 + hi));
        s[0] = E16<F16>::mfma32(kf, qf[ks], ks == 0 ? zero16 : s[0]);
      }
#pragma unroll
      for (int r = 0; r < 16; ++r) s[1][r] = -INFINITY;
    }
    if ((kv0 + 64 > S) || (CAUSAL && kv0 + 63 > q0 + w * 32)) {       // boundary tile: sequence tail / causal diagonal
      // key = kv0 + kb*32 + (r&3) + 8*(r>>2) + 4*hi is valid iff key <= min(S-1, qi): ONE per-lane limit compared with
      // compile-time constants (the per-element form kept 32 loop-invariant thresholds alive: +60 registers in the causal kernel)
      int lim = S - 1 - kv0 - 4 * hi;
      if (CAUSAL) lim = min(lim, qi - kv0 - 4 * hi);
#pragma unroll
      for (int kb = 0; kb < 2; ++kb)
#pragma unroll
        for (int r = 0; r < 16; ++r)
          if (kb * 32 + (r & 3) + 8 * (r >> 2) > lim) s[kb][r] = -INFINITY;
    }
    // V^T fragments (transpose reads) are requested one (key block, k half) group ahead of the MFMAs that consume them;
    // the first group is in flight during the softmax arithmetic.  Explicit counted waits: nothing else uses LDS in between.
    s16x4_t vlo[2][DH / 32], vhi[2][DH / 32];
#define HVLM_FWD_VREADS(SET, KB, SP)                                                                                   \
    do {                                                                                                                \
      const unsigned kbytes_ = sV + (unsigned)((KB) * 32 + 16 * (SP)) * (DH * 2);                                      \
      _Pragma("unroll") for (int db = 0; db < DH / 32; ++db)                                                            \
        asm volatile("ds_read_b64_tr_b16 %0, %2\n\tds_read_b64_tr_b16 %1, %3"                                         \
                     : "=&v"(vlo[SET][db]), "=&v"(vhi[SET][db]) : "v"(v_off[db][0] + kbytes_), "v"(v_off[db][1] + kbytes_) : "memory"); \
    } while (0)
    HVLM_FWD_VREADS(0, 0, 0);
    float mx = s[0][0];
#pragma unroll
    for (int kb = 0; kb < 2; ++kb)
#pragma unroll
      for (int r = 0; r < 16; ++r) mx = fmaxf(mx, s[kb][r]);
    mx = fmaxf(mx, xlane32(mx));
    const float mnew = fmaxf(m, mx * scale_log2);
    if (__builtin_amdgcn_ballot_w64(mnew > m)) {      // some lane's maximum moved (first tile: m = -inf -> alpha = 0)
      const float alpha = fast_exp2(m - mnew);
      lsum *= alpha;
#pragma unroll
      for (int db = 0; db < DH / 32; ++db)
#pragma unroll
        for (int r = 0; r < 16; ++r) o[db][r] *= alpha;
      m = mnew;
    }
    float psum0 = 0.f, psum1 = 0.f;
#pragma unroll
    for (int kb = 0; kb < 2; ++kb)
#pragma unroll
      for (int r = 0; r < 16; r += 2) {
        const float p0 = fast_exp2(__builtin_fmaf(s[kb][r], scale_log2, -m));
        const float p1 = fast_exp2(__builtin_fmaf(s[kb][r + 1], scale_log2, -m));
        s[kb][r] = p0; s[kb][r + 1] = p1;
        psum0 += p0; psum1 += p1;
      }
    lsum += psum0 + psum1;
    const int ngrp = kb1 ? 4 : 2;
#pragma unroll
    for (int gi = 0; gi < 4; ++gi) {
      if (gi >= ngrp) break;
      const int kb = gi >> 1, sp = gi & 1, cur = gi & 1;
      float pv[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) pv[j] = s[kb][8 * sp + j];
      const bf16x8 pf = pack8<F16>(pv);
      if (gi + 1 < ngrp) {
        if (cur == 0) HVLM_FWD_VREADS(1, (gi + 1) >> 1, (gi + 1) & 1);
        else HVLM_FWD_VREADS(0, (gi + 1) >> 1, (gi + 1) & 1);
        if constexpr (DH == 128) asm volatile("s_waitcnt lgkmcnt(8)" ::: "memory");
        else asm volatile("s_waitcnt lgkmcnt(4)" ::: "memory");
      } else {
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      }
#pragma unroll
      for (int db = 0; db < DH / 32; ++db) {
        s16x8_t vv = {vlo[cur][db][0], vlo[cur][db][1], vlo[cur][db][2], vlo[cur][db][3], vhi[cur][db][0], vhi[cur][db][1], vhi[cur][db][2], vhi[cur][db][3]};
        asm volatile("" : "+v"(vv));       // the MFMA below must stay behind the lgkmcnt wait
        o[db] = E16<F16>::mfma32(__builtin_bit_cast(bf16x8, vv), pf, o[db]);
      }
    }
#undef HVLM_FWD_VREADS
  }
  const float ltot = lsum + xlane32(lsum);
  __syncthreads();                         // everyone is done with the last K / V stage: the LDS is free for the output blocks
  {
    const int qw0 = q0 + w * 32;
    store_block_rows<DH, F16>(smem + w * (32 * DH * 2), o, 1.f / ltot, out + (tok0 + qw0) * (long)(H * DH) + h * DH, (long)H * DH,
                              S - qw0, l);
    if (qi < S && lse && hi == 0) lse[((long)b * H + h) * S_max + qi] = m * LN2 + logf(ltot);
  }
}
template <int DH, bool CAUSAL, int NW, bool F16 = false>
__global__ __launch_bounds__(NW * 64, 2) void attn_fwd_kernel(const bf16_t* __restrict__ qkv,
                                                       const bf16_t* __restrict__ vt,
                                                       bf16_t* __restrict__ out,
                                                       float* __restrict__ lse, int S_max, int H, int ld,
                                                       int S_pad, float scale_log2,
                                                       const int32_t* __restrict__ cu_seqlens) {
  attn_fwd_body<DH, CAUSAL, NW, F16>(qkv, vt, out, lse, S_max, H, ld, S_pad, scale_log2, cu_seqlens);
}

#ifdef HVLM_DEV
// =================================================================================================
// forward, software-pipelined inside each wave (round 4 EXPERIMENT, development builds only; bf16, the decoder's dh = 128 path):
//   iteration t:   S(t+1) = K(t+1) Q^T  (16 MFMAs)  ||  exp / row sum of S(t)  (VALU)   ->   O += V(t)^T P(t)
// The streaming kernel above runs QK^T, the softmax VALU and PV of one tile one after the other and relies on the second wave
// of the SIMD to fill the gaps — but the VALU stream of one wave and the MFMA stream of another serialise on the SIMD's issue
// port (tools/valu_mfma_probe.hip: 1213 cycles for two role-split waves against 881 for one wave carrying both streams).  Here the
// score MFMAs of the NEXT key tile sit in the same basic block as the exponentials of the current one (a second score set, 32
// registers), so one wave interleaves them.  LDS: K and V rings of two tiles each with K running one tile ahead of V (K(t+1) and
// V(t) in use, K(t+2) and V(t+1) in flight) — the same 64 KiB, still two workgroups per CU, one barrier per tile.
// Same arithmetic, same order per element as attn_fwd_body: bit-identical results (checked on the GPU).
// MEASURED AND NOT ADOPTED (profiles/r04g_attn_fwd_pipelined.txt): at two waves per SIMD (256 VGPRs) the second score set does not
// fit — O 64 + Q 32 + S 32 + S' 32 + K fragments 32 + V fragments 16-32 + addresses: 332-408 bytes of scratch with reloads through
// the whole loop; at one workgroup per CU (launch bounds (256, 1)) the compiler takes 356 registers, parks half of them in
// AccVGPRs (~200 v_accvgpr_read/write per tile) and still clusters the MFMAs ahead of the exponentials (sched_group_barrier
// requests are not honoured across the LDS-read waits): 343.6 us against 230.6 us for the shipped kernel at B = 16, S = 867, H = 32
// (451 vs 252 at S = 2048, H = 40).  The probe's 881-cycle arrangement needs a hand-allocated register file (assembly), not this.
template <int DH, bool CAUSAL, int NW>
__device__ __forceinline__ void attn_fwd_pipe_body(const bf16_t* __restrict__ qkv, bf16_t* __restrict__ out, float* __restrict__ lse,
                                                   int S_max, int H, int ld, float scale_log2, const int32_t* __restrict__ cu_seqlens) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr unsigned KT = 64 * DH * 2;   // one 64-key tile of K or V
  const int tid = threadIdx.x, l = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l31 = l & 31, hi = l >> 5;
  constexpr int QB = NW * 32;
  int bx, h, b;
  xcd_block(bx, h, b);
  const long tok0 = cu_seqlens ? (long)cu_seqlens[b] : (long)b * S_max;
  const int S = cu_seqlens ? (cu_seqlens[b + 1] - cu_seqlens[b]) : S_max;
  const int nqt = (S + QB - 1) / QB;
  if (bx >= nqt) return;
  const int qt = CAUSAL ? (nqt - 1 - bx) : bx;  // heavy tiles first
  const int q0 = qt * QB;
  const int qi = q0 + w * 32 + l31;
  const int qmax_w = q0 + w * 32 + 31;           // last query row of this wave

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

  TileDma<DH, NW> dma;
  dma.init(ld, w, l);
  const __amdgpu_buffer_rsrc_t rs_kv = sample_rsrc(qkv + tok0 * ld, S, ld);
  const unsigned k_col = (unsigned)(((long)H * DH + h * DH) * 2), v_col = k_col + (unsigned)((long)H * DH * 2);
  auto stage_k = [&](int kt) { dma.issue(rs_kv, (unsigned)((long)kt * 64 * ld * 2) + k_col, (unsigned)(kt & 1) * KT, smem); };
  auto stage_v = [&](int kt) { dma.issue(rs_kv, (unsigned)((long)kt * 64 * ld * 2) + v_col, 2 * KT + (unsigned)(kt & 1) * KT, smem); };
  const int p16 = l & 15, g16 = l31 >> 4;
  const unsigned lds0 = (unsigned)(unsigned long)(lds_ptr_t)smem;
  unsigned v_off[DH / 32][2];
#pragma unroll
  for (int db = 0; db < DH / 32; ++db)
#pragma unroll
    for (int hf = 0; hf < 2; ++hf)
      v_off[db][hf] = lds0 + tile_addr<DH>(4 * hi + (p16 >> 2) + 8 * hf, db * 4 + g16 * 2 + ((p16 & 3) >> 1)) + (unsigned)((p16 & 1) * 8);
  const f32x16 zero16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};

  // does this wave touch tile kt at all / its second 32-key block (wave-uniform)
  auto active = [&](int kt) { return kt < nkt && !(CAUSAL && kt * 64 > qmax_w); };
  auto has_kb1 = [&](int kt) { return (kt * 64 + 32 < S) && !(CAUSAL && kt * 64 + 32 > qmax_w); };
  // raw scores of tile kt from the K ring slot kt & 1, BOTH 32-key blocks, no branches (it has to sit in one basic block with the
  // exponentials it is interleaved with): a block past the sequence end / above the diagonal is computed on whatever the slot
  // holds and masked to -inf by the boundary code of its own iteration; two half-sweeps over the head dim, 32 fragment registers each
  auto qk_tile = [&](int kt, f32x16 (&sc)[2]) {
    const unsigned sK = (unsigned)(kt & 1) * KT;
#pragma unroll
    for (int hh = 0; hh < 2; ++hh) {
      bf16x8 kfr[DH / 32][2];
#pragma unroll
      for (int k2 = 0; k2 < DH / 32; ++k2)
#pragma unroll
        for (int kb = 0; kb < 2; ++kb) kfr[k2][kb] = lds_frag128(smem, sK + tile_addr<DH>(kb * 32 + l31, (hh * (DH / 32) + k2) * 2 + hi));
#pragma unroll
      for (int k2 = 0; k2 < DH / 32; ++k2)
#pragma unroll
        for (int kb = 0; kb < 2; ++kb) {
          const int ks = hh * (DH / 32) + k2;
          sc[kb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kfr[k2][kb], qf[ks], ks == 0 ? zero16 : sc[kb], 0, 0, 0);
        }
    }
  };

  // ---- prologue: K(0), V(0), K(1) in flight; S(0) computed before the loop ----
  stage_k(0);
  stage_v(0);
  if (nkt > 1) stage_k(1);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  f32x16 sn[2];                            // scores of the NEXT tile (set by the previous iteration)
  qk_tile(0, sn);                          // (tile 0 is active for every wave)
  (void)has_kb1;
  for (int kt = 0; kt < nkt; ++kt) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();                       // K(kt+1), V(kt) landed for everyone; everyone is past its reads of K(kt) and V(kt-1)
    if (kt + 2 < nkt) stage_k(kt + 2);     // -> the slot of K(kt)
    if (kt + 1 < nkt) stage_v(kt + 1);     // -> the slot of V(kt-1)
    if (!active(kt)) continue;             // (tiles ascend: once inactive, always inactive; the loop goes on for the barriers / DMA)
    const int kv0 = kt * 64;
    const unsigned sV = 2 * KT + (unsigned)(kt & 1) * KT;
    const bool kb1 = has_kb1(kt);
    f32x16 s[2];
    s[0] = sn[0]; s[1] = sn[1];
    if ((kv0 + 64 > S) || (CAUSAL && kv0 + 63 > q0 + w * 32)) {       // boundary tile: sequence tail / causal diagonal
      int lim = S - 1 - kv0 - 4 * hi;
      if (CAUSAL) lim = min(lim, qi - kv0 - 4 * hi);
#pragma unroll
      for (int kb = 0; kb < 2; ++kb)
#pragma unroll
        for (int r = 0; r < 16; ++r)
          if (kb * 32 + (r & 3) + 8 * (r >> 2) > lim) s[kb][r] = -INFINITY;
    }
    s16x4_t vlo[2][DH / 32], vhi[2][DH / 32];
#define HVLM_FWD_VREADS(SET, KB, SP)                                                                                   \
    do {                                                                                                                \
      const unsigned kbytes_ = sV + (unsigned)((KB) * 32 + 16 * (SP)) * (DH * 2);                                      \
      _Pragma("unroll") for (int db = 0; db < DH / 32; ++db)                                                            \
        asm volatile("ds_read_b64_tr_b16 %0, %2\n\tds_read_b64_tr_b16 %1, %3"                                         \
                     : "=&v"(vlo[SET][db]), "=&v"(vhi[SET][db]) : "v"(v_off[db][0] + kbytes_), "v"(v_off[db][1] + kbytes_) : "memory"); \
    } while (0)
    float mx = s[0][0];
#pragma unroll
    for (int kb = 0; kb < 2; ++kb)
#pragma unroll
      for (int r = 0; r < 16; ++r) mx = fmaxf(mx, s[kb][r]);
    mx = fmaxf(mx, xlane32(mx));
    const float mnew = fmaxf(m, mx * scale_log2);
    if (__builtin_amdgcn_ballot_w64(mnew > m)) {
      const float alpha = fast_exp2(m - mnew);
      lsum *= alpha;
#pragma unroll
      for (int db = 0; db < DH / 32; ++db)
#pragma unroll
        for (int r = 0; r < 16; ++r) o[db][r] *= alpha;
      m = mnew;
    }
    // ---- the pipelined block: score MFMAs of tile kt+1 beside the exponentials of tile kt (ONE basic block; the last
    // iteration's extra tile reads a stale slot and is never used) ----
    __builtin_amdgcn_sched_barrier(0);
    qk_tile(kt + 1, sn);
    float psum0 = 0.f, psum1 = 0.f;
#pragma unroll
    for (int kb = 0; kb < 2; ++kb)
#pragma unroll
      for (int r = 0; r < 16; r += 2) {
        const float p0 = fast_exp2(__builtin_fmaf(s[kb][r], scale_log2, -m));
        const float p1 = fast_exp2(__builtin_fmaf(s[kb][r + 1], scale_log2, -m));
        s[kb][r] = p0; s[kb][r + 1] = p1;
        psum0 += p0; psum1 += p1;
      }
    lsum += psum0 + psum1;
    __builtin_amdgcn_sched_barrier(0);
    HVLM_FWD_VREADS(0, 0, 0);
    const int ngrp = kb1 ? 4 : 2;
#pragma unroll
    for (int gi = 0; gi < 4; ++gi) {
      if (gi >= ngrp) break;
      const int kb = gi >> 1, sp = gi & 1, cur = gi & 1;
      float pv[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) pv[j] = s[kb][8 * sp + j];
      const bf16x8 pf = pack8<false>(pv);
      if (gi + 1 < ngrp) {
        if (cur == 0) HVLM_FWD_VREADS(1, (gi + 1) >> 1, (gi + 1) & 1);
        else HVLM_FWD_VREADS(0, (gi + 1) >> 1, (gi + 1) & 1);
        if constexpr (DH == 128) asm volatile("s_waitcnt lgkmcnt(8)" ::: "memory");
        else asm volatile("s_waitcnt lgkmcnt(4)" ::: "memory");
      } else {
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      }
#pragma unroll
      for (int db = 0; db < DH / 32; ++db) {
        s16x8_t vv = {vlo[cur][db][0], vlo[cur][db][1], vlo[cur][db][2], vlo[cur][db][3], vhi[cur][db][0], vhi[cur][db][1], vhi[cur][db][2], vhi[cur][db][3]};
        asm volatile("" : "+v"(vv));
        o[db] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, vv), pf, o[db], 0, 0, 0);
      }
    }
#undef HVLM_FWD_VREADS
  }
  const float ltot = lsum + xlane32(lsum);
  __syncthreads();
  {
    const int qw0 = q0 + w * 32;
    store_block_rows<DH, false>(smem + w * (32 * DH * 2), o, 1.f / ltot, out + (tok0 + qw0) * (long)(H * DH) + h * DH, (long)H * DH,
                                S - qw0, l);
    if (qi < S && lse && hi == 0) lse[((long)b * H + h) * S_max + qi] = m * LN2 + logf(ltot);
  }
}
template <int DH, bool CAUSAL, int NW>
__global__ __launch_bounds__(NW * 64, 1) void attn_fwd_pipe_kernel(const bf16_t* __restrict__ qkv, bf16_t* __restrict__ out,
                                                            float* __restrict__ lse, int S_max, int H, int ld, float scale_log2,
                                                            const int32_t* __restrict__ cu_seqlens) {
  attn_fwd_pipe_body<DH, CAUSAL, NW>(qkv, out, lse, S_max, H, ld, scale_log2, cu_seqlens);
}

#endif  // HVLM_DEV (attn_fwd_pipe)

// =================================================================================================
// forward, short non-causal sequences (CLIP: S = 257, dh = 64): K and V of one (batch, head) stay RESIDENT in LDS.
//   One workgroup = one (batch, head), one wave per 32 queries (S = 257 -> 9 waves), K and V ([32*NKB rows][DH], row
//   major, 16-B chunks XOR-swizzled by row) fetched ONCE by LDS-DMA, one barrier, no staging loop.  V is consumed
//   through `ds_read_b64_tr_b16` (LDS transpose read: a 16-lane group reads a 4(key) x 16(d) block, lane q receives
//   column d = q for the 4 keys), so the V^T pre-pass of the streaming kernel and its workspace traffic disappear.
//   The streaming kernel pads S = 257 to 3 x 128 queries x 5 x 64 keys (1.86x the useful MFMA work); here the padding is
//   9 x 32 by 9 x 32 (1.26x).  Same online-softmax body, 64 keys per step.
// =================================================================================================

template <int DH, bool F16 = false, int SC = 0>
__global__ __launch_bounds__(640, 3) void attn_fwd_resident_kernel(const bf16_t* __restrict__ qkv,
                                                                   bf16_t* __restrict__ out,
                                                                   float* __restrict__ lse, int S, int H, int ld,
                                                                   int nkb, int nitems, float scale_log2) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, l = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int nw = (int)blockDim.x >> 6;
  const int l31 = l & 31, hi = l >> 5;
  const int rows = nkb * 32;
  const unsigned tile_bytes = (unsigned)rows * DH * 2;      // one of K / V
  const unsigned buf_bytes = 2 * tile_bytes;                // K + V of one (batch, head)
  constexpr int CPR = DH / 8;          // 16-byte chunks per row
  constexpr int RPI = 64 / CPR;        // rows per wave-instruction (1 KiB piece)
  constexpr unsigned ROWB = DH * 2;    // bytes per LDS row
  const int qi = w * 32 + l31;

  // PERSISTENT workgroup, two K/V buffers, one extra LOADER wave: the LDS-DMA of item i+1 is issued (by the loader) while
  // the compute waves work on item i, so the fill latency is hidden.  The loader is a separate wave because the compiler
  // guards LDS reads it cannot disambiguate (the ds_read_b64_tr_b16 of V) with a vmcnt wait for every LDS-DMA still
  // pending IN THE SAME WAVE: issued by the compute waves, the prefetch was drained before the first P*V of each item.
  auto issue_kv = [&](int item, unsigned buf) {
    const int b = item / H, h = item - b * H;
    const bf16_t* kbase = qkv + (long)b * S * ld + (long)H * DH + h * DH;
    const bf16_t* vbase = kbase + (long)H * DH;
    const int npiece = rows / RPI;
    for (int pi = 0; pi < npiece; ++pi) {              // the loader wave alone (see below)
      const int r = pi * RPI + l / CPR;
      const int c = (l % CPR) ^ swz_bits<DH>(r);
      const long gr = (long)min(r, S - 1) * ld + c * 8;
      glds16(kbase, (unsigned)(gr * 2), buf + (unsigned)pi * 1024u, smem);
      glds16(vbase, (unsigned)(gr * 2), buf + tile_bytes + (unsigned)pi * 1024u, smem);
    }
  };
  auto load_q = [&](int item, bf16x8 (&q)[DH / 16]) {
    const int b = item / H, h = item - b * H;
    const bf16_t* qp = qkv + ((long)b * S + min(qi, S - 1)) * ld + h * DH + hi * 8;
#pragma unroll
    for (int ks = 0; ks < DH / 16; ++ks) q[ks] = *reinterpret_cast<const bf16x8*>(qp + ks * 16);
  };

  // The loop is VALU-bound (32 quarter-rate v_exp_f32 per 16 MFMAs), so everything else is kept off the VALU:
  //  * per-lane LDS offsets are loop invariants (the row swizzle only sees lane bits: key-block bases are multiples
  //    of 16 rows), the key block enters as ONE add per step + instruction offsets;
  //  * the score scale is folded into the exponent: p = exp2(fma(s, scale, -m)), the running max is taken on raw scores;
  //  * O and the row sum are rescaled only when some lane's maximum moved (wave-uniform branch);
  //  * the tail mask exists only in the code of the last, partial key tile; accumulators start from an inline 0.
  // V fragments: lane p of a 16-lane group points at key (p>>2), d-chunk (p&3)*4 of a 4(key) x 16(d) block and
  // receives d = db*32 + l31 for the 4 keys (ds_read_b64_tr_b16); two reads = the 8 k-slots of the P registers.
  const int p16 = l & 15, g16 = l31 >> 4;
  unsigned k_off[DH / 16], v_off[DH / 32][2];
#pragma unroll
  for (int ks = 0; ks < DH / 16; ++ks) k_off[ks] = tile_addr<DH>(l31, ks * 2 + hi);
#pragma unroll
  for (int db = 0; db < DH / 32; ++db)
#pragma unroll
    for (int hf = 0; hf < 2; ++hf)
      v_off[db][hf] = tile_bytes + tile_addr<DH>(4 * hi + (p16 >> 2) + 8 * hf, db * 4 + g16 * 2 + ((p16 & 3) >> 1)) + (unsigned)((p16 & 1) * 8);
  const f32x16 zero16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};

  const bool loader = (w == nw - 1);                   // wave-uniform
  bf16x8 qf[DH / 16], qn[DH / 16];
  int item = blockIdx.x;
  if (loader) issue_kv(item, 0);
  else load_q(item, qf);
#ifdef HVLM_ATTN_PROF
  long long t_bar = 0, t_body = 0, t_epi = 0, t_qk = 0, t_sm = 0, t_pv = 0, t_all0 = __builtin_readcyclecounter();
#endif
  for (unsigned it = 0; item < nitems; ++it, item += (int)gridDim.x) {
    const unsigned base = (it & 1u) * buf_bytes;
#ifdef HVLM_ATTN_PROF
    const long long c0 = __builtin_readcyclecounter();
#endif
    // Loader: K/V of this item landed (a real S_WAITCNT, not inline asm: the compiler's wait-count pass must see it).
    // Compute waves wait for nothing here: their next Q fragments were moved into place before the output stores of
    // the previous item were issued, so those stores (write acknowledgements) are never waited for.
    if (loader) __builtin_amdgcn_s_waitcnt(0x0F70);            // vmcnt(0)
    __syncthreads();                                     // everyone is done with the other buffer
#ifdef HVLM_ATTN_PROF
    const long long c1 = __builtin_readcyclecounter();
    t_bar += c1 - c0;
#endif
    const int nitem = item + (int)gridDim.x;
    if (loader) {
      if (nitem < nitems) issue_kv(nitem, buf_bytes - base);
      continue;
    }
    if (nitem < nitems) load_q(nitem, qn);
    f32x16 o[DH / 32];
#pragma unroll
    for (int db = 0; db < DH / 32; ++db)
#pragma unroll
      for (int r = 0; r < 16; ++r) o[db][r] = 0.f;
    float m = -INFINITY, lsum = 0.f;

    auto v_frag = [&](int db, unsigned key_bytes) -> bf16x8 {
      const s16x4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4_t*)(smem + base + v_off[db][0] + key_bytes));
      const s16x4_t hi4 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4_t*)(smem + base + v_off[db][1] + key_bytes));
      const s16x8_t v = {lo[0], lo[1], lo[2], lo[3], hi4[0], hi4[1], hi4[2], hi4[3]};
      return __builtin_bit_cast(bf16x8, v);
    };
    auto tile_body = [&](int kv0, auto nkb_tag, auto mask_tag) {
      constexpr int NKB = decltype(nkb_tag)::value;
      constexpr bool MASK = decltype(mask_tag)::value;
      const unsigned kvb = (unsigned)kv0 * ROWB;
      f32x16 sc[NKB];
#pragma unroll
      for (int ks = 0; ks < DH / 16; ++ks)
#pragma unroll
        for (int kb = 0; kb < NKB; ++kb) {
          const bf16x8 kf = lds_frag128(smem, base + k_off[ks] + kvb + (unsigned)(kb * 32) * ROWB);
          sc[kb] = E16<F16>::mfma32(kf, qf[ks], ks == 0 ? zero16 : sc[kb]);
        }
      if constexpr (MASK) {
#pragma unroll
        for (int kb = 0; kb < NKB; ++kb)
#pragma unroll
          for (int r = 0; r < 16; ++r)
            if (kv0 + kb * 32 + crow(r, hi) >= S) sc[kb][r] = -INFINITY;
      }
      float mx = sc[0][0];
#pragma unroll
      for (int kb = 0; kb < NKB; ++kb)
#pragma unroll
        for (int r = 0; r < 16; ++r) mx = fmaxf(mx, sc[kb][r]);
      mx = fmaxf(mx, xlane32(mx));
      const float mnew = fmaxf(m, mx * scale_log2);
      if (__builtin_amdgcn_ballot_w64(mnew > m)) {      // some lane's maximum moved: rescale (first tile: m = -inf -> 0)
        const float alpha = fast_exp2(m - mnew);
        lsum *= alpha;
#pragma unroll
        for (int db = 0; db < DH / 32; ++db)
#pragma unroll
          for (int r = 0; r < 16; ++r) o[db][r] *= alpha;
        m = mnew;
      }
      float psum0 = 0.f, psum1 = 0.f;
#pragma unroll
      for (int kb = 0; kb < NKB; ++kb)
#pragma unroll
        for (int r = 0; r < 16; r += 2) {
          const float p0 = fast_exp2(__builtin_fmaf(sc[kb][r], scale_log2, -m));
          const float p1 = fast_exp2(__builtin_fmaf(sc[kb][r + 1], scale_log2, -m));
          sc[kb][r] = p0; sc[kb][r + 1] = p1;
          psum0 += p0; psum1 += p1;
        }
      lsum += psum0 + psum1;
#pragma unroll
      for (int kb = 0; kb < NKB; ++kb)
#pragma unroll
        for (int sp = 0; sp < 2; ++sp) {
          float pv[8];
#pragma unroll
          for (int j = 0; j < 8; ++j) pv[j] = sc[kb][8 * sp + j];
          const bf16x8 pf = pack8<F16>(pv);
#pragma unroll
          for (int db = 0; db < DH / 32; ++db)
            o[db] = E16<F16>::mfma32(v_frag(db, kvb + (unsigned)(kb * 32 + 16 * sp) * ROWB), pf, o[db]);
        }
    };
    // SC > 0 (the sequence length is a compile-time constant: CLIP's 257): the key blocks are unrolled and software-pipelined BY HAND,
    // across blocks and around the LDS latency.  Two measurements shaped this (profiles/r03k_clip_attention_lds_pipeline.txt):
    //  * left to the compiler (tile_body above, 168-register budget) every ds_read sits right in front of the MFMA that consumes
    //    it: sixteen exposed LDS round trips per 64-key tile.  Here every batch of fragment reads is requested a phase early;
    //    with the block index a constant all LDS addresses are (4 + 4 per-item registers) + instruction offsets.
    //  * the two waves of a SIMD do not hide each other: while one wave's MFMAs occupy the matrix pipe the other's VALU does not
    //    issue, a SIMD's time is the SUM of its waves' MFMA and VALU phases.  Matrix and vector work overlap only inside ONE
    //    instruction stream — so each phase below carries, for 32-key blocks t-1, t, t+1:
    //        P V of block t-1  |  softmax of block t (VALU)  |  QK^T of block t+1
    //    on two score register sets and two P sets.  The rescale of O by exp2(m_old - m_new) is deferred to the start of the next
    //    phase (P V of block t-1 still accumulates at the old maximum) and stays a wave-uniform branch.
    unsigned ka[DH / 16], va[DH / 32][2];
    if constexpr (SC > 0) {
#pragma unroll
      for (int ks = 0; ks < DH / 16; ++ks) ka[ks] = base + k_off[ks];
#pragma unroll
      for (int db = 0; db < DH / 32; ++db) { va[db][0] = base + v_off[db][0]; va[db][1] = base + v_off[db][1]; }
    }
    constexpr int NT = (SC + 31) / 32;                 // 32-key blocks
    f32x16 sA, sB;                                     // scores of the even / odd blocks
    bf16x8 kfr[DH / 16], vfr[2][DH / 32], pA[2], pB[2];
    float alpha_p = 1.f;
    bool pend = false;
    auto k_read = [&](auto t_tag) {                    // K fragments of block t (one ds_read_b128 per 16 head dims)
      constexpr unsigned off = (unsigned)(decltype(t_tag)::value * 32) * ROWB;
#pragma unroll
      for (int ks = 0; ks < DH / 16; ++ks) kfr[ks] = lds_frag128(smem, ka[ks] + off);
    };
    auto v_read = [&](auto t_tag) {                    // V^T fragments of block t
      constexpr int t = decltype(t_tag)::value;
#pragma unroll
      for (int sp = 0; sp < 2; ++sp)
#pragma unroll
        for (int db = 0; db < DH / 32; ++db) {
          constexpr unsigned dummy = 0; (void)dummy;
          const unsigned key_bytes = (unsigned)(t * 32 + 16 * sp) * ROWB;
          const s16x4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4_t*)(smem + va[db][0] + key_bytes));
          const s16x4_t hi4 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4_t*)(smem + va[db][1] + key_bytes));
          const s16x8_t v = {lo[0], lo[1], lo[2], lo[3], hi4[0], hi4[1], hi4[2], hi4[3]};
          vfr[sp][db] = __builtin_bit_cast(bf16x8, v);
        }
    };
    auto qk = [&](f32x16& sx) {
#pragma unroll
      for (int ks = 0; ks < DH / 16; ++ks) sx = E16<F16>::mfma32(kfr[ks], qf[ks], ks == 0 ? zero16 : sx);
    };
    auto pv = [&](const bf16x8 (&pf)[2]) {
#pragma unroll
      for (int sp = 0; sp < 2; ++sp)
#pragma unroll
        for (int db = 0; db < DH / 32; ++db) o[db] = E16<F16>::mfma32(vfr[sp][db], pf[sp], o[db]);
    };
    auto rescale = [&]() {
      if (pend) {
#pragma unroll
        for (int db = 0; db < DH / 32; ++db)
#pragma unroll
          for (int r = 0; r < 16; ++r) o[db][r] *= alpha_p;
      }
    };
    auto softmax_blk = [&](auto t_tag, f32x16& sx, bf16x8 (&pf)[2]) {
      constexpr int t = decltype(t_tag)::value;
      if constexpr (t * 32 + 32 > SC) {
#pragma unroll
        for (int r = 0; r < 16; ++r)
          if (t * 32 + crow(r, hi) >= SC) sx[r] = -INFINITY;
      }
      float mx = sx[0];
#pragma unroll
      for (int r = 1; r < 16; ++r) mx = fmaxf(mx, sx[r]);
      mx = fmaxf(mx, xlane32(mx));
      const float mnew = fmaxf(m, mx * scale_log2);
      if constexpr (t > 0) {                           // (block 0: O and the row sum are still zero)
        alpha_p = fast_exp2(m - mnew);
        pend = __builtin_amdgcn_ballot_w64(mnew > m) != 0;
        lsum *= alpha_p;
      }
      m = mnew;
      // P is rounded to 16 bits pair by pair and the row sum is taken over the ROUNDED values (one v_dot2c per pair): the fp32
      // probabilities die at the conversion, and the normaliser matches the P that enters P V exactly
      float psum0 = 0.f, psum1 = 0.f;
      u32x4 pk[2];
#pragma unroll
      for (int sp = 0; sp < 2; ++sp)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const int r = 8 * sp + 2 * j;
          const float p0 = fast_exp2(__builtin_fmaf(sx[r], scale_log2, -m));
          const float p1 = fast_exp2(__builtin_fmaf(sx[r + 1], scale_log2, -m));
          const unsigned int pp2 = E16<F16>::pack2(p0, p1);
          pk[sp][j] = pp2;
          if (j & 1) psum1 = E16<F16>::add_pair(pp2, psum1);
          else psum0 = E16<F16>::add_pair(pp2, psum0);
        }
      lsum += psum0 + psum1;
      // (pinned: LLVM otherwise SINKS this block's exponentials into the next phase's basic block — past its rescale branch —
      // next to their first use, and the phase boundary is gone)
      asm volatile("" : "+v"(pk[0]), "+v"(pk[1]), "+v"(lsum), "+v"(m));
      pf[0] = __builtin_bit_cast(bf16x8, pk[0]);
      pf[1] = __builtin_bit_cast(bf16x8, pk[1]);
    };
    auto phase = [&](auto t_tag) {
      constexpr int t = decltype(t_tag)::value;
      f32x16& cur = (t & 1) ? sB : sA;
      f32x16& nxt = (t & 1) ? sA : sB;
      bf16x8 (&pc)[2] = (t & 1) ? pB : pA;
      bf16x8 (&pp)[2] = (t & 1) ? pA : pB;
      __builtin_amdgcn_sched_barrier(0);
      if constexpr (t > 1) rescale();                  // by the factor block t-1 found, before its P V joins O
      // (program order: everything that is not the softmax first — the pin at the softmax's end orders the whole phase)
      if constexpr (t > 0) pv(pp);                     // block t-1
      if constexpr (t + 1 < NT) qk(nxt);               // block t+1
      if constexpr (t > 0) v_read(t_tag);              // V^T of block t (into the registers P V of block t-1 was issued from)
      if constexpr (t + 2 < NT) k_read(std::integral_constant<int, t + 2>{});
      softmax_blk(t_tag, cur, pc);
      // (the order inside the phase is the compiler's: forcing the MFMAs evenly through the VALU stream with sched_group_barrier,
      // 1 MFMA : 5 / 7 / 9 / 12 VALU, measured 952-972 us per layer against 943 without, profiles/r03k)
    };
    // The LAST compute wave when it holds ONE useful query (S = 32 (nw - 2) + 1: CLIP's 257 tokens = 8 x 32 + 1).  In the layout
    // above it would run the whole softmax — 16 score registers per key block, 144 exponentials per lane — for a single valid
    // lane pair, and it is the third wave of its SIMD: the kernel is VALU-issue-bound (DESIGN.md §3 "Attention, round 3"), so that
    // SIMD, and with it every item, took 3 wave-loads where the others carry 2.  With the MFMA operands SWAPPED (S = Q K^T: A = the
    // query fragment, all 32 rows the same clamped query, B = a K fragment) the LANES are the keys and every accumulator register
    // holds the same row: one score register per 32-key block, 9 exponentials per lane instead of 144.  All nine blocks' scores
    // fit in registers, so the softmax is a plain two-pass one (wave-wide max, then exp and sum); P goes through a 1-KiB
    // wave-private LDS strip to reach the k-slot order of the V^T fragments, and P V is the usual MFMA (every column = the one query).
    const bool solo = (w == nw - 2) && (S - w * 32 == 1);
    if (solo) {
      // (the third wave of its SIMD and the youngest: with oldest-first issue it was served last and finished LAST — 16.3k cycles
      // for a tenth of a normal wave's work, profiles/r03k; at raised priority it is out of the way early)
      __builtin_amdgcn_s_setprio(3);
      float s9[9];
#pragma unroll
      for (int kb = 0; kb < 9; ++kb) {
        s9[kb] = -INFINITY;
        if (kb < nkb) {
          f32x16 sx;
#pragma unroll
          for (int ks = 0; ks < DH / 16; ++ks) {
            const bf16x8 kf = lds_frag128(smem, base + k_off[ks] + (unsigned)(kb * 32) * ROWB);
            sx = E16<F16>::mfma32(qf[ks], kf, ks == 0 ? zero16 : sx);
          }
          s9[kb] = (kb * 32 + l31 < S) ? sx[0] : -INFINITY;         // lane l31 (both half-waves alike) = key kb*32 + l31
        }
      }
      float mx = s9[0];
#pragma unroll
      for (int kb = 1; kb < 9; ++kb) mx = fmaxf(mx, s9[kb]);
      m = wave_max_dpp(mx) * scale_log2;
      char* strip = smem + 2 * buf_bytes;                            // [nkb * 32] 16-bit probabilities, key order
      float ls = 0.f;
#pragma unroll
      for (int kb = 0; kb < 9; ++kb) {
        if (kb < nkb) {
          const float pk = fast_exp2(__builtin_fmaf(s9[kb], scale_log2, -m));
          ls += pk;
          if (hi == 0) reinterpret_cast<bf16_t*>(strip)[kb * 32 + l31] = E16<F16>::from_f(pk);
        }
      }
      lsum = wave_sum_dpp(ls) * 0.25f;        // both half-waves hold every key once: the wave-wide sum counts each probability twice, and
                                              // the epilogue adds the other half-wave's lsum again (ltot = lsum + xlane32(lsum))
#pragma unroll
      for (int kb = 0; kb < 9; ++kb) {
        if (kb < nkb) {
#pragma unroll
          for (int sp = 0; sp < 2; ++sp) {
            // k-slot j of this lane = key kb*32 + crow(8 sp + j, hi) = b + {0,1,2,3, 8,9,10,11}, b = kb*32 + 16 sp + 4 hi (the order
            // in which the V^T fragments hold their keys, see v_off)
            const unsigned b = (unsigned)(kb * 32 + 16 * sp + 4 * hi) * 2u;
            const bf16x8 pf = lds_frag64x2(strip, b, b + 16u);
#pragma unroll
            for (int db = 0; db < DH / 32; ++db)
              o[db] = E16<F16>::mfma32(v_frag(db, (unsigned)(kb * 32 + 16 * sp) * ROWB), pf, o[db]);
          }
        }
      }
    } else
    {
      if constexpr (SC > 0) {
        static_assert(SC == 0 || (SC > 256 && SC <= 288), "nine key blocks, the last one partial");
        k_read(std::integral_constant<int, 0>{});
        v_read(std::integral_constant<int, 0>{});
        qk(sA);
        k_read(std::integral_constant<int, 1>{});
        phase(std::integral_constant<int, 0>{});
        phase(std::integral_constant<int, 1>{});
        phase(std::integral_constant<int, 2>{});
        phase(std::integral_constant<int, 3>{});
        phase(std::integral_constant<int, 4>{});
        phase(std::integral_constant<int, 5>{});
        phase(std::integral_constant<int, 6>{});
        phase(std::integral_constant<int, 7>{});
        phase(std::integral_constant<int, 8>{});
        __builtin_amdgcn_sched_barrier(0);
        rescale();
        pv((NT - 1) & 1 ? pB : pA);
      } else {
      const int nfull = S / 64;                     // key tiles without a tail
      int kb0 = 0;
      for (; kb0 < 2 * nfull; kb0 += 2) tile_body(kb0 * 32, std::integral_constant<int, 2>{}, std::false_type{});
      if (kb0 + 1 < nkb) tile_body(kb0 * 32, std::integral_constant<int, 2>{}, std::true_type{});
      else if (kb0 < nkb) tile_body(kb0 * 32, std::integral_constant<int, 1>{}, std::true_type{});
      }
    }
#ifdef HVLM_ATTN_PROF
    const long long c2 = __builtin_readcyclecounter();
    t_body += c2 - c1;
#endif
    const float ltot = lsum + xlane32(lsum);
    if (nitem < nitems) {                    // next item's Q: in place BEFORE the stores below (see the loop top)
#pragma unroll
      for (int ks = 0; ks < DH / 16; ++ks) qf[ks] = qn[ks];
    }
    if (qi < S) {
      const int b = item / H, h = item - b * H;
      const float inv = __builtin_amdgcn_rcpf(ltot);
      bf16_t* op = out + ((long)b * S + qi) * (long)(H * DH) + h * DH;
#pragma unroll
      for (int db = 0; db < DH / 32; ++db)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const int d = db * 32 + 8 * g + 4 * hi;
          u32x2 v = {E16<F16>::pack2(o[db][4 * g] * inv, o[db][4 * g + 1] * inv),
                     E16<F16>::pack2(o[db][4 * g + 2] * inv, o[db][4 * g + 3] * inv)};
          *reinterpret_cast<u32x2*>(op + d) = v;
        }
#ifndef HVLM_ATTN_PROF
      if (lse && hi == 0) lse[((long)b * H + h) * S + qi] = m * LN2 + logf(ltot);
#endif
    }
#ifdef HVLM_ATTN_PROF
    t_epi += __builtin_readcyclecounter() - c2;
#endif
  }
#ifdef HVLM_ATTN_PROF
  // (development: -DHVLM_ATTN_PROF turns the `lse` argument into 8 cycle counters per wave — barrier wait, tile loop, epilogue,
  // total, and the QK^T / softmax / P V phases of the full tiles — read by tools/clip_attn_phases.py)
  if (lse && l == 0) {
    long long* pr = reinterpret_cast<long long*>(lse) + ((long)blockIdx.x * 10 + w) * 8;
    pr[0] = t_bar; pr[1] = t_body; pr[2] = t_epi; pr[3] = __builtin_readcyclecounter() - t_all0;
    pr[4] = t_qk; pr[5] = t_sm; pr[6] = t_pv; pr[7] = 0;
  }
#endif
}

// delta[b,h,s] = sum_d O[b,s,h,d] * dO[b,s,h,d]   (one thread per 8 elements, dh/8-lane reduce).  Rows are token-major
// [tokens, H*DH]; `row_b` / `row_s` (varlen: sample and position of every packed row) may be NULL (b = row / S, s = row % S).
template <int DH>
__global__ __launch_bounds__(256) void attn_delta_kernel(const bf16_t* __restrict__ o,
                                                         const bf16_t* __restrict__ dout,
                                                         float* __restrict__ delta, long tokens, int S,
                                                         int H, const int32_t* __restrict__ cu_seqlens, int B) {
  constexpr int LPH = DH / 8;  // lanes per (token, head)
  const long total = tokens * H * LPH;
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
    long b; int s_;
    if (cu_seqlens) {                      // binary search of the sample that owns packed row `tok`
      int lo = 0, hi_ = B;
      while (hi_ - lo > 1) { const int mid = (lo + hi_) >> 1; if ((long)cu_seqlens[mid] <= tok) lo = mid; else hi_ = mid; }
      b = lo; s_ = (int)(tok - cu_seqlens[lo]);
    } else {
      b = tok / S; s_ = (int)(tok % S);
    }
    delta[(b * H + h) * S + s_] = acc;
  }
}

// =================================================================================================
// backward, dQ  (one query per lane)
// =================================================================================================
template <int DH, bool CAUSAL>
__global__ __launch_bounds__(256, 2) void attn_bwd_dq_kernel(
    const bf16_t* __restrict__ qkv, const bf16_t* __restrict__ out_g /*O [tokens, H*DH], or NULL: delta is read, not computed*/,
    const bf16_t* __restrict__ dout, const float* __restrict__ lse, float* __restrict__ delta,
    bf16_t* __restrict__ dqkv, int S_max, int H, int ld, int S_pad, float scale,
    const int32_t* __restrict__ cu_seqlens, const bf16_t* __restrict__ rope_cos, const bf16_t* __restrict__ rope_sin,
    const int32_t* __restrict__ rope_pos) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr unsigned KT = 64 * DH * 2;
  const int tid = threadIdx.x, l = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l31 = l & 31, hi = l >> 5;
  int bx, h, b;
  xcd_block(bx, h, b);
  const long tok0 = cu_seqlens ? (long)cu_seqlens[b] : (long)b * S_max;       // varlen: see attn_fwd_kernel
  const int S = cu_seqlens ? (cu_seqlens[b + 1] - cu_seqlens[b]) : S_max;
  const int nqt = (S + 127) / 128;
  if (bx >= nqt) return;
  const int qt = CAUSAL ? (nqt - 1 - bx) : bx;
  const int q0 = qt * 128;
  const int qi = q0 + w * 32 + l31;
  const int qc = min(qi, S - 1);
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
  const float lse2 = lse[((long)b * H + h) * S_max + qc] * LOG2E;
  // delta[b, h, s] = sum_d O[s, d] dO[s, d]: this lane and its lane^32 partner hold the two halves of the row's dO already, so the
  // separate pass over O and dO (attn_delta_kernel: 44 us per layer at the SFT shape) folds into 8 loads of O and 64 FMAs here;
  // the row's value is written out for the dK/dV kernel, which is launched after this one.
  float dlt;
  if (out_g) {
    const bf16_t* op = out_g + (tok0 + qc) * (long)(H * DH) + h * DH + hi * 8;
    float part = 0.f;
#pragma unroll
    for (int ks = 0; ks < DH / 16; ++ks) {
      const u32x4 ov = *reinterpret_cast<const u32x4*>(op + ks * 16);
      const u32x4 gv = __builtin_bit_cast(u32x4, dof[ks]);
#pragma unroll
      for (int e = 0; e < 4; ++e) part += bf_lo(ov[e]) * bf_lo(gv[e]) + bf_hi(ov[e]) * bf_hi(gv[e]);
    }
    dlt = part + xlane32(part);
    if (qi < S && hi == 0) delta[((long)b * H + h) * S_max + qi] = dlt;
  } else {
    dlt = delta[((long)b * H + h) * S_max + qc];
  }

  f32x16 dq[DH / 32];
#pragma unroll
  for (int db = 0; db < DH / 32; ++db)
#pragma unroll
    for (int r = 0; r < 16; ++r) dq[db][r] = 0.f;

  const int nkt_all = (S + 63) / 64;
  const int nkt = CAUSAL ? min(nkt_all, min(q0 + 127, S - 1) / 64 + 1) : nkt_all;
  const bf16_t* kbase = qkv + tok0 * ld + (long)H * DH + h * DH;
  const bf16_t* vbase = qkv + tok0 * ld + 2L * H * DH + h * DH;

  // two LDS stages: the tiles of step kt+1 are in flight while step kt is computed (the single-stage version exposed
  // the full fill latency on every step).  A stage holds K and V only: the K^T fragments of dQ^T = K^T dS^T are gathered
  // from the row-major K tile with `ds_read_b64_tr_b16` (a 16-lane group reads a 4(key) x 16(d) block, lane q receives
  // d = q) — no K^T tile, no K^T pre-pass, 64 KiB of LDS = two workgroups per CU.  Those reads are inline asm: the
  // builtin is modelled as a may-store LDS access and gets a compiler `vmcnt(0)` while the prefetch is pending.
  constexpr unsigned STAGE = 2 * KT;
  auto stage = [&](int buf, int kt) {
    const unsigned o = (unsigned)buf * STAGE;
    load_tile<64, DH>(smem, o, kbase, ld, kt * 64, S, w, l);
    load_tile<64, DH>(smem, o + KT, vbase, ld, kt * 64, S, w, l);
  };
  const int p16 = l & 15, g16 = l31 >> 4;
  const unsigned lds0 = (unsigned)(unsigned long)(lds_ptr_t)smem;
  unsigned kt_off[DH / 32][2];          // lane offsets of the two 4-key groups (keys +0..3 / +8..11 of its half) per d block
#pragma unroll
  for (int db = 0; db < DH / 32; ++db)
#pragma unroll
    for (int hf = 0; hf < 2; ++hf)
      kt_off[db][hf] = lds0 + tile_addr<DH>(4 * hi + (p16 >> 2) + 8 * hf, db * 4 + g16 * 2 + ((p16 & 3) >> 1)) + (unsigned)((p16 & 1) * 8);
  stage(0, 0);
  for (int kt = 0; kt < nkt; ++kt) {
    const int kv0 = kt * 64;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();   // tile kt landed for everyone; everyone is done with tile kt-1
    if (kt + 1 < nkt) stage((kt + 1) & 1, kt + 1);
    if (CAUSAL && kv0 > q0 + w * 32 + 31) continue;
    const unsigned sK = (unsigned)(kt & 1) * STAGE, sV = sK + KT;

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
    // dS^T = P o (dP - delta); the softmax scale is applied once to dQ at the end.  Boundary tiles (sequence tail / causal
    // diagonal) first overwrite the masked scores with -inf (p = exp2(-inf) = 0) in a small wave-uniform branch that only
    // touches the score registers; ONE copy of the arithmetic follows (two copies — masked / unmasked — cost the causal
    // kernel five spilled registers, and every scratch reload in the loop drained the LDS-DMA prefetch with its vmcnt(0)).
    if ((kv0 + 64 > S) || (CAUSAL && kv0 + 63 > q0 + w * 32)) {
      int lim = S - 1 - kv0 - 4 * hi;                 // one per-lane limit against compile-time constants (see the forward kernel)
      if (CAUSAL) lim = min(lim, qi - kv0 - 4 * hi);
#pragma unroll
      for (int kb = 0; kb < 2; ++kb)
#pragma unroll
        for (int r = 0; r < 16; ++r)
          if (kb * 32 + (r & 3) + 8 * (r >> 2) > lim) s[kb][r] = -INFINITY;
    }
#pragma unroll
    for (int kb = 0; kb < 2; ++kb)
#pragma unroll
      for (int r = 0; r < 16; ++r)
        s[kb][r] = fast_exp2(__builtin_fmaf(s[kb][r], scale_log2, -lse2)) * (dp[kb][r] - dlt);
#pragma unroll
    for (int kb = 0; kb < 2; ++kb)
#pragma unroll
      for (int sp = 0; sp < 2; ++sp) {
        float pv[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) pv[j] = s[kb][8 * sp + j];
        const bf16x8 dsf = pack8(pv);
        // keys kv-local (kb*32 + 16*sp) + {4hi + 0..3, 8 + 4hi + 0..3}: the k-slot order of the dS registers
        const unsigned kbytes = sK + (unsigned)(kb * 32 + 16 * sp) * (DH * 2);
        s16x4_t lo[DH / 32], hi4[DH / 32];
#pragma unroll
        for (int db = 0; db < DH / 32; ++db)
          asm volatile("ds_read_b64_tr_b16 %0, %2\n\tds_read_b64_tr_b16 %1, %3"
                       : "=&v"(lo[db]), "=&v"(hi4[db]) : "v"(kt_off[db][0] + kbytes), "v"(kt_off[db][1] + kbytes) : "memory");
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
        for (int db = 0; db < DH / 32; ++db) {
          s16x8_t kv = {lo[db][0], lo[db][1], lo[db][2], lo[db][3], hi4[db][0], hi4[db][1], hi4[db][2], hi4[db][3]};
          asm volatile("" : "+v"(kv));       // the MFMA below must stay behind the lgkmcnt wait
          dq[db] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, kv), dsf, dq[db], 0, 0, 0);
        }
      }
  }
  RopeRow<DH> rr;
  if (rope_cos) rr.load(rope_cos, rope_sin, rope_pos[tok0 + qc], hi);     // (in flight across the barrier)
  __syncthreads();                         // everyone is done with the last K / V stage: the LDS is free for the output blocks
  {
    const int qw0 = q0 + w * 32;
    float mul = scale;
    if (rope_cos) {                          // dQ leaves already rotated back (see rope_inverse_block)
      rope_inverse_block<DH>(dq, scale, rr);
      mul = 1.f;
    }
    store_block_rows<DH, false>(smem + w * (32 * DH * 2), dq, mul, dqkv + (tok0 + qw0) * (long)ld + h * DH, (long)ld, S - qw0, l);
  }
}

// =================================================================================================
// backward, dK / dV  (one key per lane)
// =================================================================================================
// (development builds only, `make DEV=1`; hvlm_configure("attn_dkv_fused", 0) selects it there)
#ifdef HVLM_DEV
// The TWO-PASS form (the A/B alternative of the role-split single pass further down).  MODE 1: dV only
// (S -> P -> dV^T += dO^T P), MODE 2: dK only (S, dP -> dS -> dK^T += Q^T dS): half the accumulators each, so two waves per
// SIMD fit at dh = 128 (a single pass with both accumulators needs 412 registers = one wave per SIMD and ran at 608 us per layer
// against 501 for the two passes); S and the exponentials are computed in both passes.
template <int DH, bool CAUSAL, int MODE>
__global__ __launch_bounds__(256, 2) void attn_bwd_dkv_kernel(
    const bf16_t* __restrict__ qkv, const bf16_t* __restrict__ qt_g /*Q^T*/,
    const bf16_t* __restrict__ dout, const bf16_t* __restrict__ dot_g /*dO^T*/,
    const float* __restrict__ lse, const float* __restrict__ delta, bf16_t* __restrict__ dqkv, int S_max,
    int H, int ld, int S_pad, float scale, const int32_t* __restrict__ cu_seqlens) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr unsigned KT = 64 * DH * 2;
  const int tid = threadIdx.x, l = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l31 = l & 31, hi = l >> 5;
  int kblk, h, b;
  xcd_block(kblk, h, b);
  const long tok0 = cu_seqlens ? (long)cu_seqlens[b] : (long)b * S_max;       // varlen: see attn_fwd_kernel
  const int S = cu_seqlens ? (cu_seqlens[b + 1] - cu_seqlens[b]) : S_max;
  const int k0 = kblk * 128;
  if (k0 >= S) return;
  const int ki = k0 + w * 32 + l31;
  const int kc = min(ki, S - 1);
  const float scale_log2 = scale * LOG2E;

  constexpr bool DO_DV = MODE != 2, DO_DK = MODE != 1;
  bf16x8 kf[DH / 16], vf[DO_DK ? DH / 16 : 1];
  {
    const bf16_t* kp = qkv + (tok0 + kc) * ld + (long)H * DH + h * DH + hi * 8;
    const bf16_t* vp = qkv + (tok0 + kc) * ld + 2L * H * DH + h * DH + hi * 8;
#pragma unroll
    for (int ks = 0; ks < DH / 16; ++ks) {
      kf[ks] = *reinterpret_cast<const bf16x8*>(kp + ks * 16);
      if constexpr (DO_DK) vf[ks] = *reinterpret_cast<const bf16x8*>(vp + ks * 16);
    }
  }
  f32x16 dk[DO_DK ? DH / 32 : 1], dv[DO_DV ? DH / 32 : 1];
#pragma unroll
  for (int db = 0; db < DH / 32; ++db)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      if constexpr (DO_DK) dk[db][r] = 0.f;
      if constexpr (DO_DV) dv[db][r] = 0.f;
    }

  const int nqt = (S + 63) / 64;
  const int qt_begin = CAUSAL ? (k0 / 64) : 0;
  const int mk_base = CAUSAL ? ki : 0;                                             // see the mask in the loop
  const unsigned mk_lim = ki < S ? (unsigned)(S - mk_base) : 0u;
  const bf16_t* qbase = qkv + tok0 * ld + h * DH;
  const bf16_t* dobase = dout + tok0 * (long)(H * DH) + h * DH;
  const float* lse_row = lse + ((long)b * H + h) * S_max;
  const float* dlt_row = delta + ((long)b * H + h) * S_max;

  // two LDS stages (see the dQ kernel): tiles + the 64 lse / delta values of step qt+1 are fetched during step qt.  A stage
  // holds Q and dO only; the Q^T / dO^T fragments of dK^T = Q^T dS and dV^T = dO^T P come from those tiles through
  // `ds_read_b64_tr_b16` (inline asm, see the dQ kernel): no Q^T / dO^T tiles, no pre-passes, two workgroups per CU.
  constexpr unsigned STAGE = 2 * KT + 512;
  float lse_n = 0.f, dlt_n = 0.f;               // wave 0: next step's lse*log2e / delta of query qq0 + tid
  auto stage = [&](int buf, int qt) {
    const unsigned o = (unsigned)buf * STAGE;
    const int qq0 = qt * 64;
    load_tile<64, DH>(smem, o, qbase, ld, qq0, S, w, l);
    load_tile<64, DH>(smem, o + KT, dobase, (long)H * DH, qq0, S, w, l);
    if (tid < 64) {
      const int q = min(qq0 + tid, S - 1);
      lse_n = lse_row[q] * LOG2E;
      dlt_n = dlt_row[q];
    }
  };
  const int p16 = l & 15, g16 = l31 >> 4;
  const unsigned lds0 = (unsigned)(unsigned long)(lds_ptr_t)smem;
  unsigned tr_off[DH / 32][2];
#pragma unroll
  for (int db = 0; db < DH / 32; ++db)
#pragma unroll
    for (int hf = 0; hf < 2; ++hf)
      tr_off[db][hf] = lds0 + tile_addr<DH>(4 * hi + (p16 >> 2) + 8 * hf, db * 4 + g16 * 2 + ((p16 & 3) >> 1)) + (unsigned)((p16 & 1) * 8);
  if (qt_begin < nqt) stage(0, qt_begin);
  for (int qt = qt_begin; qt < nqt; ++qt) {
    const int qq0 = qt * 64;
    const unsigned sQ = (unsigned)((qt - qt_begin) & 1) * STAGE, sDO = sQ + KT;
    float* s_lse = reinterpret_cast<float*>(smem + sQ + 2 * KT);   // [64] lse*log2e
    float* s_dlt = s_lse + 64;                                     // [64]
    if (tid < 64) { s_lse[tid] = lse_n; s_dlt[tid] = dlt_n; }       // this stage's slot: nobody reads it before the barrier
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();   // step qt landed for everyone; everyone is done with step qt-1
    if (qt + 1 < nqt) stage(((qt + 1 - qt_begin) & 1), qt + 1);
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
        if constexpr (DO_DK) {
          const bf16x8 dofr = lds_frag128(smem, sDO + a);
          dp = __builtin_amdgcn_mfma_f32_32x32x16_bf16(dofr, vf[ks], dp, 0, 0, 0);
        }
      }
      // dS = P o (dP - delta); the softmax scale is applied once to dK at the end.
      // The Q^T / dO^T fragments (4 transpose reads per d block) are software-pipelined one set ahead of the two MFMAs that
      // consume them, and the first set of a query half is in flight while its P / dS values are computed: with one wave
      // per SIMD (dh = 128) nothing else hides the LDS latency.  lse / delta of the 16 query rows are read BEFORE the first
      // set is issued (a tracked LDS read behind untracked ones would make the compiler drain them).  A second, mask-free
      // variant of this section for interior tiles (as in the dQ kernel) was tried twice — as a branch and as a second copy
      // of the tile body: both spill (344 / 560 B of scratch: the accumulators meet at the join) and run 1.8x slower.
      float lse_r[16], dlt_r[16];
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int ql = qb * 32 + crow(r, hi);
        lse_r[r] = s_lse[ql];
        dlt_r[r] = s_dlt[ql];
      }
      asm volatile("" : "+v"(lse_r[0]), "+v"(lse_r[1]), "+v"(lse_r[2]), "+v"(lse_r[3]), "+v"(lse_r[4]), "+v"(lse_r[5]), "+v"(lse_r[6]), "+v"(lse_r[7]), "+v"(lse_r[8]), "+v"(lse_r[9]), "+v"(lse_r[10]), "+v"(lse_r[11]), "+v"(lse_r[12]), "+v"(lse_r[13]), "+v"(lse_r[14]), "+v"(lse_r[15]));
      asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(dlt_r[0]), "+v"(dlt_r[1]), "+v"(dlt_r[2]), "+v"(dlt_r[3]), "+v"(dlt_r[4]), "+v"(dlt_r[5]), "+v"(dlt_r[6]), "+v"(dlt_r[7]), "+v"(dlt_r[8]), "+v"(dlt_r[9]), "+v"(dlt_r[10]), "+v"(dlt_r[11]), "+v"(dlt_r[12]), "+v"(dlt_r[13]), "+v"(dlt_r[14]), "+v"(dlt_r[15]));
      s16x4_t d0[2], d1[2], q0r[2], q1r[2];
#define HVLM_DKV_ISSUE(SET, SP, DB)                                                                                    \
      do {                                                                                                              \
        const unsigned qbytes_ = (unsigned)(qb * 32 + 16 * (SP)) * (DH * 2);                                           \
        if constexpr (MODE == 1)                                                                                        \
          asm volatile("ds_read_b64_tr_b16 %0, %2\n\tds_read_b64_tr_b16 %1, %3"                                        \
                       : "=&v"(d0[SET]), "=&v"(d1[SET])                                                                 \
                       : "v"(tr_off[DB][0] + sDO + qbytes_), "v"(tr_off[DB][1] + sDO + qbytes_) : "memory");           \
        else                                                                                                            \
          asm volatile("ds_read_b64_tr_b16 %0, %2\n\tds_read_b64_tr_b16 %1, %3"                                        \
                       : "=&v"(q0r[SET]), "=&v"(q1r[SET])                                                               \
                       : "v"(tr_off[DB][0] + sQ + qbytes_), "v"(tr_off[DB][1] + sQ + qbytes_) : "memory");             \
      } while (0)
      HVLM_DKV_ISSUE(0, 0, 0);
      bf16x8 pf, dsf;
#pragma unroll
      for (int idx = 0; idx < 2 * (DH / 32); ++idx) {
        const int sp = idx / (DH / 32), db = idx % (DH / 32);
        if (db == 0) {
          float pvals[8], dsvals[8];
#pragma unroll
          for (int j = 0; j < 8; ++j) {
            const int r = 8 * sp + j;
            const int ql = qb * 32 + crow(r, hi);
            float pj = fast_exp2(__builtin_fmaf(s[r], scale_log2, -lse_r[r]));
            // valid iff (causal: ki <= q) && q < S && ki < S  ==  (unsigned)(q - mk_base) < mk_lim : one add + one compare
            if ((unsigned)(qq0 + ql - mk_base) >= mk_lim) pj = 0.f;
            pvals[j] = pj;
            dsvals[j] = pj * (dp[r] - dlt_r[r]);
          }
          // queries (qb*32 + 16*sp) + {4hi + 0..3, 8 + 4hi + 0..3}: the k-slot order of the P / dS registers
          pf = pack8(pvals);
          dsf = pack8(dsvals);
        }
        const int cur = idx & 1;
        if (idx + 1 < 2 * (DH / 32)) {
          if (cur == 0) HVLM_DKV_ISSUE(1, (idx + 1) / (DH / 32), (idx + 1) % (DH / 32));
          else HVLM_DKV_ISSUE(0, (idx + 1) / (DH / 32), (idx + 1) % (DH / 32));
          if constexpr (MODE == 1) asm volatile("s_waitcnt lgkmcnt(2)" : "+v"(d0[cur]), "+v"(d1[cur]));
          else asm volatile("s_waitcnt lgkmcnt(2)" : "+v"(q0r[cur]), "+v"(q1r[cur]));
        } else {
          if constexpr (MODE == 1) asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(d0[cur]), "+v"(d1[cur]));
          else asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(q0r[cur]), "+v"(q1r[cur]));
        }
        if constexpr (DO_DV) {
          const s16x8_t dotv = {d0[cur][0], d0[cur][1], d0[cur][2], d0[cur][3], d1[cur][0], d1[cur][1], d1[cur][2], d1[cur][3]};
          dv[db] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, dotv), pf, dv[db], 0, 0, 0);
        }
        if constexpr (DO_DK) {
          const s16x8_t qtv = {q0r[cur][0], q0r[cur][1], q0r[cur][2], q0r[cur][3], q1r[cur][0], q1r[cur][1], q1r[cur][2], q1r[cur][3]};
          dk[db] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, qtv), dsf, dk[db], 0, 0, 0);
        }
      }
#undef HVLM_DKV_ISSUE
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
        if constexpr (DO_DK) {
          u32x2 a = {pack_bf2(dk[db][4 * g] * scale, dk[db][4 * g + 1] * scale),
                     pack_bf2(dk[db][4 * g + 2] * scale, dk[db][4 * g + 3] * scale)};
          *reinterpret_cast<u32x2*>(kp + d) = a;
        }
        if constexpr (DO_DV) {
          u32x2 c = {pack_bf2(dv[db][4 * g], dv[db][4 * g + 1]), pack_bf2(dv[db][4 * g + 2], dv[db][4 * g + 3])};
          *reinterpret_cast<u32x2*>(vp + d) = c;
        }
      }
  }
}

#endif  // HVLM_DEV

// =================================================================================================
// backward, dK AND dV in one pass by ROLE-SPLIT waves (8 waves, 128 keys per workgroup, one workgroup per CU)
//   waves 0-3 ("A", keys 32w..): S = Q K^T -> P = exp2(S*scale - lse) -> P to LDS (fp32, lane-linear) -> dV^T += dO^T P
//   waves 4-7 ("B", same keys):  dP = dO V^T -> P from LDS -> dS = P o (dP - delta) -> dK^T += Q^T dS
// Against the two-pass form (attn_bwd_dkv_kernel MODE 1 + MODE 2): S and the exponentials once instead of twice (4 matmul units
// instead of 5), the Q / dO tiles fetched once for eight waves, one launch — and with it one set of per-workgroup fixed costs —
// less.  B runs ONE 32-query half-step ("slot") behind A, so a slot is: A works on (step, qb) | B on the previous (step, qb);
// one barrier per slot hands P[slot & 1] over.  Q / dO stages: three buffers (A reads stage s, B stage s or s-1, the DMA of
// stage s+1 / s+2 is in flight), each fetched 3 slots before A needs it; the counted vmcnt leaves the youngest DMA pending.
// =================================================================================================
template <int DH, bool CAUSAL>
__device__ __forceinline__ void attn_bwd_dkv_fused_body(const bf16_t* __restrict__ qkv, const bf16_t* __restrict__ dout,
                                                        const float* __restrict__ lse, const float* __restrict__ delta,
                                                        bf16_t* __restrict__ dqkv, int S_max, int H, int ld, float scale,
                                                        const int32_t* __restrict__ cu_seqlens, int B, int* __restrict__ ctr,
                                                        const bf16_t* __restrict__ rope_cos, const bf16_t* __restrict__ rope_sin,
                                                        const int32_t* __restrict__ rope_pos) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr unsigned KT = 64 * DH * 2;
  constexpr unsigned STAGE = 2 * KT + 512;            // Q tile, dO tile, lse*log2e [64], delta [64]
  constexpr unsigned P_OFF = 3 * STAGE;               // P hand-off: [2 slots][4 key waves][4 register quads][64 lanes] x 16 B
  constexpr unsigned P_SLOT = 4 * 4 * 1024;
  constexpr unsigned NEXT_OFF = P_OFF + 2 * P_SLOT;   // one int: the entry this workgroup takes next
  constexpr unsigned EPI_OFF = 2 * STAGE;             // output staging (8 waves x 32 rows): clear of stage buffers 0 and 1
  const int tid = threadIdx.x, l = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l31 = l & 31, hi = l >> 5;
  const bool roleB = w >= 4;                          // wave-uniform
  const int wk = w & 3;
  const float scale_log2 = scale * LOG2E;
  const int p16 = l & 15, g16 = l31 >> 4;
  const unsigned lds0 = (unsigned)(unsigned long)(lds_ptr_t)smem;
  unsigned tr_off[DH / 32][2];
#pragma unroll
  for (int db = 0; db < DH / 32; ++db)
#pragma unroll
    for (int hf = 0; hf < 2; ++hf)
      tr_off[db][hf] = lds0 + tile_addr<DH>(4 * hi + (p16 >> 2) + 8 * hf, db * 4 + g16 * 2 + ((p16 & 3) >> 1)) + (unsigned)((p16 & 1) * 8);

  // ---- PERSISTENT workgroup (one per CU: the kernel needs 130 KiB of LDS, so no second workgroup hides the prologue and the
  // epilogue of this one): the (sample, head, key block) entries of this XCD's contiguous chunk (see xcd_block) are drawn
  // from a per-XCD counter in order — the key blocks of one (sample, head) run on one XCD at about the same time — and the
  // first two stages and the operand rows of the NEXT entry are requested before the output of the current one is stored.
  const int gx = (S_max + 127) / 128;
  const int total = gx * H * B;
  const int xcd = (int)blockIdx.x & 7;
  const int cq = total >> 3, cr = total & 7;
  const int c_start = xcd * cq + min(xcd, cr), c_len = cq + (xcd < cr ? 1 : 0);
  int* const next_slot = reinterpret_cast<int*>(smem + NEXT_OFF);
  struct Entry { int b, h, k0, S, nsteps, qt_begin; long tok0; };
  auto decode = [&](int idx, Entry& e) -> bool {     // false: nothing to do (past the sample's length: varlen)
    const int v = c_start + idx;
    const int kblk = v % gx, rest = v / gx;
    e.h = rest % H;
    e.b = rest / H;
    e.tok0 = cu_seqlens ? (long)cu_seqlens[e.b] : (long)e.b * S_max;       // varlen: see attn_fwd_kernel
    e.S = cu_seqlens ? (cu_seqlens[e.b + 1] - cu_seqlens[e.b]) : S_max;
    e.k0 = kblk * 128;
    if (e.k0 >= e.S) return false;
    e.qt_begin = CAUSAL ? (e.k0 / 64) : 0;
    e.nsteps = (e.S + 63) / 64 - e.qt_begin;
    return true;
  };
  auto draw = [&]() {                                 // thread 0: next entry of this XCD's chunk -> LDS (read after a barrier)
    if (tid == 0) *next_slot = (int)__hip_atomic_fetch_add(&ctr[xcd], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  };

  Entry cur;
  bf16x8 own[DH / 16];                                // A: the wave's K rows; B: its V rows (one key per lane)
  float lse_n = 0.f, dlt_n = 0.f;               // wave 0: lse*log2e / delta of query (stage's first query + tid), on their way to LDS
  auto load_own = [&](const Entry& e) {
    const int kc = min(e.k0 + wk * 32 + l31, e.S - 1);
    const bf16_t* rp = qkv + (e.tok0 + kc) * ld + (long)(roleB ? 2 : 1) * H * DH + e.h * DH + hi * 8;
#pragma unroll
    for (int ks = 0; ks < DH / 16; ++ks) own[ks] = *reinterpret_cast<const bf16x8*>(rp + ks * 16);
  };
  float lse_0 = 0.f, dlt_0 = 0.f;               // the same for stage 0 of an entry (parked behind the previous entry's epilogue)
  auto stage_issue = [&](const Entry& e, int st, float& ls, float& dl) {    // st = step index relative to qt_begin
    const unsigned o = (unsigned)(st % 3) * STAGE;
    const int qq0 = (e.qt_begin + st) * 64;
    load_tile<64, DH, 8>(smem, o, qkv + e.tok0 * ld + e.h * DH, ld, qq0, e.S, w, l);
    load_tile<64, DH, 8>(smem, o + KT, dout + e.tok0 * (long)(H * DH) + e.h * DH, (long)H * DH, qq0, e.S, w, l);
    if (tid < 64) {
      const int q = min(qq0 + tid, e.S - 1);
      ls = lse[((long)e.b * H + e.h) * S_max + q] * LOG2E;
      dl = delta[((long)e.b * H + e.h) * S_max + q];
    }
  };
  auto park_scalars = [&](int st, float ls, float dl) {   // wave 0: the values fetched with stage `st` go to its LDS slot
    if (tid < 64) {
      float* sl_ = reinterpret_cast<float*>(smem + (unsigned)(st % 3) * STAGE + 2 * KT);
      sl_[tid] = ls;
      sl_[64 + tid] = dl;
    }
  };
  auto entry_prologue = [&](const Entry& e) {   // operand rows + stages 0 and 1 in flight (their scalars: parked at the loop top)
    load_own(e);
    stage_issue(e, 0, lse_0, dlt_0);
    if (e.nsteps > 1) stage_issue(e, 1, lse_n, dlt_n);
  };

  // first entry
  draw();
  __syncthreads();
  int idx = *next_slot;
  bool have = false;
  while (idx < c_len && !(have = decode(idx, cur))) {
    __syncthreads();
    draw();
    __syncthreads();
    idx = *next_slot;
  }
  if (!have) return;
  entry_prologue(cur);

  for (;;) {
    __syncthreads();                                  // everyone has read next_slot
    draw();                                           // the entry after this one
    park_scalars(0, lse_0, dlt_0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();                                  // stages 0 / 1 landed; the output staging of the previous entry is drained
    const int nidx = *next_slot;
    const int k0 = cur.k0, S = cur.S, nsteps = cur.nsteps, qt_begin = cur.qt_begin;
    const int ki = k0 + wk * 32 + l31;
    const int mk_base = CAUSAL ? ki : 0;                                             // valid iff (unsigned)(q - mk_base) < mk_lim
    const unsigned mk_lim = ki < S ? (unsigned)(S - mk_base) : 0u;
    f32x16 acc[DH / 32];                                // A: dV^T, B: dK^T
#pragma unroll
    for (int db = 0; db < DH / 32; ++db)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[db][r] = 0.f;

    // the transposed-operand MFMAs of a slot: acc[db] += T^T(sp, db) x bfrag[sp], T = the dO (A) / Q (B) tile at LDS offset `tbase`
    auto tr_mfmas = [&](unsigned tbase, int qb, const bf16x8 (&bfrag)[2]) {
      s16x4_t t0[2], t1[2];
#define HVLM_FUSED_ISSUE(SET, SP, DB)                                                                                   \
      do {                                                                                                              \
        const unsigned qbytes_ = (unsigned)(qb * 32 + 16 * (SP)) * (DH * 2);                                            \
        asm volatile("ds_read_b64_tr_b16 %0, %2\n\tds_read_b64_tr_b16 %1, %3"                                          \
                     : "=&v"(t0[SET]), "=&v"(t1[SET])                                                                   \
                     : "v"(tr_off[DB][0] + tbase + qbytes_), "v"(tr_off[DB][1] + tbase + qbytes_) : "memory");          \
      } while (0)
      HVLM_FUSED_ISSUE(0, 0, 0);
#pragma unroll
      for (int ix = 0; ix < 2 * (DH / 32); ++ix) {
        const int sp = ix / (DH / 32), db = ix % (DH / 32);
        const int cu_ = ix & 1;
        if (ix + 1 < 2 * (DH / 32)) {
          if (cu_ == 0) HVLM_FUSED_ISSUE(1, (ix + 1) / (DH / 32), (ix + 1) % (DH / 32));
          else HVLM_FUSED_ISSUE(0, (ix + 1) / (DH / 32), (ix + 1) % (DH / 32));
          asm volatile("s_waitcnt lgkmcnt(2)" : "+v"(t0[cu_]), "+v"(t1[cu_]));
        } else {
          asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(t0[cu_]), "+v"(t1[cu_]));
        }
        const s16x8_t tv = {t0[cu_][0], t0[cu_][1], t0[cu_][2], t0[cu_][3], t1[cu_][0], t1[cu_][1], t1[cu_][2], t1[cu_][3]};
        acc[db] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, tv), bfrag[sp], acc[db], 0, 0, 0);
      }
#undef HVLM_FUSED_ISSUE
    };

    const int nslots = 2 * nsteps + 1;
    for (int sl = 0; sl < nslots; ++sl) {
      // DMA of stage (sl + 3) / 2 in the odd slots: its buffer was last read by B one slot ago
      bool issued = false;
      if ((sl & 1) && (sl + 3) / 2 < nsteps) { stage_issue(cur, (sl + 3) / 2, lse_n, dlt_n); issued = true; }
      const int mysl = roleB ? sl - 1 : sl;             // the (step, qb) this wave works on in this slot
      if (mysl >= 0 && mysl < 2 * nsteps) {
        const int st = mysl >> 1, qb = mysl & 1;
        const int qq0 = (qt_begin + st) * 64;
        const unsigned sQ = (unsigned)(st % 3) * STAGE, sDO = sQ + KT;
        // step fully masked for this wave's keys (all its queries precede them): both roles skip it, nothing is handed over
        if (!(CAUSAL && qq0 + 63 < k0 + wk * 32)) {
          const float* s_sc = reinterpret_cast<const float*>(smem + sQ + 2 * KT) + (roleB ? 64 : 0);   // lse (A) / delta (B)
          char* pbuf = smem + P_OFF + (unsigned)(mysl & 1) * P_SLOT + (unsigned)(wk * 4) * 1024u + (unsigned)l * 16u;
          f32x16 x;                                       // A: S, then P;  B: dP, then dS
#pragma unroll
          for (int r = 0; r < 16; ++r) x[r] = 0.f;
          const unsigned opnd = roleB ? sDO : sQ;
          {
            // all operand fragments of the slot requested before the first MFMA of the (dependent) chain
            bf16x8 fr[DH / 16];
#pragma unroll
            for (int ks = 0; ks < DH / 16; ++ks) fr[ks] = lds_frag128(smem, opnd + tile_addr<DH>(qb * 32 + l31, ks * 2 + hi));
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int ks = 0; ks < DH / 16; ++ks) x = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fr[ks], own[ks], x, 0, 0, 0);
          }
          float sc_r[16];                                  // registers 4g..4g+3 = queries 8g + 4hi + 0..3: one 16-byte read each
#pragma unroll
          for (int g4 = 0; g4 < 4; ++g4) {
            const f32x4 v4 = *reinterpret_cast<const f32x4*>(s_sc + qb * 32 + 8 * g4 + 4 * hi);
            sc_r[4 * g4] = v4[0]; sc_r[4 * g4 + 1] = v4[1]; sc_r[4 * g4 + 2] = v4[2]; sc_r[4 * g4 + 3] = v4[3];
          }
          bf16x8 bfrag[2];
          if (!roleB) {
            // mask (valid iff causal: key <= query, query < S, key < S) only in boundary slots — the causal diagonal, the
            // sequence tail — as a pre-mask of the scores (exp2(-inf) = 0): a small wave-uniform branch that touches x only
            const int q_lo = qq0 + qb * 32;
            if ((CAUSAL && q_lo < k0 + wk * 32 + 32) || q_lo + 32 > S || k0 + wk * 32 + 32 > S) {
#pragma unroll
              for (int r = 0; r < 16; ++r)
                if ((unsigned)(q_lo + crow(r, hi) - mk_base) >= mk_lim) x[r] = -INFINITY;
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) x[r] = fast_exp2(__builtin_fmaf(x[r], scale_log2, -sc_r[r]));
#pragma unroll
            for (int g4 = 0; g4 < 4; ++g4) {
              const f32x4 pv4 = {x[4 * g4], x[4 * g4 + 1], x[4 * g4 + 2], x[4 * g4 + 3]};
              *reinterpret_cast<f32x4*>(pbuf + g4 * 1024) = pv4;
            }
          } else {
            float pr[16];
#pragma unroll
            for (int g4 = 0; g4 < 4; ++g4) {
              const f32x4 pv4 = *reinterpret_cast<const f32x4*>(pbuf + g4 * 1024);
              pr[4 * g4] = pv4[0]; pr[4 * g4 + 1] = pv4[1]; pr[4 * g4 + 2] = pv4[2]; pr[4 * g4 + 3] = pv4[3];
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) x[r] = pr[r] * (x[r] - sc_r[r]);
          }
#pragma unroll
          for (int sp = 0; sp < 2; ++sp) {
            float v8[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) v8[j] = x[8 * sp + j];
            bfrag[sp] = pack8(v8);
          }
          // every tracked LDS access of the slot is retired before the untracked transpose reads start
          asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(bfrag[0]), "+v"(bfrag[1]) :: "memory");
          tr_mfmas(roleB ? sQ : sDO, qb, bfrag);
        }
      }
      // end of slot: the stage A starts next (stage (sl + 1) / 2 at an even slot sl + 1) was fetched 3 slots ago and must have
      // landed; the DMA issued in THIS slot (4 pieces per wave at dh = 128, + the two scalar loads of wave 0) stays in flight
      if (!(sl & 1)) park_scalars(sl / 2 + 1, lse_n, dlt_n);   // (tracked loads: the compiler waits for them here)
      if (sl & 1) {
        if (issued) {
          if constexpr (DH == 128) {
            if (w == 0) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
          } else {
            if (w == 0) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
          }
        } else {
          asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
      }
      __syncthreads();
    }
    // the next entry: its operand rows and first two stages travel while this entry's output is stored
    Entry nxt;
    bool more = false;
    int ni = nidx;
    while (ni < c_len && !(more = decode(ni, nxt))) {   // (entries past a short sample's length: draw again — rare, varlen only)
      __syncthreads();
      draw();
      __syncthreads();
      ni = *next_slot;
    }
    const Entry done = cur;
    if (more) { cur = nxt; entry_prologue(cur); }
    // epilogue through LDS (16-byte row-contiguous stores); the staging area does not touch stage buffers 0 / 1
    {
      const int kw0 = done.k0 + wk * 32;
      bf16_t* g0 = dqkv + (done.tok0 + kw0) * (long)ld + (long)(roleB ? 1 : 2) * H * DH + done.h * DH;
      float mul = roleB ? scale : 1.f;
      if (roleB && rope_cos) {               // dK leaves already rotated back (see rope_inverse_block); dV has no rotation
        RopeRow<DH> rr;                      // (requested here: held across the entry switch, the 32 registers spill)
        rr.load(rope_cos, rope_sin, rope_pos[done.tok0 + min(kw0 + l31, done.S - 1)], hi);
        rope_inverse_block<DH>(acc, scale, rr);
        mul = 1.f;
      }
      store_block_rows<DH, false>(smem + EPI_OFF + w * (32 * DH * 2), acc, mul, g0, (long)ld, done.S - kw0, l);
    }
    if (!more) break;
  }
}
template <int DH, bool CAUSAL>
__global__ __launch_bounds__(512, 2) void attn_bwd_dkv_fused_kernel(const bf16_t* __restrict__ qkv, const bf16_t* __restrict__ dout,
                                                                    const float* __restrict__ lse, const float* __restrict__ delta,
                                                                    bf16_t* __restrict__ dqkv, int S_max, int H, int ld, float scale,
                                                                    const int32_t* __restrict__ cu_seqlens, int B, int* __restrict__ ctr,
                                                                    const bf16_t* __restrict__ rope_cos, const bf16_t* __restrict__ rope_sin,
                                                                    const int32_t* __restrict__ rope_pos) {
  attn_bwd_dkv_fused_body<DH, CAUSAL>(qkv, dout, lse, delta, dqkv, S_max, H, ld, scale, cu_seqlens, B, ctr, rope_cos, rope_sin, rope_pos);
}

static inline int spad(int S) { return (S + 63) / 64 * 64; }

template <typename K>
static int set_lds(K kernel, int bytes, const char* what) {
  hipError_t e = hipFuncSetAttribute((const void*)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
  if (e != hipSuccess) { set_error("%s: hipFuncSetAttribute: %s", what, hipGetErrorString(e)); return HVLM_ERR_LAUNCH; }
  return HVLM_OK;
}

template <int DH, bool CAUSAL, bool F16 = false>
static int fwd_impl(const bf16_t* qkv, bf16_t* out, float* lse, bf16_t* ws, int B, int S, int H, int ld,
                    float scale, hipStream_t st, const int32_t* cu = nullptr) {
  int rc;
  if (!cu) if constexpr (!CAUSAL && DH == 64) {
    // short sequences: K/V resident in LDS, one workgroup per (batch, head), no V^T pre-pass
    const int nkb = (S + 31) / 32;
    if (nkb <= 9 && !config().attn_no_resident) {
      const int lds = 2 * 2 * nkb * 32 * DH * 2 + 1024;   // two (K + V) buffers + the one-query wave's probability strip
      rc = set_lds(attn_fwd_resident_kernel<DH, F16>, lds, "attn_fwd_resident");
      if (rc) return rc;
      static int cus = 0;
      if (!cus) {
        int dev = 0; hipDeviceProp_t prop;
        cus = (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess) ? prop.multiProcessorCount : 256;
      }
      const int nitems = B * H;
      if constexpr (F16) {
        // CLIP ViT-L/14 at 224 px, the tower's own dtype: the hand-scheduled one-wave-per-SIMD kernel (csrc/asm/attn_clip_d64.py)
        // (calls beyond its 32-bit byte offsets, or a module that did not load, fall through to the C++ kernel)
        if (S == 257 && !lse && config().attn_clip_asm && attn_fwd_clip_s257_asm_ok(B, H, ld))
          return attn_fwd_clip_s257_asm(qkv, out, B, H, ld, scale * LOG2E, st);
      }
      if (S == 257) {                          // CLIP ViT-L/14 at 224 px: the unrolled, LDS-latency-pipelined schedule
        rc = set_lds(attn_fwd_resident_kernel<DH, F16, 257>, lds, "attn_fwd_resident");
        if (rc) return rc;
        hipLaunchKernelGGL((attn_fwd_resident_kernel<DH, F16, 257>), dim3(nitems < cus ? nitems : cus), dim3(64 * (nkb + 1)), lds, st, qkv, out,
                           lse, S, H, ld, nkb, nitems, scale * LOG2E);
        return check_launch("attn_fwd_resident");
      }
      hipLaunchKernelGGL((attn_fwd_resident_kernel<DH, F16>), dim3(nitems < cus ? nitems : cus), dim3(64 * (nkb + 1)), lds, st, qkv, out, lse,
                         S, H, ld, nkb, nitems, scale * LOG2E);
      return check_launch("attn_fwd_resident");
    }
  }
  if constexpr (DH == 128 && CAUSAL && !F16) {
    // the decoder's forward: hand-scheduled one-wave-per-SIMD kernel (csrc/asm/attn_fwd_d128.py, attention_asm.hip)
    // Its eight work-queue counters live in `ws` (hvlm.h: >= hvlm_attn_fwd_ws_bytes() = 1024 bytes of which 512 are written, 64-byte aligned — the size is
    // the caller's contract, only the alignment can be checked here).  A null / misaligned workspace, a module that did not
    // load, or >= 2^24 work items run on the C++ kernel below, which needs none of it.
    if (config().attn_fwd_asm && attn_fwd_d128_causal_asm_ok(ws, B, S, H))
      return attn_fwd_d128_causal_asm(qkv, out, lse, ws, B, S, H, ld, scale * LOG2E, cu, st);
  }
  constexpr int LDS = 2 * 2 * 64 * DH * 2;   // (no V^T pre-pass: `ws` is unused, kept in the signature for ABI stability)
  // (96-row workgroups — whichever of 128 / 96 wastes fewer query rows, S = 257 -> 96 — measured slower in situ, 649 vs 593 us
  // per CLIP layer, and were removed)
#ifdef HVLM_DEV
  if constexpr (DH == 128 && !F16) {
    if (config().attn_fwd_pipe) {            // the intra-wave software pipeline (attn_fwd_pipe_body), development builds
      rc = set_lds(attn_fwd_pipe_kernel<DH, CAUSAL, 4>, LDS, "attn_fwd_pipe");
      if (rc) return rc;
      dim3 grid((S + 127) / 128, H, B);
      hipLaunchKernelGGL((attn_fwd_pipe_kernel<DH, CAUSAL, 4>), grid, dim3(256), LDS, st, qkv, out, lse, S, H, ld, scale * LOG2E, cu);
      return check_launch("attn_fwd_pipe");
    }
  }
#endif
  {
    rc = set_lds(attn_fwd_kernel<DH, CAUSAL, 4, F16>, LDS, "attn_fwd");
    if (rc) return rc;
    dim3 grid((S + 127) / 128, H, B);
    hipLaunchKernelGGL((attn_fwd_kernel<DH, CAUSAL, 4, F16>), grid, dim3(256), LDS, st, qkv, (const bf16_t*)ws, out,
                       lse, S, H, ld, spad(S), scale * LOG2E, cu);
  }
  return check_launch("attn_fwd");
}

template <int DH, bool CAUSAL>
static int bwd_impl(const bf16_t* qkv, const bf16_t* out, const bf16_t* dout, const float* lse,
                    bf16_t* dqkv, char* ws, int B, int S, int H, int ld, float scale, hipStream_t st,
                    const int32_t* cu = nullptr, long total_tokens = 0, const bf16_t* rope_cos = nullptr,
                    const bf16_t* rope_sin = nullptr, const int32_t* rope_pos = nullptr) {
  bf16_t* qT = nullptr; bf16_t* kT = nullptr; bf16_t* doT = nullptr;     // (unused kernel arguments, kept for signature stability)
  float* delta = reinterpret_cast<float*>(ws);
  const int fused = config().attn_dkv_fused;
  int rc;   // (no Q^T / K^T / dO^T pre-passes any more: the kernels gather those fragments with LDS transpose reads)
  const bool delta_sep = config().attn_delta_separate != 0;     // 1: delta by its own pass (A/B); default: inside the dQ kernel
  if (delta_sep) {
    const long tokens = cu ? total_tokens : (long)B * S;
    const long total = tokens * H * (DH / 8);
    hipLaunchKernelGGL((attn_delta_kernel<DH>), dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, out,
                       dout, delta, tokens, S, H, cu, B);
    if ((rc = check_launch("attn_delta"))) return rc;
  }
  (void)kT;
  {
    constexpr int LDS = 2 * 2 * 64 * DH * 2;            // two stages of (K, V)
    if ((rc = set_lds(attn_bwd_dq_kernel<DH, CAUSAL>, LDS, "attn_bwd_dq"))) return rc;
    dim3 grid((S + 127) / 128, H, B);
    hipLaunchKernelGGL((attn_bwd_dq_kernel<DH, CAUSAL>), grid, dim3(256), LDS, st, qkv, delta_sep ? (const bf16_t*)nullptr : out,
                       dout, lse, delta, dqkv, S, H, ld, spad(S), scale, cu,
                       fused ? rope_cos : (const bf16_t*)nullptr, rope_sin, rope_pos);
    if ((rc = check_launch("attn_bwd_dq"))) return rc;
  }
  {
    constexpr int LDS = 2 * (2 * 64 * DH * 2 + 512);    // two stages of (Q, dO, lse, delta)
    dim3 grid((S + 127) / 128, H, B);
    if (fused) {
      // dK and dV in one pass by role-split waves (see attn_bwd_dkv_fused_body)
      constexpr int LDSF = 3 * (2 * 64 * DH * 2 + 512) + 2 * 4 * 4 * 1024 + 16;
      if ((rc = set_lds(attn_bwd_dkv_fused_kernel<DH, CAUSAL>, LDSF, "attn_bwd_dkv_fused"))) return rc;
      static int cusf = 0;
      if (!cusf) {
        int dev = 0; hipDeviceProp_t prop;
        cusf = (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess) ? prop.multiProcessorCount : 256;
        cusf = cusf / 8 * 8;
      }
      // per-XCD entry counters: the 256 spare bytes behind delta[B, H, S] in the workspace (hvlm_attn_bwd_ws_bytes)
      int* ctr = reinterpret_cast<int*>(ws + (((size_t)B * H * S * 4 + 15) & ~(size_t)15));
      if (hipMemsetAsync(ctr, 0, 32, st) != hipSuccess) { set_error("attn_bwd: hipMemsetAsync failed"); return HVLM_ERR_LAUNCH; }
      const long total = (long)((S + 127) / 128) * H * B;
      hipLaunchKernelGGL((attn_bwd_dkv_fused_kernel<DH, CAUSAL>), dim3((unsigned)(total < cusf ? total : cusf)), dim3(512), LDSF, st,
                         qkv, dout, lse, (const float*)delta, dqkv, S, H, ld, scale, cu, B, ctr, rope_cos, rope_sin, rope_pos);
      if ((rc = check_launch("attn_bwd_dkv_fused"))) return rc;
    } else {
#ifdef HVLM_DEV
      // the two-pass form: dV, then dK (two waves per SIMD each)
      if ((rc = set_lds(attn_bwd_dkv_kernel<DH, CAUSAL, 1>, LDS, "attn_bwd_dv"))) return rc;
      hipLaunchKernelGGL((attn_bwd_dkv_kernel<DH, CAUSAL, 1>), grid, dim3(256), LDS, st, qkv, (const bf16_t*)qT,
                         dout, (const bf16_t*)doT, lse, (const float*)delta, dqkv, S, H, ld, spad(S), scale, cu);
      if ((rc = check_launch("attn_bwd_dv"))) return rc;
      if ((rc = set_lds(attn_bwd_dkv_kernel<DH, CAUSAL, 2>, LDS, "attn_bwd_dk"))) return rc;
      hipLaunchKernelGGL((attn_bwd_dkv_kernel<DH, CAUSAL, 2>), grid, dim3(256), LDS, st, qkv, (const bf16_t*)qT,
                         dout, (const bf16_t*)doT, lse, (const float*)delta, dqkv, S, H, ld, spad(S), scale, cu);
      if ((rc = check_launch("attn_bwd_dk"))) return rc;
      // the two-pass kernels do not rotate: the separate (inverse) rope pass over dq and dk, as before
      if (rope_cos) {
        const long tokens = cu ? total_tokens : (long)B * S;
        if ((rc = hvlm_rope_qk(dqkv, rope_cos, rope_sin, rope_pos, (int)tokens, H, DH, ld, 1, st))) return rc;
      }
#else
      (void)grid; (void)LDS; (void)qT; (void)doT;
      set_error("attn_bwd: the two-pass dK/dV kernels are compiled into development builds only (make DEV=1)");
      return HVLM_ERR_UNSUPPORTED;
#endif
    }
  }
  return HVLM_OK;
}

}  // namespace
}  // namespace hvlm

using namespace hvlm;

// The kernels gather every transposed operand with LDS transpose reads: the forward needs no workspace any more (a
// token amount is returned so callers keep passing a valid pointer), the backward only delta[B,H,S] fp32.
extern "C" size_t hvlm_attn_fwd_ws_bytes(int B, int S, int H, int dh) {
  (void)B; (void)S; (void)H; (void)dh;
  return 1024;            // the eight per-XCD work-queue counters of the persistent dh = 128 forward (64 bytes apart), zeroed per call
}
extern "C" size_t hvlm_attn_bwd_ws_bytes(int B, int S, int H, int dh) {
  (void)dh;
  return (size_t)B * H * S * 4 + 256;
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


// Variable-length (packed) causal attention — the contract of the reference's kernel swap,
// llava/train/llama_flash_attn_monkey_patch.py:81-96: unpad_input -> flash_attn_varlen_qkvpacked_func(qkv [total, 3, H, dh],
// cu_seqlens [B+1] int32, max_seqlen, causal=True) -> pad_input.  Here the engine keeps the tokens packed end to end, so
// there is no unpad / pad step at all.
extern "C" int hvlm_attn_fwd_varlen(const void* qkv, void* out, float* lse, void* ws, const int32_t* cu_seqlens, int B,
                                    int max_seqlen, int H, int dh, int ld, int causal, float scale, void* stream) {
  HVLM_REQUIRE(qkv && out && ws && cu_seqlens, "attn_fwd_varlen: null pointer");
  HVLM_REQUIRE(B > 0 && max_seqlen > 0 && H > 0 && ld >= 3 * H * dh && ld % 8 == 0, "attn_fwd_varlen: bad shape");
  HVLM_REQUIRE(dh == 64 || dh == 128, "attn_fwd_varlen: dh must be 64 or 128 (got %d)", dh);
  hipStream_t st = as_stream(stream);
  const bf16_t* q = (const bf16_t*)qkv;
  if (dh == 128) return causal ? fwd_impl<128, true>(q, (bf16_t*)out, lse, (bf16_t*)ws, B, max_seqlen, H, ld, scale, st, cu_seqlens)
                               : fwd_impl<128, false>(q, (bf16_t*)out, lse, (bf16_t*)ws, B, max_seqlen, H, ld, scale, st, cu_seqlens);
  return causal ? fwd_impl<64, true>(q, (bf16_t*)out, lse, (bf16_t*)ws, B, max_seqlen, H, ld, scale, st, cu_seqlens)
                : fwd_impl<64, false>(q, (bf16_t*)out, lse, (bf16_t*)ws, B, max_seqlen, H, ld, scale, st, cu_seqlens);
}

extern "C" int hvlm_attn_bwd_varlen(const void* qkv, const void* out, const void* dout, const float* lse, void* dqkv,
                                    void* ws, const int32_t* cu_seqlens, int B, int max_seqlen, long total_tokens, int H,
                                    int dh, int ld, int causal, float scale, void* stream) {
  HVLM_REQUIRE(qkv && out && dout && lse && dqkv && ws && cu_seqlens, "attn_bwd_varlen: null pointer");
  HVLM_REQUIRE(B > 0 && max_seqlen > 0 && total_tokens > 0 && H > 0 && ld >= 3 * H * dh && ld % 8 == 0, "attn_bwd_varlen: bad shape");
  HVLM_REQUIRE(dh == 64 || dh == 128, "attn_bwd_varlen: dh must be 64 or 128 (got %d)", dh);
  HVLM_REQUIRE(((uintptr_t)ws & 15) == 0, "attn_bwd_varlen: workspace must be 16-byte aligned");
  hipStream_t st = as_stream(stream);
  const bf16_t* q = (const bf16_t*)qkv;
  const bf16_t* o = (const bf16_t*)out;
  const bf16_t* g = (const bf16_t*)dout;
  if (dh == 128) return causal ? bwd_impl<128, true>(q, o, g, lse, (bf16_t*)dqkv, (char*)ws, B, max_seqlen, H, ld, scale, st, cu_seqlens, total_tokens)
                               : bwd_impl<128, false>(q, o, g, lse, (bf16_t*)dqkv, (char*)ws, B, max_seqlen, H, ld, scale, st, cu_seqlens, total_tokens);
  return causal ? bwd_impl<64, true>(q, o, g, lse, (bf16_t*)dqkv, (char*)ws, B, max_seqlen, H, ld, scale, st, cu_seqlens, total_tokens)
                : bwd_impl<64, false>(q, o, g, lse, (bf16_t*)dqkv, (char*)ws, B, max_seqlen, H, ld, scale, st, cu_seqlens, total_tokens);
}


// Attention backward with the gradient of apply_rotary_pos_emb folded into the dQ / dK epilogues: `dqkv` comes out as the
// gradient with respect to the UN-rotated q / k (what the QKV projection's backward consumes), bit-identical to
// hvlm_attn_bwd[_varlen] followed by hvlm_rope_qk(dqkv, ..., inverse = 1).  cu_seqlens may be NULL (padded batch; total_tokens
// is then ignored).  cos_tab / sin_tab: bf16 [max_pos, dh/2]; pos: int32 per token row.
extern "C" int hvlm_attn_bwd_rope(const void* qkv, const void* out, const void* dout, const float* lse, void* dqkv, void* ws,
                                  const int32_t* cu_seqlens, int B, int max_seqlen, long total_tokens, int H, int dh, int ld,
                                  int causal, float scale, const void* cos_tab, const void* sin_tab, const int32_t* pos,
                                  void* stream) {
  HVLM_REQUIRE(qkv && out && dout && lse && dqkv && ws && cos_tab && sin_tab && pos, "attn_bwd_rope: null pointer");
  HVLM_REQUIRE(B > 0 && max_seqlen > 0 && H > 0 && ld >= 3 * H * dh && ld % 8 == 0, "attn_bwd_rope: bad shape");
  HVLM_REQUIRE(!cu_seqlens || total_tokens > 0, "attn_bwd_rope: total_tokens must be given with cu_seqlens");
  HVLM_REQUIRE(dh == 64 || dh == 128, "attn_bwd_rope: dh must be 64 or 128 (got %d)", dh);
  HVLM_REQUIRE(((uintptr_t)ws & 15) == 0, "attn_bwd_rope: workspace must be 16-byte aligned");
  hipStream_t st = as_stream(stream);
  const bf16_t* q = (const bf16_t*)qkv;
  const bf16_t* o = (const bf16_t*)out;
  const bf16_t* g = (const bf16_t*)dout;
  const bf16_t* c = (const bf16_t*)cos_tab;
  const bf16_t* sn = (const bf16_t*)sin_tab;
  if (dh == 128) return causal ? bwd_impl<128, true>(q, o, g, lse, (bf16_t*)dqkv, (char*)ws, B, max_seqlen, H, ld, scale, st, cu_seqlens, total_tokens, c, sn, pos)
                               : bwd_impl<128, false>(q, o, g, lse, (bf16_t*)dqkv, (char*)ws, B, max_seqlen, H, ld, scale, st, cu_seqlens, total_tokens, c, sn, pos);
  return causal ? bwd_impl<64, true>(q, o, g, lse, (bf16_t*)dqkv, (char*)ws, B, max_seqlen, H, ld, scale, st, cu_seqlens, total_tokens, c, sn, pos)
                : bwd_impl<64, false>(q, o, g, lse, (bf16_t*)dqkv, (char*)ws, B, max_seqlen, H, ld, scale, st, cu_seqlens, total_tokens, c, sn, pos);
}

// fp16 instantiation (IEEE half q/k/v/out, v_mfma_f32_32x32x16_f16; P is rounded to fp16 for the PV product): the CLIP
// tower's attention in the reference's own dtype (handsonvlm/train/train.py:316).  Non-causal, dh = 64 (CLIPAttention).
extern "C" int hvlm_attn_fwd_f16(const void* qkv, void* out, float* lse, void* ws, int B, int S, int H, int dh, int ld,
                                 int causal, float scale, void* stream) {
  HVLM_REQUIRE(qkv && out && ws, "attn_fwd_f16: null pointer");
  HVLM_REQUIRE(B > 0 && S > 0 && H > 0 && ld >= 3 * H * dh && ld % 8 == 0, "attn_fwd_f16: bad shape");
  HVLM_REQUIRE(dh == 64 && !causal, "attn_fwd_f16: the fp16 instantiation is the CLIP attention (dh = 64, non-causal)");
  return fwd_impl<64, false, true>((const bf16_t*)qkv, (bf16_t*)out, lse, (bf16_t*)ws, B, S, H, ld, scale, as_stream(stream));
}
