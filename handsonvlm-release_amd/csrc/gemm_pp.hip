// bf16 NT GEMM, "ping-pong" schedule for gfx950:  C[M,N] = act(A[M,K] @ B[N,K]^T + bias) + residual.
// Two schedules of the same structure: 8 phases per K-tile (described first; used by the K-strided TN instantiation) and
// 4 segments per K-tile (SEG4, the default for K-contiguous operands: see the comment at `if constexpr (SEG4)`).
//
// Tile geometry and LDS image: 256x256 output tile, BK = 64, 8 waves, 128-B LDS rows with the 16-B chunk permutation
// c ^ ((row>>1)&7), swapped-operand MFMAs (weight rows as MFMA A) so a lane owns one output row and 4 consecutive columns.
// The DEFAULT kernels (gemm_pp4m_kernel, gemm_pp4m_f16_kernel) issue v_mfma_f32_16x16x32 — 15 % more power-efficient than
// 32x32x16 on this power-capped chip, see the M16 comment at gemm_pp_body; the 32x32x16 form described below is kept as
// gemm_pp4_kernel (HVLM_GEMM_M16=0) and is what the TN and fp8 instantiations use.  The schedule:
//
//   * the two wave groups (waves 0-3 / 4-7 = one wave of each group on every SIMD) run ONE BARRIER APART:
//     while one group issues its MFMA cluster the other issues its LDS reads and LDS-DMA, then they swap.
//     Every phase is  {ds_read sub-tile ; stage one half-tile ; s_waitcnt vmcnt(8) ; barrier ; MFMA ; barrier};
//   * a K-tile is 4 phases = the 4 quadrants (64x32) of the wave's 128x64 output; the wave's rows / columns are
//     split in two 64-row / 32-column halves 128 apart, so that quadrant order consumes the operand HALF-TILES
//     (A rows 0-127 / 128-255, B rows 0-127 / 128-255; 16 KiB each) in the fixed order A0,B0 | B1 | A1 | -;
//   * each phase re-stages ONE half-tile (2 LDS-DMA instructions per lane) two K-tiles ahead, into the slot
//     whose last reader finished >= 2 phases earlier; 4 half-tiles (8 loads per lane) stay in flight across
//     barriers and the wait is always the counted `vmcnt(8)`, never 0: every load gets a full K-tile of latency;
//   * a half-tile is read one phase after the wait that retires it (both groups pass a barrier in between);
//   * LDS-DMA uses buffer loads (`buffer_load_dwordx4 ... lds`): per-lane offsets are tile-independent 32-bit
//     values, the tile base lives in the (scalar) descriptor, rows past M / N fall outside num_records;
//   * persistent workgroups walk tiles in an XCD-aware grouped order (tile_coords); the staging cursor runs across
//     tile boundaries, so the epilogue of tile i overlaps the loads of tile i+1.  The two groups re-align at the
//     epilogue (group 1 skips its last barrier, re-staggers after it) so both epilogues run concurrently;
//   * epilogue through a wave-private LDS transpose (16-B row-contiguous stores / residual loads), see `epilogue`;
//   * tiles are assigned dynamically (per-XCD atomic counters + stealing) when there are >= 3 per workgroup, see PPSched;
//   * the same body is instantiated for K-strided operands (TN: dW = dY^T X) and for fp8 e4m3 operands on the scaled
//     MFMA (v_mfma_scale_f32_32x32x64_f8f6f4), see gemm_pp_body;
//   * SEG4: the four 8-MFMA phases become two 16-MFMA segments (output rows of half 0, then half 1, each x both column
//     fragments), 4 barriers per K-tile instead of 8: MFMA busy 70.5 % -> 78.3 %, cycles -10 %, time -3..4 % (the kernel is
//     clock/power-limited: profiles/r01p_gemm_pp4_pmc.txt);
//   * instantiations: bf16 / fp16 (CLIP tower) / e4m3 operands, K-contiguous (NT) or K-strided (TN), 8-phase or 4-segment
//     schedule, 32x32x16 or 16x16x32 (fp8: 32x32x64 or 16x16x128) MFMAs — one body, template flags.
#include "common.h"
#include "gemm_args.h"
#include <type_traits>

namespace hvlm {

namespace {

constexpr int BM = 256, BN = 256, BK = 64;
constexpr unsigned HALF_BYTES = 128 * BK * 2;      // 16 KiB: 128 rows x 128 B
constexpr unsigned TILE_BYTES = 2 * HALF_BYTES;    // one operand, one K-tile
constexpr unsigned STAGE_BYTES = 2 * TILE_BYTES;   // A + B
constexpr unsigned RING_BYTES = 2 * STAGE_BYTES;    // 128 KiB operand ring
constexpr unsigned EPI_BYTES = 4096;                // per-wave epilogue transpose buffer (one 32x32 fp32 fragment)
constexpr unsigned PP_LDS_BYTES = RING_BYTES + 8 * EPI_BYTES;   // 160 KiB: the whole LDS of a CU

typedef __attribute__((address_space(3))) void* lds_ptr_t;

// x * sigmoid(1.702 x) with the hardware exp2 / rcp (1 ulp each; the result is rounded to bf16 anyway)
__device__ __forceinline__ float quick_gelu_pp(float x) {
  return x * __builtin_amdgcn_rcpf(1.f + __builtin_amdgcn_exp2f(-1.702f * 1.4426950408889634f * x));
}

// 16-byte global store the compiler's wait-count tracking does not see (gfx9 reads store data at issue, so the
// registers are free immediately).  With tracked stores pending at the top of the K loop the compiler guards every
// first ds_read of a K-tile with its own `s_waitcnt vmcnt(0)` (it cannot order them against the LDS-DMA in flight).
// The trailing s_nop is the ISA's "VMEM store of more than 64 bits of data -> write of those data VGPRs" hazard: the hardware keeps
// reading the upper data dwords for a cycle or two after issue, and the compiler — which inserts the wait state itself behind a
// store it can see — does not look inside inline asm.  Without it the first dword of a row's 16 bytes was clobbered whenever the
// register allocator happened to reuse v[data] for the next address computation right behind the store (round 4: seen the moment the
// epilogue loop was specialised and rescheduled; before that the instruction order hid it).
__device__ __forceinline__ void store16_untracked(void* ptr, u32x4 v) {
  asm volatile("global_store_dwordx4 %0, %1, off\n\ts_nop 1" ::"v"(ptr), "v"(v) : "memory");
}

__device__ __forceinline__ void store8_untracked(void* ptr, u32x2 v) {
  asm volatile("global_store_dwordx2 %0, %1, off" ::"v"(ptr), "v"(v) : "memory");
}
// eight bf16 (as stored by the epilogue) * s -> eight e4m3 bytes, saturating
__device__ __forceinline__ u32x2 bf16x8_to_e4m3(u32x4 o, float s) {
  float f[8];
#pragma unroll
  for (int e = 0; e < 4; ++e) { f[2 * e] = bf_lo(o[e]) * s; f[2 * e + 1] = bf_hi(o[e]) * s; }
#pragma unroll
  for (int e = 0; e < 8; ++e) f[e] = fminf(fmaxf(f[e], -448.f), 448.f);
  unsigned lo = 0, hi = 0;
  lo = __builtin_amdgcn_cvt_pk_fp8_f32(f[0], f[1], lo, false);
  lo = __builtin_amdgcn_cvt_pk_fp8_f32(f[2], f[3], lo, true);
  hi = __builtin_amdgcn_cvt_pk_fp8_f32(f[4], f[5], hi, false);
  hi = __builtin_amdgcn_cvt_pk_fp8_f32(f[6], f[7], hi, true);
  return u32x2{lo, hi};
}

// Dynamic tile scheduling state (GemmArgs::dyn): one tile counter per XCD chunk of the swizzled tile order, a completion
// counter (the last workgroup to finish resets everything for the next launch) and, per workgroup, a 4-deep ring of
// fetched tile indices through which wave 0 hands its fetches to the other waves one tile ahead of their use.
// Launches that use it must be serialised on one stream.
struct PPSched {
  unsigned ctr[8][32];      // [xcd][0], 128 B apart
  unsigned done[32];
  int slots[512][4];
};
constexpr int PP_NSCHED = 8;             // one state per stream that launches this kernel (gemm_args.h: pp_sched_slot)
__device__ PPSched g_pp_all[PP_NSCHED];

#define HVLM_VMCNT(n) asm volatile("s_waitcnt vmcnt(" #n ")" ::: "memory")

// -DHVLM_GEMM_PROF (tools/gemm_tile_timeline.sh; never in the shipped libraries): s_memtime stamps of the tile cycle — for the
// workgroups on XCD 0's first 8 CUs, every wave, the first 32 tiles: [0] loop top, [1] K loop done, [2] epilogue entered, [3] epilogue
// done, [4] next-unit bookkeeping done, [5] first K-tile done, [6..9] residual epilogue: row fragment i entered, [10] its loop left.  The stamp stores are untracked 8-byte stores by lane 0.
#ifdef HVLM_GEMM_PROF
__device__ unsigned long long g_pp_prof[8][8][32][12];
#define HVLM_PP_STAMP(k)                                                                                              \
  do {                                                                                                                \
    if ((blockIdx.x & 7) == 0 && blockIdx.x < 64 && it < 32 && l == 0) {                                             \
      const unsigned long long t_ = __builtin_readcyclecounter();                                                     \
      asm volatile("global_store_dwordx2 %0, %1, off" ::"v"(&g_pp_prof[blockIdx.x >> 3][w][it][k]), "v"(t_) : "memory"); \
    }                                                                                                                 \
  } while (0)
#else
#define HVLM_PP_STAMP(k) do {} while (0)
#endif

// TN = both operands K-strided: A is [K, M] (element (m,k) at A[k*lda + m]), B is [K, N].  This is dW = dY^T X on the
// activations exactly as the forward / backward passes left them (token-major): no transposed copies.  What changes:
//   * a half-tile is 64 k-rows x 128 columns (256-B LDS rows), a DMA piece = 4 k-rows;
//   * MFMA fragments (8 consecutive k for one column) are gathered with `ds_read_b64_tr_b16` (a 16-lane group reads a
//     4(k) x 16(col) block, lane q receives column q): two reads per fragment, same LDS bytes as the NT path's b128;
//   * the 64-B segment of a row is XOR-ed with (k & 3): the four k-rows of one transpose read sit in four bank quarters;
//   * the descriptor is re-based every K-tile and num_records ends at row K, so the K tail (K = tokens, any
//     multiple of 8... any value) reads as zeros; columns past M / N read neighbouring data that is never stored.
// (a __device__ template under two plain __global__ wrappers: instantiating a __global__ TEMPLATE whose body holds
// inline asm with AMDGPU register constraints fails silently in the host pass and leaves its launch stub undefined)
//
// FP8 = both operands are OCP e4m3 bytes ([M, K] / [N, K], K contiguous), C = alpha * A B^T (+ epilogue): the CDNA4
// `v_mfma_scale_f32_32x32x64_f8f6f4` path (K = 64 per instruction at twice the bf16 rate; block scales fixed to 2^0 —
// per-tensor scaling enters through `alpha`).  A K-tile is 128 fp8 = the same 128 LDS bytes per row as 64 bf16, so the
// LDS image, the DMA pieces, the swizzle and the whole 8-phase schedule are unchanged; a fragment is two ds_read_b128
// (32 k-bytes per lane), 4 MFMAs of 64 cycles per phase instead of 8 of 32.
typedef __attribute__((ext_vector_type(8))) int i32x8;
// M16: the MFMAs of the K loop are v_mfma_f32_16x16x32 (16 output rows x 16 columns x 32 k) instead of 32x32x16 — the same
// flops per cycle, but half the accumulator traffic per flop: under the package power cap the matrix pipe alone sustains
// 2140 TFLOP/s on random bf16 data with 16x16x32 against 1854 with 32x32x16 (tools/mfma_power_probe.hip; 2480 either way on
// zeros), and this kernel is power-limited.  Same LDS image, same 24 ds_read_b128 per K-tile and wave, same
// register count: a 32x32 accumulator fragment is simply held as four 16x16 quadrants (registers 4t..4t+3 = quadrant
// t = 2*(column half) + (row half)), so only the fragment lane mapping, the MFMA calls and the (row, column) a register
// stands for in the epilogue change.
template <bool TN, bool FP8, bool SEG4 = false, bool DBG = false, bool F16 = false, bool M16 = false, bool EPI_SWAP = false>
__device__ __forceinline__ void gemm_pp_body(const GemmArgs& p) {
  static_assert(!M16 || (SEG4 && !TN), "the 16x16 form exists for the K-contiguous 4-segment kernels (16-bit: 16x16x32, fp8: 16x16x128)");
  static_assert(!(TN && FP8), "fp8 operands must be K-contiguous");
  static_assert(!(F16 && (TN || FP8)), "the fp16 instantiation (CLIP tower) is K-contiguous, 16-bit");
#ifdef HVLM_DEV
  constexpr bool HALF_OK = SEG4 && !TN && !DBG;     // half-tile tail units (development builds, gemm_tail_half = 1; see fetch_unit)
#else
  constexpr bool HALF_OK = false;                   // production: every unit is a whole tile, the code below folds away
#endif
  constexpr bool LNF = F16 && M16 && !EPI_SWAP && !TN && !FP8;   // LayerNorm-fold epilogue modes: the CLIP tower's fp16 kernel only
  using EL = E16<F16>;                      // 16-bit element type of A / B / C / bias / residual: bfloat16 or IEEE half
  constexpr int ES = FP8 ? 1 : 2;          // operand element size in bytes
  constexpr int NKS = FP8 ? 2 : 4;         // MFMA k-steps per K-tile
  using frag_t = std::conditional_t<FP8, i32x8, bf16x8>;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x;
  const int l = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l31 = l & 31, hi = l >> 5;
  const int grp = w >> 2, wn = w & 3;    // grp: row half-pair AND ping-pong group; wn: column block

  // ---- XCD-aware, grouped tile mapping (same order as gemm.hip) -------------------------------
  const int ntile = p.tiles_m * p.tiles_n;
  const int nwg = ntile * p.splits;
  // unit index in the swizzled order (workgroup b -> XCD b%8 gets a contiguous chunk) and its coordinates
  const int ch_q = nwg >> 3, ch_r = nwg & 7;
  auto chunk_start = [&](int x) { return x < ch_r ? x * (ch_q + 1) : ch_r * (ch_q + 1) + (x - ch_r) * ch_q; };
  auto swz_of = [&](int vb) { return chunk_start(vb & 7) + (vb >> 3); };
  // A unit is a tile or — dynamic scheduling, the last partial round of an XCD's chunk — one HALF of a tile: the unit code keeps
  // the swizzled tile index in bits 0..27 and, in bits 28..29, 0 = the whole tile, 1 / 2 = its rows 0-127 / 128-255 only.
  constexpr int UNIT_MASK = 0x0fffffff;
  auto tile_coords = [&](int swz, int& m0_, int& n0_, int& ksl_) {
    swz &= UNIT_MASK;
    ksl_ = swz / ntile;
    swz -= ksl_ * ntile;
    const int gsz = p.group_m * p.tiles_n;
    const int gid = swz / gsz;
    const int first_m = gid * p.group_m;
    const int gm = min(p.tiles_m - first_m, p.group_m);
    const int in = swz - gid * gsz;
    m0_ = (first_m + in % gm) * BM;
    n0_ = (in / gm) * BN;
  };
  // ---- dynamic scheduling (p.dyn): the next unit comes from the XCD's counter, then from the other XCDs' (stealing),
  // so a workgroup that starts late — CUs held by another stream, e.g. an RCCL all-reduce — simply takes fewer tiles.
  // With the static assignment (b, b+grid, ...) 8 busy CUs out of 256 cost this kernel 17-42 % (tools/contention_bench.py).
  const bool dyn = p.dyn != 0;
  PPSched& g_pp = (p.sched ? reinterpret_cast<PPSched*>(p.sched) : g_pp_all)[dyn ? p.dyn - 1 : 0];
  const int my_xcd = (int)blockIdx.x & 7;
  // HALF-TILE TAIL (p.tail_half): all tiles cost the same, so an XCD's chunk of cnt tiles on its W workgroups is floor(cnt / W)
  // full rounds and one partial round of t = cnt % W tiles during which W - t CUs idle (880 tiles on 256 CUs: 3.44 rounds, the
  // N = 4096 GEMMs of the decoder).  When t <= W / 2 those last t tiles are handed out as 2 t half-tile units (the K loop of a
  // half unit issues only the MFMA segment of its 128 rows; staging and barriers are unchanged), so the partial round costs about
  // two thirds of a tile-time instead of a whole one.  Every output element is still computed by one workgroup in the same
  // order: results are bit-identical (tests/test_gpu_ops.py).  MEASURED AND NOT ADOPTED (profiles/r03c_gemm_tail_half_ab.txt):
  // o_proj 13872x4096x4096 +3.6 %, qkv +1.7 %, every other shape within +-1 %, the SFT step unchanged — a half unit is not
  // cheaper than a whole tile, because the ping-pong schedule hides one group's load phase (ds_reads + LDS-DMA issue, ~500
  // cycles) under the OTHER group's 512-cycle MFMA segment: with one of the two MFMA segments gone the load phases are exposed
  // and the K-tile takes its four barrier-to-barrier phases all the same.  Kept in development builds only.
  const int wg_x = (int)gridDim.x >> 3;
  auto chunk_tail = [&](int cnt) {
    if (!(HALF_OK && p.tail_half && p.splits == 1 && wg_x > 0)) return 0;
    const int t = cnt % wg_x;
    return (t > 0 && 2 * t <= wg_x) ? t : 0;
  };
  // A fetch is split in two so that its round trip (1300 - 3200 cycles measured per tile, tools/gemm_tile_timeline.py: wave 0 stalled
  // in front of its epilogue and the other waves then waited for it at the next tile's first barrier) hides under the epilogue:
  // fetch_issue() = the atomic on the workgroup's own XCD counter, fetch_resolve(i) = everything that needs its result (and, once
  // that chunk is exhausted — the last tiles of a launch — the synchronous walk over the other XCDs' counters).
  auto unit_of = [&](int x, unsigned i) -> int {      // the unit ticket i of XCD x's counter stands for, -1: chunk exhausted
    const int cnt = ch_q + (x < ch_r ? 1 : 0);
    const int tl = chunk_tail(cnt), full = cnt - tl;
    if ((int)i < full) return chunk_start(x) + (int)i;
    if ((int)i < full + 2 * tl) {
      const int v = (int)i - full;
      return (chunk_start(x) + full + (v >> 1)) | ((1 + (v & 1)) << 28);
    }
    return -1;
  };
  // (atomic INC with a wrap bound of 2^32 - 1 = fetch-and-add 1: LLVM's atomic optimizer rewrites a one-lane atomic ADD into a wave
  // reduction whose broadcast — s_waitcnt vmcnt(0) ; v_readfirstlane — sits right behind the instruction, i.e. the stall this split
  // is meant to remove; it leaves INC alone, and the result is then waited for at its first use)
  auto fetch_issue = [&]() -> unsigned {    // one lane
    return __builtin_amdgcn_atomic_inc32(&g_pp.ctr[my_xcd][0], 0xffffffffu, __ATOMIC_RELAXED, "agent");
  };
  auto fetch_resolve = [&](unsigned i0) -> int {
    int u = unit_of(my_xcd, i0);
    for (int a = 1; a < 8 && u < 0; ++a) {
      const int x = (my_xcd + a) & 7;
      u = unit_of(x, __hip_atomic_fetch_add(&g_pp.ctr[x][0], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
    }
    return u;
  };
  auto fetch_unit = [&]() -> int { return fetch_resolve(fetch_issue()); };
  int t_cur, t_nxt;                        // swizzled unit index of the tile being computed / the one after (-1: none)
  constexpr int BKE = BK * 2 / ES;          // K-tile in elements: 64 bf16 or 128 fp8 (128 bytes either way)
  const int nk = TN ? (p.K + BK - 1) / BK : (p.splits > 1 ? p.k_split : p.K) / BKE;

  // ---- LDS-DMA: per-lane byte offsets inside an operand tile (tile independent) -----------------
  // piece = 8 rows x 128 B written lane-linear; wave w feeds pieces 2w, 2w+1 of every half-tile.
  unsigned voa[2][2], vob[2][2];
#pragma unroll
  for (int h = 0; h < 2; ++h)
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      if constexpr (!TN) {
        const int row = h * 128 + (2 * w + i) * 8 + (l >> 3);
        const int c = (l & 7) ^ ((row >> 1) & 7);              // source chunk for chunk position l&7
        voa[h][i] = (unsigned)row * (unsigned)p.lda * ES + c * 16u;
        // B tile row r holds output column (r&127 >> 5)*64 + (r>>7)*32 + (r&31): a wave's two 32-column fragments
        // (one in each B half-tile) are ADJACENT in the output, 64 contiguous columns per wave.
        const int rr = row & 127;
        const int bcol = (rr >> 5) * 64 + h * 32 + (rr & 31);
        vob[h][i] = (unsigned)bcol * (unsigned)p.ldb * ES + c * 16u;
      } else {
        // piece = 4 k-rows x 256 B; lane -> k-row l>>4, chunk position l&15 holding source chunk cp ^ ((k&3)<<2)
        const int krow = (2 * w + i) * 4 + (l >> 4);
        const int c = (l & 15) ^ ((krow & 3) << 2);
        voa[h][i] = ((unsigned)krow * (unsigned)p.lda + h * 128 + c * 8) * 2u;
        const int bcol = (c >> 2) * 64 + h * 32 + (c & 3) * 8;  // same column permutation as the NT path
        vob[h][i] = ((unsigned)krow * (unsigned)p.ldb + bcol) * 2u;
      }
    }
  const unsigned dma_off = (unsigned)(2 * w) * 1024u;

  // staging cursor: descriptors of the tile being staged + K offset (bytes); all wave-uniform
  struct Cursor {
    __amdgpu_buffer_rsrc_t ra, rb;
    unsigned koff;
    int vb, kt;        // vb: static mode: launch-order index of its tile; dynamic mode: 0 / 1 = on t_cur / t_nxt
    int t;             // swizzled unit index of its tile
  };
  auto make_desc = [&](Cursor& c) {
    int m0_, n0_, ksl_;
    tile_coords(c.t, m0_, n0_, ksl_);
    if constexpr (!TN) {
      const size_t kbase = (size_t)ksl_ * p.k_split;
      const size_t krem = (size_t)p.K - kbase;
      const size_t ba = ((size_t)(p.M - m0_ - 1) * p.lda + krem) * ES;
      const size_t bb = ((size_t)(p.N - n0_ - 1) * p.ldb + krem) * ES;
      char* pa = const_cast<char*>(reinterpret_cast<const char*>(p.A)) + ((size_t)m0_ * p.lda + kbase) * ES;
      char* pb = const_cast<char*>(reinterpret_cast<const char*>(p.B)) + ((size_t)n0_ * p.ldb + kbase) * ES;
      c.ra = __builtin_amdgcn_make_buffer_rsrc(pa, 0, ba > 0xffffffffull ? 0xffffffffu : (unsigned)ba, 0x00020000);
      c.rb = __builtin_amdgcn_make_buffer_rsrc(pb, 0, bb > 0xffffffffull ? 0xffffffffu : (unsigned)bb, 0x00020000);
    } else {
      // base = &A[kt*64][m0]; valid bytes = up to the end of row K-1 (rows >= K are out of range -> zeros)
      const size_t k0 = (size_t)c.kt * BK;
      const size_t ba = (((size_t)p.K - k0) * p.lda - m0_) * 2;
      const size_t bb = (((size_t)p.K - k0) * p.ldb - n0_) * 2;
      c.ra = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(p.A + k0 * p.lda + m0_), 0,
                                               ba > 0xffffffffull ? 0xffffffffu : (unsigned)ba, 0x00020000);
      c.rb = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(p.B + k0 * p.ldb + n0_), 0,
                                               bb > 0xffffffffull ? 0xffffffffu : (unsigned)bb, 0x00020000);
    }
  };
  auto advance = [&](Cursor& c) {
    if (++c.kt == nk) {
      c.kt = 0;
      if (dyn) {
        if (c.vb == 0 && t_nxt >= 0) { c.vb = 1; c.t = t_nxt; }      // (nk >= 3: a cursor is never two tiles ahead)
      } else {
        const int nv = c.vb + (int)gridDim.x;
        if (nv < nwg) { c.vb = nv; c.t = swz_of(nv); }
      }
      // past the last tile of this workgroup: keep loading this tile's K-tiles again (valid memory, slots nobody
      // reads) so the counted vmcnt stays exact.
      make_desc(c);
    } else if constexpr (TN) {
      make_desc(c);                          // re-base on the next 64 k-rows
    }
    c.koff = TN ? 0u : (unsigned)c.kt * (BK * 2);
  };
  auto stage_a = [&](unsigned buf, int h, const Cursor& c) {
    const unsigned base = buf * STAGE_BYTES + h * HALF_BYTES + dma_off;
    __builtin_amdgcn_raw_ptr_buffer_load_lds(c.ra, (lds_ptr_t)(smem + base), 16, voa[h][0], c.koff, 0, 0);
    __builtin_amdgcn_raw_ptr_buffer_load_lds(c.ra, (lds_ptr_t)(smem + base + 1024u), 16, voa[h][1], c.koff, 0, 0);
  };
  auto stage_b = [&](unsigned buf, int h, const Cursor& c) {
    const unsigned base = buf * STAGE_BYTES + TILE_BYTES + h * HALF_BYTES + dma_off;
    __builtin_amdgcn_raw_ptr_buffer_load_lds(c.rb, (lds_ptr_t)(smem + base), 16, vob[h][0], c.koff, 0, 0);
    __builtin_amdgcn_raw_ptr_buffer_load_lds(c.rb, (lds_ptr_t)(smem + base + 1024u), 16, vob[h][1], c.koff, 0, 0);
  };

  // ---- fragment read offsets ------------------------------------------------------------------
  // A fragment i (32 rows): tile row (i>>1)*128 + grp*64 + (i&1)*32 + l31;  B fragment j: j*128 + wn*32 + l31
  const int sw = (l31 >> 1) & 7;
  unsigned koffl[4];
#pragma unroll
  for (int ks = 0; ks < 4; ++ks) koffl[ks] = (unsigned)(((ks * 2 + hi) ^ sw) << 4);
  const unsigned a_base = (unsigned)(grp * 64 + l31) * 128u;
  const unsigned b_base = TILE_BYTES + (unsigned)(wn * 32 + l31) * 128u;
  // M16: a 16-row fragment: lane -> row (l & 15) of its 16-row block, k = 32*ks2 + 8*(l >> 4) .. +7 (one 16-B chunk); the
  // row swizzle only sees (l & 15) >> 1 (block bases are multiples of 16 rows).  f[s*2 + ks2] = 16-row block s, k half ks2.
  const int l15 = l & 15, q4 = l >> 4;
  unsigned koff16[2];
#pragma unroll
  for (int ks2 = 0; ks2 < 2; ++ks2) koff16[ks2] = (unsigned)(((q4 + 4 * ks2) ^ ((l15 >> 1) & 7)) << 4);
  unsigned koff816[2];      // fp8, 16x16x128: the 32 k-bytes of a lane = 16-B chunks 2*q4, 2*q4 + 1 of its row (the whole K-tile)
#pragma unroll
  for (int e = 0; e < 2; ++e) koff816[e] = (unsigned)(((2 * q4 + e) ^ ((l15 >> 1) & 7)) << 4);
  auto read816 = [&](unsigned base) -> i32x8 {
    const u32x4 lo = *reinterpret_cast<const u32x4*>(smem + base + koff816[0]);
    const u32x4 hi4 = *reinterpret_cast<const u32x4*>(smem + base + koff816[1]);
    const i32x8 v = {(int)lo[0], (int)lo[1], (int)lo[2], (int)lo[3], (int)hi4[0], (int)hi4[1], (int)hi4[2], (int)hi4[3]};
    return v;
  };
  const unsigned a_base16 = (unsigned)(grp * 64 + l15) * 128u;
  const unsigned b_base16 = TILE_BYTES + (unsigned)(wn * 32 + l15) * 128u;
  // TN: lane p of a 16-lane group addresses k-row (p>>2) [+ 8*hi + 4*rr + 16*ks], columns (p&3)*4.. of its 16-column
  // block and receives column l31 of the fragment; the 16-B chunk index is XOR-ed with ((k&3)<<2)
  const int p16 = l & 15, g16 = l31 >> 4;
  unsigned ta_off[2], tb_off;
  {
    const unsigned rowb = (unsigned)((p16 >> 2) + 8 * hi) * 256u + (unsigned)((p16 & 1) * 8);
    const unsigned x = (unsigned)((p16 >> 2) & 3) << 2;
#pragma unroll
    for (int e = 0; e < 2; ++e)
      ta_off[e] = rowb + (((unsigned)(grp * 8 + e * 4 + g16 * 2 + ((p16 & 3) >> 1)) ^ x) << 4);
    tb_off = TILE_BYTES + rowb + (((unsigned)(wn * 4 + g16 * 2 + ((p16 & 3) >> 1)) ^ x) << 4);
  }
  // The transpose reads are issued by inline asm: the `ds_read_tr16_b64` builtin is modelled as a may-store LDS access,
  // and the compiler guards every such access with `s_waitcnt vmcnt(0)` while an LDS-DMA of the same wave is pending —
  // which is always, in this pipeline.  The wait for their results is the explicit lgkmcnt(0) in HVLM_MFMA_QUAD.
  typedef __attribute__((ext_vector_type(4))) short s16x4_t;
  typedef __attribute__((ext_vector_type(8))) short s16x8_t;
  const unsigned lds0 = (unsigned)(unsigned long)(lds_ptr_t)smem;
#define HVLM_TR2(DST, ADDR, IMM)                                                                            \
  do {                                                                                                      \
    s16x4_t lo_, hi_;                                                                                       \
    asm volatile("ds_read_b64_tr_b16 %0, %2 offset:%3\n\tds_read_b64_tr_b16 %1, %2 offset:%4"              \
                 : "=&v"(lo_), "=&v"(hi_) : "v"(ADDR), "n"(IMM), "n"((IMM) + 1024) : "memory");             \
    const s16x8_t v_ = {lo_[0], lo_[1], lo_[2], lo_[3], hi_[0], hi_[1], hi_[2], hi_[3]};                    \
    DST = __builtin_bit_cast(bf16x8, v_);                                                                   \
  } while (0)
  // fp8: the 32 k-bytes of a lane are the two 16-B chunks (ks*4 + hi*2, +1) of its row
  unsigned koff8[2][2];
#pragma unroll
  for (int ks = 0; ks < 2; ++ks)
#pragma unroll
    for (int e = 0; e < 2; ++e) koff8[ks][e] = (unsigned)(((ks * 4 + hi * 2 + e) ^ sw) << 4);
  auto read8 = [&](unsigned base, int ks) -> i32x8 {
    const u32x4 lo = *reinterpret_cast<const u32x4*>(smem + base + koff8[ks][0]);
    const u32x4 hi4 = *reinterpret_cast<const u32x4*>(smem + base + koff8[ks][1]);
    const i32x8 v = {(int)lo[0], (int)lo[1], (int)lo[2], (int)lo[3], (int)hi4[0], (int)hi4[1], (int)hi4[2], (int)hi4[3]};
    return v;
  };
  auto read_a = [&](unsigned sb, auto i_tag, frag_t (&f)[NKS]) {
    constexpr int i = decltype(i_tag)::value;
    if constexpr (FP8 && M16) {
#pragma unroll
      for (int sx = 0; sx < 2; ++sx) f[sx] = read816(sb + a_base16 + (unsigned)(i >> 1) * HALF_BYTES + (unsigned)(i & 1) * 4096u + (unsigned)sx * 2048u);
    } else if constexpr (FP8) {
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) f[ks] = read8(sb + a_base + (unsigned)(i >> 1) * HALF_BYTES + (unsigned)(i & 1) * 4096u, ks);
    } else if constexpr (M16) {
#pragma unroll
      for (int sx = 0; sx < 2; ++sx)
#pragma unroll
        for (int ks2 = 0; ks2 < 2; ++ks2)
          f[sx * 2 + ks2] = *reinterpret_cast<const bf16x8*>(smem + sb + a_base16 + (unsigned)(i >> 1) * HALF_BYTES + (unsigned)(i & 1) * 4096u + (unsigned)sx * 2048u + koff16[ks2]);
    } else if constexpr (!TN) {
#pragma unroll
      for (int ks = 0; ks < 4; ++ks)
        f[ks] = *reinterpret_cast<const bf16x8*>(smem + sb + a_base + (unsigned)(i >> 1) * HALF_BYTES + (unsigned)(i & 1) * 4096u + koffl[ks]);
    } else {
      const unsigned ad = lds0 + sb + ta_off[i & 1];
      HVLM_TR2(f[0], ad, (i >> 1) * 16384 + 0 * 4096);
      HVLM_TR2(f[1], ad, (i >> 1) * 16384 + 1 * 4096);
      HVLM_TR2(f[2], ad, (i >> 1) * 16384 + 2 * 4096);
      HVLM_TR2(f[3], ad, (i >> 1) * 16384 + 3 * 4096);
    }
  };
  auto read_b = [&](unsigned sb, auto j_tag, frag_t (&f)[NKS]) {
    constexpr int j = decltype(j_tag)::value;
    if constexpr (FP8 && M16) {
#pragma unroll
      for (int sx = 0; sx < 2; ++sx) f[sx] = read816(sb + b_base16 + (unsigned)j * HALF_BYTES + (unsigned)sx * 2048u);
    } else if constexpr (FP8) {
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) f[ks] = read8(sb + b_base + (unsigned)j * HALF_BYTES, ks);
    } else if constexpr (M16) {
#pragma unroll
      for (int sx = 0; sx < 2; ++sx)
#pragma unroll
        for (int ks2 = 0; ks2 < 2; ++ks2)
          f[sx * 2 + ks2] = *reinterpret_cast<const bf16x8*>(smem + sb + b_base16 + (unsigned)j * HALF_BYTES + (unsigned)sx * 2048u + koff16[ks2]);
    } else if constexpr (!TN) {
#pragma unroll
      for (int ks = 0; ks < 4; ++ks)
        f[ks] = *reinterpret_cast<const bf16x8*>(smem + sb + b_base + (unsigned)j * HALF_BYTES + koffl[ks]);
    } else {
      const unsigned ad = lds0 + sb + tb_off;
      HVLM_TR2(f[0], ad, j * 16384 + 0 * 4096);
      HVLM_TR2(f[1], ad, j * 16384 + 1 * 4096);
      HVLM_TR2(f[2], ad, j * 16384 + 2 * 4096);
      HVLM_TR2(f[3], ad, j * 16384 + 3 * 4096);
    }
  };
  using T0 = std::integral_constant<int, 0>; using T1 = std::integral_constant<int, 1>;
  using T2 = std::integral_constant<int, 2>; using T3 = std::integral_constant<int, 3>;

  f32x16 acc[2][4];                       // [column half j][row fragment i]
  unsigned it = 0;                        // tiles done by this workgroup

  // ---- epilogues: register quad g (registers 4g..4g+3) of a 32x32 fragment = 4 consecutive columns of ONE row:
  //   32x32x16: row l31 of the fragment, columns 8g + 4hi ..;   16x16x32 (M16): quadrant g = 2*(column half) + (row half):
  //   row (g & 1)*16 + (l & 15), columns (g >> 1)*16 + 4*(l >> 4) ..
  auto qrow = [&](int g) { return M16 ? ((g & 1) * 16 + l15) : l31; };
  auto qcol = [&](int g) { return M16 ? ((g >> 1) * 16 + 4 * q4) : (8 * g + 4 * hi); };
  auto col_of = [&](int n0, int j, int g) { return n0 + wn * 64 + j * 32 + qcol(g); };
  auto epilogue_generic = [&](int m0, int n0, size_t c_slab, int i_lo, int i_hi) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      if (i < i_lo || i >= i_hi) continue;
#pragma unroll
      for (int j = 0; j < 2; ++j) {
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const int m = m0 + (i >> 1) * 128 + grp * 64 + (i & 1) * 32 + qrow(g);
          if (m >= p.M) continue;
          const int nb = col_of(n0, j, g);
          if (nb >= p.N) continue;
          float v[4];
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] = acc[j][i][4 * g + e];
          const bool full = (nb + 3 < p.N);
          if (p.bias) {
            for (int e = 0; e < 4; ++e) if (nb + e < p.N) v[e] += EL::to_f(p.bias[nb + e]);
          }
          if (p.act == HVLM_ACT_QUICK_GELU) {
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = quick_gelu_pp(v[e]);
          }
          if (p.res) {
            const bf16_t* rp = p.res + (size_t)m * p.ldr + nb;
            for (int e = 0; e < 4; ++e) if (nb + e < p.N) v[e] += EL::to_f(rp[e]);
          }
          if constexpr (FP8) {
            if (p.out_inv) {                 // e4m3 output (see GemmArgs::out_inv)
              const float sq = 1.f / p.out_inv[m];
              unsigned char* cp = reinterpret_cast<unsigned char*>(p.C) + (size_t)m * p.ldc + nb;
              const u32x4 ob = {EL::pack2(v[0], v[1]), EL::pack2(v[2], v[3]), 0u, 0u};
              const u32x2 o8 = bf16x8_to_e4m3(ob, sq);
              for (int e = 0; e < 4; ++e) if (nb + e < p.N) cp[e] = (unsigned char)((o8[0] >> (8 * e)) & 0xffu);
              continue;
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
              u32x2 o = {EL::pack2(v[0], v[1]), EL::pack2(v[2], v[3])};
              *reinterpret_cast<u32x2*>(cp) = o;
            } else {
              for (int e = 0; e < 4; ++e) if (nb + e < p.N) cp[e] = EL::from_f(v[e]);
            }
          }
        }
      }
    }
  };
  // Fast path (tile inside N, bf16 output).  In the accumulator layout a lane owns ONE row and 4-column quads, so a
  // direct store instruction touches 32 different cache lines with 8 bytes each; measured, that epilogue cost ~8 us
  // per 256x256 tile (30 % of a K = 1024 GEMM).  Here every 32x32 fragment takes a round trip through a wave-private
  // 4 KiB LDS buffer (fp32, 16-B chunks XOR-swizzled by row) and comes back with 8 consecutive columns per lane and
  // 4 lanes per row: stores, residual and bias loads are 16 B per lane, 16 rows x 64 contiguous bytes per instruction.
  char* const epi = smem + RING_BYTES + w * EPI_BYTES;
  const int t_row = l >> 2, t_c = l & 3;                       // transposed layout: row (0..15) + 16*q, columns 8*t_c..+7
  const bool fast_ok = !p.out_f32 && (p.ldc & 7) == 0 && (!p.bias || ((uintptr_t)p.bias & 15) == 0) &&
                       (!p.res || ((p.ldr & 7) == 0 && ((uintptr_t)p.res & 15) == 0));
  // Register-exchange epilogue (EPI_SWAP; development builds, gemm_epi_swap = 1): the LDS round trip of the fast path below
  // replaced by lane swaps.  Measured bit-identical and 0.4 - 2.7 % SLOWER than the LDS path on every shape of the SFT step
  // (profiles/r03b_gemm_epilogue_swap_ab.txt): the epilogue is bound by VALU issue and by the 128 store instructions of a tile
  // (1 KB each against the CU's 64 B / clk store path), not by the LDS round trip, which the software-pipelined LDS path hides.
  // After bias / activation / residual and the 16-bit packing a lane holds, for one output row, two 4-column groups that are 16
  // (16x16x32 layout: quadrants g and g + 2) or 4 (32x32x16 layout: quads 2k and 2k + 1, other half-wave) columns apart; ONE
  // v_permlane16_swap (resp. v_permlane32_swap) per dword exchanges them with the partner lane so that every lane ends up with
  // 8 CONSECUTIVE columns = one 16-byte store, rows x 64 (32) contiguous bytes per instruction — the store shape of the
  // LDS-transposed path without its 4 ds_write_b64 + 2 ds_read_b128 + waits per fragment.  The residual is loaded in the store
  // layout (16 B per lane) and brought to the accumulator layout by the same swap (it is an involution), so the add stays in
  // fp32 before the single rounding: bit-identical to the LDS path.
  //   16x16x32: pair pr = row half: quads (pr, pr + 2), row pr*16 + (l & 15); v_permlane16_swap(vdst = quad pr, src = quad pr + 2)
  //             swaps vdst's odd 16-lane rows with src's even ones -> lane (l15, q4) holds columns (q4 & 1)*16 + (q4 >> 1)*8 .. + 7
  //   32x32x16: pair pr = quads (2 pr, 2 pr + 1), row l & 31; v_permlane32_swap -> lane (l31, hi) holds columns 16 pr + 8 hi .. + 7
  auto lane_swap = [&](unsigned& a, unsigned& b) {
    if constexpr (M16) { const auto r = __builtin_amdgcn_permlane16_swap(a, b, false, false); a = r[0]; b = r[1]; }
    else { const auto r = __builtin_amdgcn_permlane32_swap(a, b, false, false); a = r[0]; b = r[1]; }
  };
  auto epilogue_swap = [&](int m0, int n0, int i_lo, int i_hi) {
    const int nb0 = n0 + wn * 64;
    // bias in the accumulator layout: 4 columns per quad g (the same columns for every row fragment)
    u32x2 bias_q[2][4];
    if (p.bias) {
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int g4 = 0; g4 < 4; ++g4) bias_q[j][g4] = *reinterpret_cast<const u32x2*>(p.bias + col_of(n0, j, g4));
    }
    auto srow = [&](int pr) { return M16 ? (pr * 16 + l15) : l31; };                                    // row inside the 32-row fragment
    auto scol = [&](int pr) { return M16 ? ((q4 & 1) * 16 + (q4 >> 1) * 8) : (16 * pr + 8 * hi); };       // first of the lane's 8 columns
    const bool has_res = p.res != nullptr;
    u32x4 res_n[2][2];                        // residual of the NEXT row fragment, store layout: [j][pr]
    auto load_res = [&](int i, u32x4 (&dst)[2][2]) {
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int pr = 0; pr < 2; ++pr) {
          const int mc = min(m0 + (i >> 1) * 128 + grp * 64 + (i & 1) * 32 + srow(pr), p.M - 1);
          dst[j][pr] = *reinterpret_cast<const u32x4*>(p.res + (size_t)mc * p.ldr + nb0 + j * 32 + scol(pr));
        }
    };
    if (has_res) load_res(i_lo, res_n);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      if (i < i_lo || i >= i_hi) continue;
      u32x4 res_r[2][2];
      if (has_res) {
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
          for (int pr = 0; pr < 2; ++pr) res_r[j][pr] = res_n[j][pr];
        if (i + 1 < i_hi) load_res(i + 1, res_n);
      }
      float osc[2] = {1.f, 1.f};              // e4m3 output: 1 / out_inv of the lane's rows
      if constexpr (FP8) {
        if (p.out_inv) {
#pragma unroll
          for (int pr = 0; pr < 2; ++pr)
            osc[pr] = __builtin_amdgcn_rcpf(p.out_inv[min(m0 + (i >> 1) * 128 + grp * 64 + (i & 1) * 32 + srow(pr), p.M - 1)]);
        }
      }
#pragma unroll
      for (int j = 0; j < 2; ++j) {
#pragma unroll
        for (int pr = 0; pr < 2; ++pr) {
          const int ga = M16 ? pr : 2 * pr, gb = M16 ? pr + 2 : 2 * pr + 1;
          float va[4], vb[4];
#pragma unroll
          for (int e = 0; e < 4; ++e) { va[e] = acc[j][i][4 * ga + e]; vb[e] = acc[j][i][4 * gb + e]; }
          if (p.bias) {
            const u32x2 ba = bias_q[j][ga], bb = bias_q[j][gb];
            va[0] += EL::lo(ba[0]); va[1] += EL::hi(ba[0]); va[2] += EL::lo(ba[1]); va[3] += EL::hi(ba[1]);
            vb[0] += EL::lo(bb[0]); vb[1] += EL::hi(bb[0]); vb[2] += EL::lo(bb[1]); vb[3] += EL::hi(bb[1]);
          }
          if (p.act == HVLM_ACT_QUICK_GELU) {
#pragma unroll
            for (int e = 0; e < 4; ++e) { va[e] = quick_gelu_pp(va[e]); vb[e] = quick_gelu_pp(vb[e]); }
          }
          if (has_res) {
            u32x4 rr = res_r[j][pr];
            asm volatile("" : "+v"(rr));       // consume the load on every path: nothing tracked stays pending into the K loop
            unsigned r0 = rr[0], r1 = rr[1], r2 = rr[2], r3 = rr[3];
            lane_swap(r0, r2);                 // store layout -> accumulator layout (the swap is its own inverse)
            lane_swap(r1, r3);
            va[0] += EL::lo(r0); va[1] += EL::hi(r0); va[2] += EL::lo(r1); va[3] += EL::hi(r1);
            vb[0] += EL::lo(r2); vb[1] += EL::hi(r2); vb[2] += EL::lo(r3); vb[3] += EL::hi(r3);
          }
          unsigned a0 = EL::pack2(va[0], va[1]), a1 = EL::pack2(va[2], va[3]);
          unsigned b0 = EL::pack2(vb[0], vb[1]), b1 = EL::pack2(vb[2], vb[3]);
          lane_swap(a0, b0);
          lane_swap(a1, b1);
          const u32x4 o = {a0, a1, b0, b1};
          const int m = m0 + (i >> 1) * 128 + grp * 64 + (i & 1) * 32 + srow(pr);
          const int nc = nb0 + j * 32 + scol(pr);
          bool done8 = false;
          if constexpr (FP8) {
            if (p.out_inv) {
              if (m < p.M) store8_untracked(reinterpret_cast<unsigned char*>(p.C) + (size_t)m * p.ldc + nc, bf16x8_to_e4m3(o, osc[pr]));
              done8 = true;
            }
          }
          if (!done8 && m < p.M) store16_untracked(reinterpret_cast<bf16_t*>(p.C) + (size_t)m * p.ldc + nc, o);
        }
      }
    }
  };
  // PEEL (the 16x16 kernels, see the tile loop): every epilogue path holds ONE full vmcnt(0) before its first store — behind the loads
  // it needs itself (bias, LayerNorm-fold vectors, first residual fragment), so that their latency and the landing of the pieces
  // staged for the next tile's first K-tile (HVLM_PHASE_SYNC_K relies on it) are waited for together
  // (nothing in the language enforces the invariant.  Development builds carry a flag that staged_landed() raises and the
  //  peeled first K-tile consumes: an epilogue path that forgets the wait traps on the first tile it hands over instead of reading LDS
  //  pieces that may not have landed (HVLM_LIB=lib/libhvlm_hip_dev.so python -m pytest tests -m gpu -k 'gemm or clip or layernorm_fold'
  //  runs every epilogue variant through it: profiles/r05j_dev_invariant.txt).)
#ifdef HVLM_DEV
  bool dev_landed = true;                    // (the prologue's HVLM_VMCNT(0) covers the first tile)
#endif
  auto staged_landed = [&]() {
    if constexpr (M16) __builtin_amdgcn_s_waitcnt(0x0F70);
#ifdef HVLM_DEV
    dev_landed = true;
#endif
  };
  auto epilogue = [&](int m0, int n0, size_t c_slab, int i_lo, int i_hi) {   // row fragments i_lo .. i_hi - 1 (a half-tile unit: 2 of 4)
    if (!fast_ok || n0 + BN > p.N) {
      staged_landed();
      epilogue_generic(m0, n0, c_slab, i_lo, i_hi);
      __builtin_amdgcn_s_waitcnt(0x0F70);   // tracked stores must not be pending at the top of the K loop (see above)
      return;
    }
    if constexpr (EPI_SWAP) {
      staged_landed();
      epilogue_swap(m0, n0, i_lo, i_hi);
      return;
    } else {
    const int ncol0 = n0 + wn * 64 + 8 * t_c;
    if (!p.res) {
      // no residual: bias / activation in the accumulator layout, bf16 through LDS.  Two 2 KiB buffers (64-B rows, the
      // 16-B column pair p of row r stored at pair position p ^ ((r>>1)&3): 2-way conflicts on the 8-B writes, none on
      // the 16-B reads of the transposed layout).
      u32x2 bias_q[2][4];
      if (p.bias) {
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
          for (int g = 0; g < 4; ++g) bias_q[j][g] = *reinterpret_cast<const u32x2*>(p.bias + col_of(n0, j, g));
      }
      // software-pipelined over the 8 fragments f = (i, j):  write(f) ; store(f-1) ; read(f), alternating the two
      // buffers: write(f) never targets the buffer read(f-1) may still be reading (issuing a write right behind an
      // unfinished read of the same buffer produced wrong elements), and write(f+2) is ordered after read(f) through
      // the registers store(f) consumed.
      float osc[4][2];                       // e4m3 output: out_inv of the eight rows this lane stores (fetching them together
      if constexpr (FP8) {                   // with the row / column scales, ahead of the de-scale block, measured no faster and
        if (p.out_inv) {                     // doubled the scratch of the kernel)
#pragma unroll
          for (int ip = 0; ip < 4; ++ip)
#pragma unroll
            for (int q = 0; q < 2; ++q)
              osc[ip][q] = p.out_inv[min(m0 + (ip >> 1) * 128 + grp * 64 + (ip & 1) * 32 + t_row + 16 * q, p.M - 1)];
        }
      }
      // LayerNorm fold (consumer side): the A rows are the RAW residual stream and B = W (.) ln_w, so that
      //   LN(x) W^T + b = rstd * (x B^T - mean * c1) + (b + W ln_b),  c1[n] = sum_k B[n, k]
      // — the normalised operand is never written or read (CLIPEncoderLayer.layer_norm1 -> q|k|v, layer_norm2 -> fc1).  Per lane:
      // (mean, rstd) of its 8 rows and c1 of its 16 columns, loaded once per tile.
      float lmu[4][2], lrs[4][2], c1f[2][2][4];
      bool lnf = false;
      if constexpr (LNF) {
        lnf = p.ln_stats != nullptr;
        if (lnf) {
#pragma unroll
          for (int ii = 0; ii < 4; ++ii)
#pragma unroll
            for (int h = 0; h < 2; ++h) {
              const int row = min(m0 + (ii >> 1) * 128 + grp * 64 + (ii & 1) * 32 + h * 16 + l15, p.M - 1);
              const f32x2 mr = *reinterpret_cast<const f32x2*>(p.ln_stats + 2 * (size_t)row);
              lmu[ii][h] = mr[0]; lrs[ii][h] = mr[1];
            }
#pragma unroll
          for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int hc = 0; hc < 2; ++hc) {
              const f32x4 cc = *reinterpret_cast<const f32x4*>(p.ln_c1 + col_of(n0, j, 2 * hc));
              c1f[j][hc][0] = cc[0]; c1f[j][hc][1] = cc[1]; c1f[j][hc][2] = cc[2]; c1f[j][hc][3] = cc[3];
            }
        }
      }
      staged_landed();
      // The loop below exists in a few SPECIALISATIONS of (bias, QuickGELU, LayerNorm fold): with the three as run-time flags the
      // compiler emitted a scalar branch around every 4-element group, four v_cndmask per group for the fold select, and ran the
      // exp -> add -> rcp chain of one element at a time behind s_nop hazards.  (1 / 0 = known at compile time, -1 = run time.)
      const int f_lo = 2 * i_lo, f_hi = 2 * i_hi;       // fragments of this unit (a half-tile unit: 4 of the 8)
      auto run_nores = [&](auto hb_t, auto act_t, auto lnf_t) {
      constexpr int HBc = decltype(hb_t)::value, ACc = decltype(act_t)::value, LNc = decltype(lnf_t)::value;
      const bool hb = HBc >= 0 ? (HBc == 1) : (p.bias != nullptr);
      const bool gelu = ACc >= 0 ? (ACc == 1) : (p.act == HVLM_ACT_QUICK_GELU);
      const bool lnf_on = LNc >= 0 ? (LNc == 1) : lnf;
      u32x4 o_prev[2];
#pragma unroll
      for (int f = 0; f <= 8; ++f) {
        const int i = f >> 1, j = f & 1;
        if (f < f_lo || f > f_hi) continue;
        if (f < f_hi) {
#pragma unroll
          for (int g = 0; g < 4; ++g) {
            float v[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = acc[j][i][4 * g + e];
            if constexpr (LNF) {
              if (lnf_on) {
                const float nmu = -lmu[i][g & 1], rs = lrs[i][g & 1];
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = __builtin_fmaf(nmu, c1f[j][g >> 1][e], v[e]) * rs;
              }
            }
            if (hb) {
              const u32x2 bb = bias_q[j][g];
              v[0] += EL::lo(bb[0]); v[1] += EL::hi(bb[0]); v[2] += EL::lo(bb[1]); v[3] += EL::hi(bb[1]);
            }
            if (gelu) {                      // x * sigmoid(1.702 x), the four elements' chains interleaved stage by stage
              float z[4];
#pragma unroll
              for (int e = 0; e < 4; ++e) z[e] = __builtin_amdgcn_exp2f(-1.702f * 1.4426950408889634f * v[e]);
#pragma unroll
              for (int e = 0; e < 4; ++e) z[e] = __builtin_amdgcn_rcpf(1.f + z[e]);
#pragma unroll
              for (int e = 0; e < 4; ++e) v[e] *= z[e];
            }
            const u32x2 o = {EL::pack2(v[0], v[1]), EL::pack2(v[2], v[3])};
            const int qr = qrow(g), qc = qcol(g);   // row qr, bf16 columns qc..qc+3 of the 32x32 fragment (64-B rows)
            *reinterpret_cast<u32x2*>(epi + (f & 1) * 2048 + qr * 64 + (((qc >> 3) ^ ((qr >> 1) & 3)) << 4) + ((qc >> 2) & 1) * 8) = o;
          }
        }
        if (f > f_lo) {
          const int ip = (f - 1) >> 1, jp = (f - 1) & 1;
          const int mrow0 = m0 + (ip >> 1) * 128 + grp * 64 + (ip & 1) * 32 + t_row;
#pragma unroll
          for (int q = 0; q < 2; ++q) {
            const int m = mrow0 + 16 * q;
            bool done8 = false;
            if constexpr (FP8) {
              if (p.out_inv) {
                if (m < p.M)
                  store8_untracked(reinterpret_cast<unsigned char*>(p.C) + (size_t)m * p.ldc + ncol0 + jp * 32,
                                   bf16x8_to_e4m3(o_prev[q], __builtin_amdgcn_rcpf(osc[ip][q])));
                done8 = true;
              }
            }
            if (!done8 && m < p.M) store16_untracked(reinterpret_cast<bf16_t*>(p.C) + (size_t)m * p.ldc + ncol0 + jp * 32, o_prev[q]);
          }
        }
        if (f < f_hi) {
#pragma unroll
          for (int q = 0; q < 2; ++q) {
            const int r = t_row + 16 * q;
            o_prev[q] = *reinterpret_cast<const u32x4*>(epi + (f & 1) * 2048 + r * 64 + ((t_c ^ ((r >> 1) & 3)) << 4));
          }
        }
      }
      };   // run_nores
      using TM = std::integral_constant<int, -1>;
      const bool hb_r = p.bias != nullptr, gelu_r = p.act == HVLM_ACT_QUICK_GELU, plain_r = p.act == HVLM_ACT_NONE;
      if constexpr (LNF) {                   // the CLIP tower's fp16 kernel: q|k|v, fc1 (folded and explicit-LayerNorm towers)
        if (hb_r && lnf && gelu_r) run_nores(T1{}, T1{}, T1{});
        else if (hb_r && lnf && plain_r) run_nores(T1{}, T0{}, T1{});
        else run_nores(TM{}, TM{}, TM{});
      } else if constexpr (FP8) {            // e4m3: the decoder prefill (no bias) and the e4m3 CLIP tower (bias, bias + QuickGELU)
        if (!hb_r && plain_r) run_nores(T0{}, T0{}, T0{});
        else if (hb_r && gelu_r) run_nores(T1{}, T1{}, T0{});
        else run_nores(TM{}, TM{}, T0{});
      } else {                               // bf16: the decoder's GEMMs carry neither
        if (!hb_r && plain_r) run_nores(T0{}, T0{}, T0{});
        else run_nores(TM{}, TM{}, T0{});
      }
      return;
    }
    float bias_f[2][8];
    if (p.bias) {
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const u32x4 bb = *reinterpret_cast<const u32x4*>(p.bias + ncol0 + j * 32);
#pragma unroll
        for (int e = 0; e < 4; ++e) { bias_f[j][2 * e] = EL::lo(bb[e]); bias_f[j][2 * e + 1] = EL::hi(bb[e]); }
      }
    }
    // residual fragments are fetched ONE ROW-FRAGMENT AHEAD: the loads of fragment i+1 are in flight during the LDS round
    // trips and stores of fragment i (issued at the top of iteration i, their ~1 us HBM latency was exposed four times per tile)
    u32x4 res_n[2][2];
    auto load_res = [&](int i, u32x4 (&dst)[2][2]) {
      const int mr = m0 + (i >> 1) * 128 + grp * 64 + (i & 1) * 32 + t_row;
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int q = 0; q < 2; ++q) {
          const int mc = min(mr + 16 * q, p.M - 1);
          dst[j][q] = *reinterpret_cast<const u32x4*>(p.res + (size_t)mc * p.ldr + ncol0 + j * 32);
        }
    };
    // (specialised like the no-residual loop above: bias / activation / statistics known at compile time, or all at run time)
    auto run_res = [&](auto hb_t, auto st_t) {
    constexpr int HBc = decltype(hb_t)::value, STc = decltype(st_t)::value;
    const bool hb = HBc >= 0 ? (HBc == 1) : (p.bias != nullptr);
    const bool gelu = HBc >= 0 ? false : (p.act == HVLM_ACT_QUICK_GELU);      // the specialised forms are dispatched for act == NONE only
    load_res(i_lo, res_n);
    staged_landed();
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      if (i < i_lo || i >= i_hi) continue;
      const int mrow0 = m0 + (i >> 1) * 128 + grp * 64 + (i & 1) * 32 + t_row;   // + 16*q
      u32x4 res_r[2][2];          // p.res != 0 on this path: loads and uses stay unconditional (see store16_untracked)
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int q = 0; q < 2; ++q) res_r[j][q] = res_n[j][q];
      if (i + 1 < i_hi) load_res(i + 1, res_n);
      float st1[2] = {0.f, 0.f}, st2[2] = {0.f, 0.f};      // LayerNorm fold (producer side): sum / sum of squares of this lane's outputs
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        // accumulator layout -> LDS: row l31, 16-B chunk (2g + hi) at chunk position (2g + hi) ^ (l31 & 7)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const f32x4 o = {acc[j][i][4 * g], acc[j][i][4 * g + 1], acc[j][i][4 * g + 2], acc[j][i][4 * g + 3]};
          const int qr = qrow(g), qc = qcol(g);     // row qr, fp32 columns qc..qc+3: 16-B chunk qc/4 of a 128-B row
          *reinterpret_cast<f32x4*>(epi + qr * 128 + ((((qc >> 2)) ^ (qr & 7)) << 4)) = o;
        }
#pragma unroll
        for (int q = 0; q < 2; ++q) {
          const int r = t_row + 16 * q;
          const f32x4 lo = *reinterpret_cast<const f32x4*>(epi + r * 128 + (((2 * t_c) ^ (r & 7)) << 4));
          const f32x4 hi4 = *reinterpret_cast<const f32x4*>(epi + r * 128 + (((2 * t_c + 1) ^ (r & 7)) << 4));
          float v[8] = {lo[0], lo[1], lo[2], lo[3], hi4[0], hi4[1], hi4[2], hi4[3]};
          if (hb) {
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] += bias_f[j][e];
          }
          if (gelu) {
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] = quick_gelu_pp(v[e]);
          }
          {
            u32x4 rr = res_r[j][q];
            asm volatile("" : "+v"(rr));   // consume the load outside the `m < M` branch: nothing stays pending on any path
#pragma unroll
            for (int e = 0; e < 4; ++e) { v[2 * e] += EL::lo(rr[e]); v[2 * e + 1] += EL::hi(rr[e]); }
          }
          const int m = mrow0 + 16 * q;
          const u32x4 o = {EL::pack2(v[0], v[1]), EL::pack2(v[2], v[3]), EL::pack2(v[4], v[5]), EL::pack2(v[6], v[7])};
          if (m < p.M) store16_untracked(reinterpret_cast<bf16_t*>(p.C) + (size_t)m * p.ldc + ncol0 + j * 32, o);
          if constexpr (LNF) {
            if (STc >= 0 ? (STc == 1) : (p.stats_out != nullptr)) {                 // statistics of the values AS STORED (rounded to 16 bits): what the consumer reads
#pragma unroll
              for (int e = 0; e < 4; ++e) {
                const float a = EL::lo(o[e]), b = EL::hi(o[e]);
                st1[q] += a + b;
                st2[q] = __builtin_fmaf(a, a, st2[q]);
                st2[q] = __builtin_fmaf(b, b, st2[q]);
              }
            }
          }
        }
      }
      if constexpr (LNF) {
        if (STc >= 0 ? (STc == 1) : (p.stats_out != nullptr)) {
          // the four lanes t_c = 0..3 of a row hold its 4 x 8 columns per j: quad reduction, lane 0 of the quad writes the wave's
          // partial for (row, tile column, wave column block) — one writer per entry, fixed order: deterministic
#pragma unroll
          for (int q = 0; q < 2; ++q) {
            float s1 = st1[q], s2 = st2[q];
            s1 += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(s1), 0xB1, 0xF, 0xF, true));
            s2 += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(s2), 0xB1, 0xF, 0xF, true));
            s1 += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(s1), 0x4E, 0xF, 0xF, true));
            s2 += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(s2), 0x4E, 0xF, 0xF, true));
            const int m = mrow0 + 16 * q;
            if (t_c == 0 && m < p.M) {
              const size_t part = (size_t)(n0 / BN) * 4 + wn;
              const u32x2 sv = {__float_as_uint(s1), __float_as_uint(s2)};
              store8_untracked(p.stats_out + ((size_t)m * (p.tiles_n * 4) + part) * 2, sv);
            }
          }
        }
      }
    }
    };   // run_res
    // (ONE instantiation, everything tested at run time: a second, specialised copy of this loop — bias + statistics known at compile
    // time — pushed the fp16 kernel to 108 bytes of scratch with reloads all over the epilogue, each behind a vmcnt wait that also
    // drains the next tile's LDS-DMA prefetch; measured on the no-residual loop the specialisation is worth 2-3 %, here it is not
    // worth the spills)
    run_res(std::integral_constant<int, -1>{}, std::integral_constant<int, -1>{});
    }   // !EPI_SWAP
  };

  // ---- prologue: K-tile 0 (4 half-tiles) and A0,B0 of K-tile 1 in flight -------------------------
  if (dyn) {
    // wave 0 draws the first three units; the first two are needed by everyone right away (through LDS, free now)
    int* lq = reinterpret_cast<int*>(smem);
    if (w == 0 && l == 0) {
      const int u0 = fetch_unit(), u1 = fetch_unit(), u2 = fetch_unit();
      lq[0] = u0; lq[1] = u1;
      g_pp.slots[blockIdx.x][2] = u2;
    }
    __syncthreads();
    t_cur = __builtin_amdgcn_readfirstlane(lq[0]);
    t_nxt = __builtin_amdgcn_readfirstlane(lq[1]);
    __syncthreads();
    if (t_cur < 0) {                      // more workgroups than units (cannot happen with grid <= units; kept for safety)
      if (w == 0 && l == 0 && __hip_atomic_fetch_add(&g_pp.done[0], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == gridDim.x - 1) {
        for (int x = 0; x < 8; ++x) g_pp.ctr[x][0] = 0;
        g_pp.done[0] = 0;
      }
      return;
    }
  } else {
    t_cur = swz_of((int)blockIdx.x);
    t_nxt = -1;
  }
  Cursor c1, c2;
  c1.vb = dyn ? 0 : (int)blockIdx.x; c1.t = t_cur; c1.kt = 0; c1.koff = 0;
  make_desc(c1);
  stage_a(0, 0, c1); stage_b(0, 0, c1); stage_b(0, 1, c1); stage_a(0, 1, c1);
  advance(c1);                            // c1 = K-tile 1 (its B1, A1 are staged in phases 0, 1 below)
  stage_a(1, 0, c1); stage_b(1, 0, c1);
  if constexpr (SEG4) stage_b(1, 1, c1);  // the 4-segment schedule stages A0, B0, B1 two K-tiles ahead, A1 one ahead
  c2 = c1;
  advance(c2);                            // c2 = K-tile 2
  if constexpr (M16) HVLM_VMCNT(0);       // PEEL (below): the first K-tile of a tile does not wait for loads — everything staged so far has landed
  else HVLM_VMCNT(8);                     // A0, B0 of K-tile 0 landed (this wave's pieces)
  __builtin_amdgcn_s_barrier();

  int vb = blockIdx.x;                    // static mode: launch-order index of the tile being computed
  unsigned g = 0;                         // running K-tile counter; K-tile g lives in LDS stage g & 1

  // W0..W3 (TN only): lgkmcnt that retires the transpose reads feeding k-step 0..3 of this phase (reads are issued
  // k-step major); the wait asm takes the fragments as operands so the MFMAs cannot be scheduled above it.
#define HVLM_MFMA_STEP(J, I0, BF, AF0, AF1, KS, W)                                                        \
  do {                                                                                                    \
    if constexpr (KS < NKS) {                                                                             \
      if constexpr (TN) asm volatile("s_waitcnt lgkmcnt(" #W ")" : "+v"(BF[KS]), "+v"(AF0[KS]), "+v"(AF1[KS])); \
      if constexpr (FP8) {                                                                                \
        acc[J][I0] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(BF[KS], AF0[KS], acc[J][I0], 0, 0, 0, SCALE1, 0, SCALE1); \
        acc[J][I0 + 1] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(BF[KS], AF1[KS], acc[J][I0 + 1], 0, 0, 0, SCALE1, 0, SCALE1); \
      } else {                                                                                            \
        acc[J][I0] = EL::mfma32(BF[KS], AF0[KS], acc[J][I0]);       \
        acc[J][I0 + 1] = EL::mfma32(BF[KS], AF1[KS], acc[J][I0 + 1]); \
      }                                                                                                   \
    }                                                                                                     \
  } while (0)
#define HVLM_MFMA_QUAD(J, I0, BF, AF0, AF1, W0, W1, W2, W3)                                               \
  do {                                                                                                    \
    __builtin_amdgcn_sched_barrier(0);                                                                    \
    asm volatile("" : "+v"(acc[J][I0]), "+v"(acc[J][I0 + 1]));  /* pin: MFMAs stay inside this phase */    \
    __builtin_amdgcn_s_setprio(1);                                                                        \
    HVLM_MFMA_STEP(J, I0, BF, AF0, AF1, 0, W0);                                                           \
    HVLM_MFMA_STEP(J, I0, BF, AF0, AF1, 1, W1);                                                           \
    HVLM_MFMA_STEP(J, I0, BF, AF0, AF1, 2, W2);                                                           \
    HVLM_MFMA_STEP(J, I0, BF, AF0, AF1, 3, W3);                                                           \
    __builtin_amdgcn_s_setprio(0);                                                                        \
    asm volatile("" : "+v"(acc[J][I0]), "+v"(acc[J][I0 + 1]));                                             \
    __builtin_amdgcn_sched_barrier(0);                                                                    \
  } while (0)
#define HVLM_MFMA_OCT_STEP(I0, AF0, AF1, KS)                                                              \
  do {                                                                                                    \
    if constexpr (KS < NKS) {                                                                             \
      if constexpr (FP8) {                                                                                \
        acc[0][I0] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(fb0[KS], AF0[KS], acc[0][I0], 0, 0, 0, SCALE1, 0, SCALE1); \
        acc[0][I0 + 1] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(fb0[KS], AF1[KS], acc[0][I0 + 1], 0, 0, 0, SCALE1, 0, SCALE1); \
        acc[1][I0] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(fb1[KS], AF0[KS], acc[1][I0], 0, 0, 0, SCALE1, 0, SCALE1); \
        acc[1][I0 + 1] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(fb1[KS], AF1[KS], acc[1][I0 + 1], 0, 0, 0, SCALE1, 0, SCALE1); \
      } else {                                                                                            \
        acc[0][I0] = EL::mfma32(fb0[KS], AF0[KS], acc[0][I0]);      \
        acc[0][I0 + 1] = EL::mfma32(fb0[KS], AF1[KS], acc[0][I0 + 1]); \
        acc[1][I0] = EL::mfma32(fb1[KS], AF0[KS], acc[1][I0]);      \
        acc[1][I0 + 1] = EL::mfma32(fb1[KS], AF1[KS], acc[1][I0 + 1]); \
      }                                                                                                   \
    }                                                                                                     \
  } while (0)
#define HVLM_MFMA_OCT(I0, AF0, AF1)                                                                       \
  do {                                                                                                    \
    __builtin_amdgcn_sched_barrier(0);                                                                    \
    asm volatile("" : "+v"(acc[0][I0]), "+v"(acc[0][I0 + 1]), "+v"(acc[1][I0]), "+v"(acc[1][I0 + 1]));      \
    __builtin_amdgcn_s_setprio(1);                                                                        \
    HVLM_MFMA_OCT_STEP(I0, AF0, AF1, 0);                                                                  \
    HVLM_MFMA_OCT_STEP(I0, AF0, AF1, 1);                                                                  \
    HVLM_MFMA_OCT_STEP(I0, AF0, AF1, 2);                                                                  \
    HVLM_MFMA_OCT_STEP(I0, AF0, AF1, 3);                                                                  \
    __builtin_amdgcn_s_setprio(0);                                                                        \
    asm volatile("" : "+v"(acc[0][I0]), "+v"(acc[0][I0 + 1]), "+v"(acc[1][I0]), "+v"(acc[1][I0 + 1]));      \
    __builtin_amdgcn_sched_barrier(0);                                                                    \
  } while (0)
  // 16x16x32 form of the 16-MFMA segment: 4 (j, i) fragments x 4 quadrants x 2 k-halves = 32 MFMAs of 16 cycles.  A
  // quadrant t = 2*sb + sa of acc[j][i] (registers 4t..4t+3) += W-block sb (16 output columns, MFMA A) x X-block sa (16 rows, B).
  constexpr bool PEEL = M16;               // first K-tile peeled, C = 0 by inline constant (see the tile loop)
  constexpr int NKH16 = FP8 ? 1 : 2;      // MFMA k-steps per K-tile in the 16x16 form (fp8: one 16x16x128; 16-bit: two 16x16x32)
  // `zero`: the first k-step of a tile takes C = 0 as the MFMA's inline constant instead of reading a zeroed accumulator (PEEL below)
  auto mfma16 = [&](f32x16& c, int t, const frag_t& wf, const frag_t& xf, bool zero) {
    if constexpr (M16) {
      f32x4 q = {c[4 * t], c[4 * t + 1], c[4 * t + 2], c[4 * t + 3]};
      if (zero) q = f32x4{0.f, 0.f, 0.f, 0.f};
      if constexpr (FP8) q = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(wf, xf, q, 0, 0, 0, 0x7f7f7f7f, 0, 0x7f7f7f7f);
      else if constexpr (F16) q = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, wf), __builtin_bit_cast(f16x8, xf), q, 0, 0, 0);
      else q = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf, xf, q, 0, 0, 0);
      c[4 * t] = q[0]; c[4 * t + 1] = q[1]; c[4 * t + 2] = q[2]; c[4 * t + 3] = q[3];
    }
  };
#define HVLM_MFMA_OCT16(I0, AF0, AF1)                                                                     \
  do {                                                                                                    \
    __builtin_amdgcn_sched_barrier(0);                                                                    \
    if constexpr (!FIRSTK) asm volatile("" : "+v"(acc[0][I0]), "+v"(acc[0][I0 + 1]), "+v"(acc[1][I0]), "+v"(acc[1][I0 + 1])); \
    __builtin_amdgcn_s_setprio(1);                                                                        \
    _Pragma("unroll") for (int ks2_ = 0; ks2_ < NKH16; ++ks2_) {                                          \
      _Pragma("unroll") for (int t_ = 0; t_ < 4; ++t_) {                                                  \
        const int sb_ = t_ >> 1, sa_ = t_ & 1;                                                            \
        mfma16(acc[0][I0], t_, fb0[sb_ * NKH16 + ks2_], AF0[sa_ * NKH16 + ks2_], FIRSTK && ks2_ == 0);     \
        mfma16(acc[0][I0 + 1], t_, fb0[sb_ * NKH16 + ks2_], AF1[sa_ * NKH16 + ks2_], FIRSTK && ks2_ == 0); \
        mfma16(acc[1][I0], t_, fb1[sb_ * NKH16 + ks2_], AF0[sa_ * NKH16 + ks2_], FIRSTK && ks2_ == 0);     \
        mfma16(acc[1][I0 + 1], t_, fb1[sb_ * NKH16 + ks2_], AF1[sa_ * NKH16 + ks2_], FIRSTK && ks2_ == 0); \
      }                                                                                                   \
    }                                                                                                     \
    __builtin_amdgcn_s_setprio(0);                                                                        \
    asm volatile("" : "+v"(acc[0][I0]), "+v"(acc[0][I0 + 1]), "+v"(acc[1][I0]), "+v"(acc[1][I0 + 1]));      \
    __builtin_amdgcn_sched_barrier(0);                                                                    \
  } while (0)
#define HVLM_PHASE_SYNC()                 \
  do {                                    \
    __builtin_amdgcn_sched_barrier(0);    \
    HVLM_VMCNT(8);                        \
    __builtin_amdgcn_s_barrier();         \
  } while (0)
// the same inside the 4-segment K-tile: the peeled first K-tile of a tile (FIRSTK) reads only pieces that were staged BEFORE the previous
// tile's epilogue (K-tile 0, and A0 / B0 / B1 of K-tile 1), and that epilogue starts with a full vmcnt(0) — so it needs the barrier but
// no wait.  With the counted wait there, the first syncs of a tile also waited for the previous tile's 16 stores per wave to be
// acknowledged (they count in vmcnt and sit between the staged loads and the new ones): first K-tile 3.8-4.4k cycles against 2.4k.
#define HVLM_PHASE_SYNC_K()               \
  do {                                    \
    __builtin_amdgcn_sched_barrier(0);    \
    if constexpr (!FIRSTK) { if (DBG && (p.dbg & 16)) HVLM_VMCNT(12); else HVLM_VMCNT(8); } \
    __builtin_amdgcn_s_barrier();         \
  } while (0)

  while (true) {
    HVLM_PP_STAMP(0);
    // PEEL (the 16x16 kernels): no zeroing pass over the 128 accumulator registers at the top of a tile (128 v_mov per wave, two waves
    // per SIMD: ~1000 cycles before the first MFMA of the tile, tools/gemm_tile_timeline.py) — the first K-tile is a copy of the loop
    // body whose first k-step passes C = 0 to the MFMA as an inline constant; same sums, bit for bit
    if constexpr (!PEEL) {
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int r = 0; r < 16; ++r) acc[j][i][r] = 0.f;
    }
    if (grp == 1) __builtin_amdgcn_s_barrier();          // group 1 runs one barrier behind group 0
    // hm = 0: the whole tile; 1 / 2: a half-tile unit — only MFMA segment A (rows 0-127) / B (rows 128-255) and its fragment
    // reads are issued, everything else (staging, barriers, cursors) is the same code.  Three copies of the loop, so that the
    // whole-tile loop is compiled exactly as before.
    // Next-unit bookkeeping of the tile, issued at the top of its LAST K-tile and consumed right behind the K loop (a K-tile of latency
    // for both round trips): wave 0's ticket for the unit three tiles ahead (fetch_issue) and everyone's load of the unit two tiles ahead
    // from the workgroup's ring.  Round 4 (tools/gemm_tile_timeline.py): consumed behind the epilogue, their wait was a vmcnt(0) that
    // also waited for the tile's 16 stores per wave to be acknowledged — ~800 cycles per tile on every wave.
    int m0 = 0, n0 = 0, ksl = 0, t_aft_v = -1;
    unsigned u_ticket = 0;
    auto ktile = [&](auto hm_tag, auto first_tag, const int kt) {
    constexpr int HM = decltype(hm_tag)::value;
    constexpr bool FIRSTK = decltype(first_tag)::value != 0;
    {
      if (kt + 1 == nk) {
        tile_coords(t_cur, m0, n0, ksl);
        if (dyn) {
          if (w == 0 && l == 0) u_ticket = fetch_issue();
          t_aft_v = __hip_atomic_load(&g_pp.slots[blockIdx.x][(it + 2) & 3], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
      }
      const unsigned cb = g & 1u, ob = cb ^ 1u;
      const unsigned sb = cb * STAGE_BYTES;
      frag_t fb0[NKS], fb1[NKS], fa0[NKS], fa1[NKS];
      constexpr int SCALE1 = 0x7f7f7f7f;        // e8m0 block scales = 2^0
      if constexpr (SEG4) {
        // 4-segment schedule: two 16-MFMA segments per K-tile (4 barriers instead of 8).  Load segment A reads B0, B1
        // and this group's A rows of half 0 (16 reads) and stages A1 of K-tile g+1; load segment B reads the rows of
        // half 1 (8 reads) and stages A0, B0, B1 of K-tile g+2 into the slots segment A just read.  Every DMA piece
        // has four segments to land; the counted wait stays vmcnt(8) at both syncs.
        read_b(sb, T0{}, fb0);
        if constexpr (HM != 2) read_a(sb, T0{}, fa0);
        __builtin_amdgcn_sched_barrier(0);
        read_b(sb, T1{}, fb1);
        if constexpr (HM != 2) read_a(sb, T1{}, fa1);
        if (!(DBG && (p.dbg & 2))) stage_a(ob, 1, c1);
        HVLM_PHASE_SYNC_K();
        if constexpr (TN) {   // the transpose reads are inline asm (untracked): retire them here, fragments as operands
          asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(fb0[0]), "+v"(fb0[1]), "+v"(fb0[2]), "+v"(fb0[3]), "+v"(fb1[0]), "+v"(fb1[1]),
                       "+v"(fb1[2]), "+v"(fb1[3]), "+v"(fa0[0]), "+v"(fa0[1]), "+v"(fa0[2]), "+v"(fa0[3]), "+v"(fa1[0]), "+v"(fa1[1]),
                       "+v"(fa1[2]), "+v"(fa1[3]));
        }
        if constexpr (HM != 2) { if constexpr (M16) HVLM_MFMA_OCT16(0, fa0, fa1); else HVLM_MFMA_OCT(0, fa0, fa1); }
        __builtin_amdgcn_s_barrier();
        if (HM != 1 && !(DBG && (p.dbg & 8))) {           // dbg 8: a third of the fragment reads gone (stale A fragments of half 0)
          read_a(sb, T2{}, fa0);
          read_a(sb, T3{}, fa1);
        }
        if (!(DBG && (p.dbg & 1))) {
          stage_a(cb, 0, c2);
          stage_b(cb, 0, c2);
          stage_b(cb, 1, c2);
          // dbg 16 (round 6: what 1.5 x the operand bytes per flop cost — the L2 -> LDS stream of a 256 x 128 re-tiling): the B
          // half-tiles are staged TWICE (same slots, same bytes: 12 instead of 8 LDS-DMA pieces per K-tile and wave, the extra ones L2 hits like the
          // extra bytes of a smaller tile would be); the counted wait of both syncs becomes vmcnt(12) so that every piece keeps its four segments
          // to land.  dbg 32 selects this kernel with nothing changed (the baseline of the A/B: the 32x32x16 form of the production schedule)
          if (DBG && (p.dbg & 16)) {
            stage_b(cb, 0, c2);
            stage_b(cb, 1, c2);
          }
        }
        HVLM_PHASE_SYNC_K();
        if constexpr (TN) {
          asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(fa0[0]), "+v"(fa0[1]), "+v"(fa0[2]), "+v"(fa0[3]), "+v"(fa1[0]), "+v"(fa1[1]),
                       "+v"(fa1[2]), "+v"(fa1[3]));
        }
        if constexpr (HM != 1) { if constexpr (M16) HVLM_MFMA_OCT16(2, fa0, fa1); else HVLM_MFMA_OCT(2, fa0, fa1); }
      } else {
      // phase 0: quadrant (rows 0-63, cols 0-31) — reads A0, B0; stages B1 of K-tile g+1
      const unsigned adb = lds0 + sb + tb_off, ada0 = lds0 + sb + ta_off[0], ada1 = lds0 + sb + ta_off[1];   // TN
      if constexpr (!TN) {
        read_b(sb, T0{}, fb0);
        __builtin_amdgcn_sched_barrier(0);
        read_a(sb, T0{}, fa0);
        read_a(sb, T1{}, fa1);
      } else {
        HVLM_TR2(fb0[0], adb, 0 * 4096); HVLM_TR2(fa0[0], ada0, 0 * 4096); HVLM_TR2(fa1[0], ada1, 0 * 4096);
        HVLM_TR2(fb0[1], adb, 1 * 4096); HVLM_TR2(fa0[1], ada0, 1 * 4096); HVLM_TR2(fa1[1], ada1, 1 * 4096);
        HVLM_TR2(fb0[2], adb, 2 * 4096); HVLM_TR2(fa0[2], ada0, 2 * 4096); HVLM_TR2(fa1[2], ada1, 2 * 4096);
        HVLM_TR2(fb0[3], adb, 3 * 4096); HVLM_TR2(fa0[3], ada0, 3 * 4096); HVLM_TR2(fa1[3], ada1, 3 * 4096);
      }
      stage_b(ob, 1, c1);
      HVLM_PHASE_SYNC();
      HVLM_MFMA_QUAD(0, 0, fb0, fa0, fa1, 15, 12, 6, 0);
      __builtin_amdgcn_s_barrier();
      // phase 1: quadrant (rows 0-63, cols 128-159) — reads B1; stages A1 of K-tile g+1
      read_b(sb, T1{}, fb1);
      stage_a(ob, 1, c1);
      HVLM_PHASE_SYNC();
      HVLM_MFMA_QUAD(1, 0, fb1, fa0, fa1, 6, 4, 2, 0);
      __builtin_amdgcn_s_barrier();
      // phase 2: quadrant (rows 128-191, cols 128-159) — reads A1; stages A0 of K-tile g+2 (slot read in phase 0)
      if constexpr (!TN) {
        read_a(sb, T2{}, fa0);
        read_a(sb, T3{}, fa1);
      } else {
        HVLM_TR2(fa0[0], ada0, 16384 + 0 * 4096); HVLM_TR2(fa1[0], ada1, 16384 + 0 * 4096);
        HVLM_TR2(fa0[1], ada0, 16384 + 1 * 4096); HVLM_TR2(fa1[1], ada1, 16384 + 1 * 4096);
        HVLM_TR2(fa0[2], ada0, 16384 + 2 * 4096); HVLM_TR2(fa1[2], ada1, 16384 + 2 * 4096);
        HVLM_TR2(fa0[3], ada0, 16384 + 3 * 4096); HVLM_TR2(fa1[3], ada1, 16384 + 3 * 4096);
      }
      stage_a(cb, 0, c2);
      HVLM_PHASE_SYNC();
      HVLM_MFMA_QUAD(1, 2, fb1, fa0, fa1, 12, 8, 4, 0);
      __builtin_amdgcn_s_barrier();
      // phase 3: quadrant (rows 128-191, cols 0-31) — B0 still in registers; stages B0 of K-tile g+2
      stage_b(cb, 0, c2);
      HVLM_PHASE_SYNC();
      HVLM_MFMA_QUAD(0, 2, fb0, fa0, fa1, 15, 15, 15, 15);
      }
      if (!(DBG && (p.dbg & 4))) {
        c1 = c2;
        advance(c2);
      }
      if (!(grp == 1 && kt + 1 == nk)) __builtin_amdgcn_s_barrier();   // group 1 re-aligns at the epilogue
#ifdef HVLM_GEMM_PROF
      if (kt == 0) HVLM_PP_STAMP(5);
#endif
    }
    };   // ktile
    auto run_k = [&](auto hm_tag) {
      if constexpr (PEEL) {
#ifdef HVLM_DEV
        if (!dev_landed) __builtin_trap();   // an epilogue path returned without staged_landed(): see there
        dev_landed = false;
#endif
        ktile(hm_tag, T1{}, 0);
        ++g;
        for (int kt = 1; kt < nk; ++kt, ++g) ktile(hm_tag, T0{}, kt);
      } else {
        for (int kt = 0; kt < nk; ++kt, ++g) ktile(hm_tag, T0{}, kt);
      }
    };
    const int half = HALF_OK ? (t_cur >> 28) : 0;          // (wave-uniform: t_cur is)
    if constexpr (HALF_OK) {
      if (half == 0) run_k(T0{}); else if (half == 1) run_k(T1{}); else run_k(T2{});
    } else {
      run_k(T0{});
    }
    HVLM_PP_STAMP(1);
    // dynamic mode: wave 0 parks the unit three tiles ahead in this workgroup's ring (an untracked store, see store16_untracked);
    // every wave picked up the unit two tiles ahead, parked one tile ago (L1-bypassing load)
    int t_aft = -1;
    if (dyn) {
      if (w == 0 && l == 0) {
        const int u_fetched = fetch_resolve(u_ticket);
        int* sp = &g_pp.slots[blockIdx.x][(it + 3) & 3];
        asm volatile("global_store_dword %0, %1, off" ::"v"(sp), "v"(u_fetched) : "memory");
      }
      t_aft = __builtin_amdgcn_readfirstlane(t_aft_v);
    }
    if constexpr (FP8) {
      // de-quantisation: tensor scale (alpha) x per-token scale of the row x per-output-channel scale of the column
      // (register quad g4 of fragment (j, i) holds row qrow(g4), columns qcol(g4).. — see the epilogue mapping above).
      // All scale loads are issued first — 4 (8 with 16-row fragments) row scales and 8 x 16 bytes of column scales per lane —
      // and the multiplies follow behind ONE memory latency.  The first version loaded inside the (j, g4, i) nest: 32 dword
      // column loads and 32 row loads per lane and tile, which made this block longer than the K loop of a K = 1024 tile
      // (o-projection of the CLIP tower: K loop 281 us, everything after it 384 us per launch).
      constexpr int RH = M16 ? 2 : 1;
      float rsv[4][RH];
      f32x4 csv[2][4];
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int h = 0; h < RH; ++h) {
          const int row = m0 + (i >> 1) * 128 + grp * 64 + (i & 1) * 32 + qrow(h);      // qrow(g) depends on g & 1 only (M16) or not at all
          rsv[i][h] = p.row_scale ? p.row_scale[min(row, p.M - 1)] : 1.f;
        }
      const bool cs_vec = p.col_scale && n0 + BN <= p.N && (((uintptr_t)p.col_scale) & 15) == 0;
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int g4 = 0; g4 < 4; ++g4) {
          const int nb = col_of(n0, j, g4);
          if (cs_vec) csv[j][g4] = *reinterpret_cast<const f32x4*>(p.col_scale + nb);
          else {
#pragma unroll
            for (int e = 0; e < 4; ++e) csv[j][g4][e] = p.col_scale ? p.col_scale[min(nb + e, p.N - 1)] : 1.f;
          }
        }
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int g4 = 0; g4 < 4; ++g4)
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            const float rs = p.row_scale ? p.alpha * rsv[i][M16 ? (g4 & 1) : 0] : p.alpha;
#pragma unroll
            for (int e = 0; e < 4; ++e) acc[j][i][4 * g4 + e] *= rs * csv[j][g4][e];
          }
    }
    HVLM_PP_STAMP(2);
    epilogue(m0, n0, (size_t)ksl * p.slab, half == 2 ? 2 : 0, half == 1 ? 2 : 4);
    HVLM_PP_STAMP(3);
    HVLM_PP_STAMP(4);
    ++it;
    if (dyn) {
      if (t_nxt < 0) break;
      t_cur = t_nxt; t_nxt = t_aft;
      if (c1.vb == 1) c1.vb = 0;          // the cursors that already crossed into the new current tile
      if (c2.vb == 1) c2.vb = 0;
    } else {
      vb += (int)gridDim.x;
      if (vb >= nwg) break;
      t_cur = swz_of(vb);
    }
  }
  if (dyn && w == 0 && l == 0) {
    // everyone's fetches precede their increment of `done`: the last one resets the counters for the next launch
    if (__hip_atomic_fetch_add(&g_pp.done[0], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == gridDim.x - 1) {
      for (int x = 0; x < 8; ++x) __hip_atomic_store(&g_pp.ctr[x][0], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_store(&g_pp.done[0], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
  HVLM_VMCNT(0);
#undef HVLM_MFMA_QUAD
#undef HVLM_MFMA_OCT
#undef HVLM_MFMA_OCT16
#undef HVLM_MFMA_OCT_STEP
#undef HVLM_MFMA_STEP
#undef HVLM_TR2
#undef HVLM_PHASE_SYNC
#undef HVLM_PHASE_SYNC_K
}

// Production kernels: the 4-segment schedule; 16-bit operands on 16x16x32 MFMAs (bf16, and the CLIP tower's IEEE-half twin),
// e4m3 operands on 32x32x64, K-strided (TN) operands on 32x32x16.
__global__ __launch_bounds__(512, 2) void gemm_pp4m_kernel(GemmArgs p) { gemm_pp_body<false, false, true, false, false, true>(p); }
__global__ __launch_bounds__(512, 2) void gemm_pp4m_f16_kernel(GemmArgs p) { gemm_pp_body<false, false, true, false, true, true>(p); }
__global__ __launch_bounds__(512, 2) void gemm_pp4_fp8_kernel(GemmArgs p) { gemm_pp_body<false, true, true>(p); }
__global__ __launch_bounds__(512, 2) void gemm_pp4_tn_kernel(GemmArgs p) { gemm_pp_body<true, false, true>(p); }
#ifdef HVLM_DEV
// Development builds only (`make DEV=1`): the superseded schedules / MFMA shapes kept for A/B measurements, and the K-loop
// ablation kernel (GemmArgs::dbg — WRONG RESULTS by construction, timing / PMC only).  hvlm_configure selects them.
__global__ __launch_bounds__(512, 2) void gemm_pp_kernel(GemmArgs p) { gemm_pp_body<false, false>(p); }          // 8-phase
__global__ __launch_bounds__(512, 2) void gemm_pp_tn_kernel(GemmArgs p) { gemm_pp_body<true, false>(p); }        // 8-phase
__global__ __launch_bounds__(512, 2) void gemm_pp_fp8_kernel(GemmArgs p) { gemm_pp_body<false, true>(p); }       // 8-phase
__global__ __launch_bounds__(512, 2) void gemm_pp4_kernel(GemmArgs p) { gemm_pp_body<false, false, true>(p); }   // 32x32x16
__global__ __launch_bounds__(512, 2) void gemm_pp4_f16_kernel(GemmArgs p) { gemm_pp_body<false, false, true, false, true>(p); }
__global__ __launch_bounds__(512, 2) void gemm_pp4_dbg_kernel(GemmArgs p) { gemm_pp_body<false, false, true, true>(p); }
__global__ __launch_bounds__(512, 2) void gemm_pp4m_fp8_kernel(GemmArgs p) { gemm_pp_body<false, true, true, false, false, true>(p); }   // 16x16x128
// the production kernels with the register-exchange epilogue (v_permlane swaps) instead of the LDS round trip: gemm_epi_swap = 1
__global__ __launch_bounds__(512, 2) void gemm_pp4m_swapepi_kernel(GemmArgs p) { gemm_pp_body<false, false, true, false, false, true, true>(p); }
__global__ __launch_bounds__(512, 2) void gemm_pp4m_f16_swapepi_kernel(GemmArgs p) { gemm_pp_body<false, false, true, false, true, true, true>(p); }
__global__ __launch_bounds__(512, 2) void gemm_pp4_fp8_swapepi_kernel(GemmArgs p) { gemm_pp_body<false, true, true, false, false, false, true>(p); }
#endif

template <void (*kfn)(GemmArgs)>
int launch_pp(const GemmArgs& p, int grid, hipStream_t stream) {
  static bool attr_done = false;          // one per kernel (the kernel is the template argument)
  if (!attr_done) {
    hipError_t e = hipFuncSetAttribute((const void*)kfn, hipFuncAttributeMaxDynamicSharedMemorySize, PP_LDS_BYTES);
    if (e != hipSuccess) {
      set_error("gemm_pp: hipFuncSetAttribute failed: %s", hipGetErrorString(e));
      return HVLM_ERR_LAUNCH;
    }
    attr_done = true;
  }
  hipLaunchKernelGGL(kfn, dim3(grid), dim3(512), PP_LDS_BYTES, stream, p);
  return HVLM_OK;
}

}  // namespace

// (the scheduler-state block of the current device's handle — hvlm_create — is looked up in runtime.hip: pp_sched_ws())

int pp_sched_slot(hipStream_t stream) {
  static hipStream_t known[PP_NSCHED];
  static int n = 0;
  for (int i = 0; i < n; ++i)
    if (known[i] == stream) return i;
  if (n == PP_NSCHED) return -1;
  known[n] = stream;
  return n++;
}

// The A/B switches (config().gemm_seg4 / gemm_m16 / gemm_tn_seg4 / gemm_fp8_m16 / gemm_dbg) select kernels that exist in
// development builds only; hvlm_configure refuses non-default values in a production build, so they are constant here.
static bool use_seg4() { return config().gemm_seg4 != 0; }
static bool use_m16() { return config().gemm_m16 != 0; }
int launch_gemm_pp_f16(const GemmArgs& p, int grid, hipStream_t stream) {
#ifdef HVLM_DEV
  if (!use_m16()) return launch_pp<gemm_pp4_f16_kernel>(p, grid, stream);
  if (p.epi_swap) return launch_pp<gemm_pp4m_f16_swapepi_kernel>(p, grid, stream);
#endif
  return launch_pp<gemm_pp4m_f16_kernel>(p, grid, stream);
}
int launch_gemm_pp(const GemmArgs& p, int grid, hipStream_t stream) {
#ifdef HVLM_DEV
  if (p.dbg) return launch_pp<gemm_pp4_dbg_kernel>(p, grid, stream);
  if (!use_seg4()) return launch_pp<gemm_pp_kernel>(p, grid, stream);
  if (!use_m16()) return launch_pp<gemm_pp4_kernel>(p, grid, stream);
  if (p.epi_swap) return launch_pp<gemm_pp4m_swapepi_kernel>(p, grid, stream);
#endif
  return launch_pp<gemm_pp4m_kernel>(p, grid, stream);
}
int launch_gemm_pp_tn(const GemmArgs& p, int grid, hipStream_t stream) {
#ifdef HVLM_DEV
  if (!config().gemm_tn_seg4) return launch_pp<gemm_pp_tn_kernel>(p, grid, stream);
#endif
  return launch_pp<gemm_pp4_tn_kernel>(p, grid, stream);
}
int launch_gemm_pp_fp8(const GemmArgs& p, int grid, hipStream_t stream) {
  // fp8: the 16x16x128 form is NOT the default — the matrix pipe alone gains 5 % from it under the power cap (4605 vs 4369
  // TFLOP/s), but this kernel is not matrix-power-bound at fp8 (operand traffic per unit time doubles): measured 2471 vs 2506
  // TFLOP/s on 13872x12288x4096, 2875 vs 2859 on 8192^3.
#ifdef HVLM_DEV
  if (!use_seg4()) return launch_pp<gemm_pp_fp8_kernel>(p, grid, stream);
  if (config().gemm_fp8_m16 == 1) return launch_pp<gemm_pp4m_fp8_kernel>(p, grid, stream);
  if (p.epi_swap) return launch_pp<gemm_pp4_fp8_swapepi_kernel>(p, grid, stream);
#endif
  return launch_pp<gemm_pp4_fp8_kernel>(p, grid, stream);
}

}  // namespace hvlm

#ifdef HVLM_GEMM_PROF
extern "C" int hvlm_debug_gemm_prof(void* host_dst, int clear) {
  if (clear) {
    void* dptr = nullptr;
    if (hipGetSymbolAddress(&dptr, HIP_SYMBOL(hvlm::g_pp_prof)) != hipSuccess) return 1;
    return hipMemset(dptr, 0, sizeof(hvlm::g_pp_prof)) == hipSuccess ? 0 : 1;
  }
  return hipMemcpyFromSymbol(host_dst, HIP_SYMBOL(hvlm::g_pp_prof), sizeof(hvlm::g_pp_prof)) == hipSuccess ? 0 : 1;
}
#endif
extern "C" size_t hvlm_gemm_sched_ws_bytes(void) { return sizeof(hvlm::PPSched) * hvlm::PP_NSCHED; }
