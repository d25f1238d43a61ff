// Weight-streaming GEMM for M <= 32 rows (lock-step generation of a batch of samples: BASELINE config 5's decode steps, where
// every projection is out[B, N] = x[B, K] @ W[N, K]^T with B = 32 and the whole cost is reading W once from HBM).
//
// The 256 x 256 tile kernel spends 8x the MFMA work and a 256-row LDS pipeline on such a problem and streams W at 3.0 TB/s.
// Here W never touches LDS: a wave owns 16 rows of W and streams them through a register ring, DEPTH loads of 16 rows x 64 B in
// flight per wave.  A load is issued in the COALESCED lane order (lane = row l >> 2, 16-B chunk l & 3: four adjacent lanes
// read 64 contiguous bytes) and turned into the MFMA B-fragment order (v_mfma_f32_16x16x32_bf16: lane = row l & 15, chunk
// l >> 4) by four ds_bpermute_b32 when it is consumed.  Loading in fragment order directly makes every lane its own 16-B
// TCP access (adjacent lanes are rows 8 KB apart): 4x the tag look-ups, TA_BUSY 58 % at 3.9 TB/s (profiles/r02v_*).
// x (at most 32 rows) is the A-fragment, staged once per workgroup into a padded LDS image (row stride + 16 B: the sixteen
// rows of a fragment read land on distinct banks) and shared by the eight waves.  K is split over blockIdx.y; partial sums go to fp32 slabs and the
// split-K finalize kernel of gemm.hip applies bias / activation / residual.  (Reducing in the same launch — last workgroup
// of a column block behind an agent-scope release / acquire pair — was measured 2x SLOWER: every workgroup's release
// writes back its XCD's L2, 25 -> 58 us on the qkv projection; profiles/r02v_skinny_gemm.txt.)
//
// Also here: the same kernel on e4m3 weights (W8 = true: one byte per element, converted to bf16 in registers; per-channel scale
// in the finalize pass), and the two finalize passes of the decode step that do their consumer's work (RoPE + cache row write
// for q|k|v, SwiGLU for gate|up).
//
// Bound: HBM.  Algorithmic bytes per launch = N * K * 2 (W once; N * K for e4m3); x is 32 * K * 2 per workgroup column, L2-resident.
#include "common.h"

namespace hvlm {
int launch_splitk_finalize(const float* ws, int splits, size_t slab, const float* col_scale, const bf16_t* bias,
                           const bf16_t* res, int ldr, int act, bf16_t* C, int ldc, int M, int N, hipStream_t stream);
namespace {

constexpr int SK_WAVES = 8;            // 128 rows of W per workgroup
#ifndef HVLM_SK_DEPTH
#define HVLM_SK_DEPTH 8
#endif
constexpr int SK_DEPTH = HVLM_SK_DEPTH; // W loads in flight per lane (x 16 B)
static_assert(SK_WAVES == 8 && SK_DEPTH % 2 == 0, "x staging assumes 8 waves; the e4m3 ring is half as deep");
constexpr int SK_PAD = 8;              // elements of padding per x row in LDS

// eight e4m3 bytes (two dwords, k ascending with the byte address) -> eight bf16: exact (3 mantissa bits)
__device__ __forceinline__ bf16x8 e4m3x8_to_bf16(unsigned a, unsigned b) {
  const f32x2 p0 = __builtin_amdgcn_cvt_pk_f32_fp8((int)a, false), p1 = __builtin_amdgcn_cvt_pk_f32_fp8((int)a, true);
  const f32x2 p2 = __builtin_amdgcn_cvt_pk_f32_fp8((int)b, false), p3 = __builtin_amdgcn_cvt_pk_f32_fp8((int)b, true);
  const u32x4 r = {pack_bf2(p0[0], p0[1]), pack_bf2(p1[0], p1[1]), pack_bf2(p2[0], p2[1]), pack_bf2(p3[0], p3[1])};
  return __builtin_bit_cast(bf16x8, r);
}

// HALVES: 16-row groups of x, 1 (M <= 16) or 2 (M <= 32).  W8: W is e4m3, one byte per element (weight-only quantisation:
// converted to bf16 in registers, x stays bf16, the per-output-channel scale is applied by the finalize pass).  A "step" is
// the 64 bytes of a W row one load instruction covers: 32 k of bf16 or 64 k of e4m3, DEPTH steps in flight.
template <int HALVES, bool W8>
__global__ __launch_bounds__(SK_WAVES * 64) void gemm_skinny_kernel(const bf16_t* __restrict__ A, const char* __restrict__ W,
                                                                     float* __restrict__ ws, int M, int N, int K, int lda,
                                                                     int ldb, int steps_per_split, size_t slab) {
  constexpr int KPS = W8 ? 64 : 32;      // k per step
  constexpr int DEPTH = W8 ? SK_DEPTH / 2 : SK_DEPTH;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  bf16_t* xs = reinterpret_cast<bf16_t*>(smem);
  const int t = threadIdx.x, w = t >> 6, l = t & 63, r16 = l & 15, g = l >> 4;
  const int total_steps = K / KPS;
  const int s0 = blockIdx.y * steps_per_split;
  const int nsteps = min(steps_per_split, total_steps - s0);                  // steps of this slice
  const int padded = (steps_per_split + DEPTH - 1) / DEPTH * DEPTH;            // the loop runs whole rings; x is zero beyond nsteps
  const int xstride = padded * KPS + SK_PAD;
  const int kbase = s0 * KPS;
  const int nw = blockIdx.x * (SK_WAVES * 16) + w * 16;                         // first of this wave's 16 rows of W
  const int n = nw + r16;                                                       // MFMA column of this lane
  // coalesced load order: lane = (row l >> 2, 16-byte chunk l & 3); byte addressing, 64 B per row and step in both formats
  const char* wp = W + ((size_t)min(nw + (l >> 2), N - 1) * ldb + kbase) * (W8 ? 1 : 2) + (l & 3) * 16;
  const int perm = (4 * r16 + g) << 2;                                           // fragment lane (r16, g) <- load lane 4 * r16 + g

  u32x4 ring[DEPTH];
#pragma unroll
  for (int i = 0; i < DEPTH; ++i) ring[i] = *reinterpret_cast<const u32x4*>(wp + min(i, nsteps - 1) * 64);

  // x slice -> LDS: wave w fills rows w, w + 8, ...; rows >= M and steps >= nsteps are zero.  A slice is at most 1024 k = 128
  // chunks of 16 B per row, two per lane: all 4 * HALVES loads are issued behind the ring's, straight-line (a loop here
  // gets a vmcnt(0) in its preheader, which drains the ring, and one load -> wait -> write per trip)
  {
    constexpr int XR = 2 * HALVES, CPS = KPS / 8;      // 16-byte chunks of x per step
    u32x4 v[XR][2];
#pragma unroll
    for (int mi = 0; mi < XR; ++mi) {
      const bf16_t* ap = A + (size_t)min(w + SK_WAVES * mi, M - 1) * lda + kbase;
#pragma unroll
      for (int u = 0; u < 2; ++u) v[mi][u] = *reinterpret_cast<const u32x4*>(ap + min(l + 64 * u, nsteps * CPS - 1) * 8);
    }
#pragma unroll
    for (int mi = 0; mi < XR; ++mi)
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        const int m = w + SK_WAVES * mi, c = l + 64 * u;
        const u32x4 z = {0u, 0u, 0u, 0u};
        if (c < padded * CPS) *reinterpret_cast<u32x4*>(xs + m * xstride + c * 8) = (m < M && c < nsteps * CPS) ? v[mi][u] : z;
      }
  }
  __syncthreads();

  f32x4 acc[HALVES];
#pragma unroll
  for (int h = 0; h < HALVES; ++h) acc[h] = f32x4{0.f, 0.f, 0.f, 0.f};
  // bf16: fragment lane (r16, g) holds k = 8 g .. 8 g + 7 of the step; e4m3: its 16 bytes are k = 16 g .. 16 g + 15, consumed
  // as two MFMAs (first / second eight) against the x chunks at the same k
  const bf16_t* xr = xs + r16 * xstride + g * (W8 ? 16 : 8);
  for (int s = 0; s < padded; s += DEPTH) {
#pragma unroll
    for (int i = 0; i < DEPTH; ++i) {
      // the reload is issued after the MFMAs that read the slot, so it can land in the same registers (a reload ahead of
      // them makes the compiler rotate the ring with copies at the loop end, behind vmcnt(0))
      u32x4 wq;
#pragma unroll
      for (int j = 0; j < 4; ++j) wq[j] = (unsigned)__builtin_amdgcn_ds_bpermute(perm, (int)ring[i][j]);
      if constexpr (W8) {
        const bf16x8 wa = e4m3x8_to_bf16(wq[0], wq[1]), wb = e4m3x8_to_bf16(wq[2], wq[3]);
#pragma unroll
        for (int h = 0; h < HALVES; ++h) {
          const bf16_t* xp = xr + h * 16 * xstride + (s + i) * KPS;
          const bf16x8 xa = *reinterpret_cast<const bf16x8*>(xp), xb = *reinterpret_cast<const bf16x8*>(xp + 8);
          acc[h] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(xa, wa, acc[h], 0, 0, 0);
          acc[h] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(xb, wb, acc[h], 0, 0, 0);
        }
      } else {
        const bf16x8 wf = __builtin_bit_cast(bf16x8, wq);
#pragma unroll
        for (int h = 0; h < HALVES; ++h) {
          const bf16x8 xf = *reinterpret_cast<const bf16x8*>(xr + h * 16 * xstride + (s + i) * KPS);
          acc[h] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(xf, wf, acc[h], 0, 0, 0);
        }
      }
      ring[i] = *reinterpret_cast<const u32x4*>(wp + min(s + i + DEPTH, nsteps - 1) * 64);
      __builtin_amdgcn_sched_barrier(0);     // keep the reload here: sunk to the loop end, the ring runs dry
    }
  }

  // D[m][n]: lane holds column n = l & 15, rows (l >> 4) * 4 + i of each half
  if (n < N) {
    float* o = ws + blockIdx.y * slab + n;
#pragma unroll
    for (int h = 0; h < HALVES; ++h)
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int m = h * 16 + g * 4 + i;
        if (m < M) o[(size_t)m * N] = acc[h][i];
      }
  }
}

// ---- finalize passes of the lock-step decode step that also do the consumer's work (same slab order and the same bf16 rounding
// of the GEMM output as the plain finalize pass followed by the separate kernel: bit-identical, three launches fewer per layer)

// gate|up projection: act[m, f] = SwiGLU(bf16(sum gate), bf16(sum up)) — hvlm_swiglu_fwd's arithmetic
__global__ __launch_bounds__(256) void skinny_finalize_swiglu_kernel(const float* __restrict__ ws, int splits, size_t slab,
                                                                     const float* __restrict__ col_scale, bf16_t* __restrict__ act,
                                                                     int M, int F) {
  const long total = (long)M * (F / 4);
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
    const long m = i / (F / 4);
    const int f = (int)(i - m * (F / 4)) * 4;
    const float* p = ws + m * (2L * F) + f;
    f32x4 g = *reinterpret_cast<const f32x4*>(p), u = *reinterpret_cast<const f32x4*>(p + F);
    for (int s = 1; s < splits; ++s) {
      g += *reinterpret_cast<const f32x4*>(p + s * slab);
      u += *reinterpret_cast<const f32x4*>(p + s * slab + F);
    }
    if (col_scale) {
#pragma unroll
      for (int e = 0; e < 4; ++e) { g[e] *= col_scale[f + e]; u[e] *= col_scale[F + f + e]; }
    }
    float o[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const float gb = bf2f(f2bf(g[e])), ub = bf2f(f2bf(u[e]));               // the bf16 gate|up row the unfused path stores
      const float sg = bf2f(f2bf(gb * (1.f / (1.f + __expf(-gb)))));
      o[e] = sg * ub;
    }
    const u32x2 pk = {pack_bf2(o[0], o[1]), pack_bf2(o[2], o[3])};
    *reinterpret_cast<u32x2*>(act + m * (long)F + f) = pk;
  }
}

// q|k|v projection: bf16(sum) -> RoPE of the q and k heads at position lens[m] (hvlm_rope_qk's arithmetic) -> the qkv row AND
// the k | v columns of cache row lens[m] of sample m.  One item = 4 rotation pairs of a q / k head, or 4 v columns.
__global__ __launch_bounds__(256) void skinny_finalize_qkv_kernel(const float* __restrict__ ws, int splits, size_t slab,
                                                                  const float* __restrict__ col_scale, bf16_t* __restrict__ qkv,
                                                                  int ldq, bf16_t* __restrict__ cache, long max_len,
                                                                  const int32_t* __restrict__ lens, const bf16_t* __restrict__ cos_tab,
                                                                  const bf16_t* __restrict__ sin_tab, int M, int H, int dh) {
  const int D = H * dh, half = dh / 2, N = 3 * D;
  const int rot_items = 2 * H * (half / 4), v_items = D / 4, per_row = rot_items + v_items;
  const long total = (long)M * per_row;
  auto sum4 = [&](long m, int c) {
    const float* p = ws + m * (long)N + c;
    f32x4 v = *reinterpret_cast<const f32x4*>(p);
    for (int s = 1; s < splits; ++s) v += *reinterpret_cast<const f32x4*>(p + s * slab);
    if (col_scale) {
#pragma unroll
      for (int e = 0; e < 4; ++e) v[e] *= col_scale[c + e];
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) v[e] = bf2f(f2bf(v[e]));                    // the bf16 qkv row the unfused path stores
    return v;
  };
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
    const long m = i / per_row;
    const int r = (int)(i - m * per_row);
    const int pos = lens[m];
    bf16_t* qrow = qkv + m * (long)ldq;
    bf16_t* crow = cache + (m * max_len + pos) * (long)N;
    if (r < rot_items) {
      const int hh = r / (half / 4), c = (r - hh * (half / 4)) * 4;        // head 0 .. 2H-1 (q heads, then k heads), pair offset
      const int col = hh * dh + c;
      const f32x4 x1 = sum4(m, col), x2 = sum4(m, col + half);
      const u32x2 cv = *reinterpret_cast<const u32x2*>(cos_tab + (long)pos * half + c);
      const u32x2 sv = *reinterpret_cast<const u32x2*>(sin_tab + (long)pos * half + c);
      const float cc[4] = {bf_lo(cv[0]), bf_hi(cv[0]), bf_lo(cv[1]), bf_hi(cv[1])};
      const float ss[4] = {bf_lo(sv[0]), bf_hi(sv[0]), bf_lo(sv[1]), bf_hi(sv[1])};
      float r1[4], r2[4];
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        r1[e] = bf2f(f2bf(x1[e] * cc[e])) + bf2f(f2bf(-x2[e] * ss[e]));
        r2[e] = bf2f(f2bf(x2[e] * cc[e])) + bf2f(f2bf(x1[e] * ss[e]));
      }
      const u32x2 o1 = {pack_bf2(r1[0], r1[1]), pack_bf2(r1[2], r1[3])}, o2 = {pack_bf2(r2[0], r2[1]), pack_bf2(r2[2], r2[3])};
      *reinterpret_cast<u32x2*>(qrow + col) = o1;
      *reinterpret_cast<u32x2*>(qrow + col + half) = o2;
      if (hh >= H) {                                                           // k heads also go to the cache
        *reinterpret_cast<u32x2*>(crow + col) = o1;
        *reinterpret_cast<u32x2*>(crow + col + half) = o2;
      }
    } else {
      const int col = 2 * D + (r - rot_items) * 4;
      const f32x4 v = sum4(m, col);
      const u32x2 o = {pack_bf2(v[0], v[1]), pack_bf2(v[2], v[3])};
      *reinterpret_cast<u32x2*>(qrow + col) = o;
      *reinterpret_cast<u32x2*>(crow + col) = o;
    }
  }
}

}  // namespace
}  // namespace hvlm

using namespace hvlm;

// Steps per K slice: q, 2q, 3q or 4q (q = 8 steps of 32 k for bf16 weights, 4 steps of 64 k for e4m3: slices of 256 .. 1024 k;
// the x slice of a workgroup lives in LDS).  The kernel is a stream with one workgroup-sized quantum, so the choice is a
// balance question: with `cols` workgroup columns and s slices there are cols * s workgroups dealt over the CUs, the busiest
// CU streams ceil(cols * s / CUs) of them, and each costs its steps (8 KB of W per workgroup and step in both formats) plus
// about four steps' worth of bytes for its fp32 partial tile (written here, read by the finalize pass).  Measured on MI355X
// (profiles/r02v_skinny_gemm.txt): down-projection 11 slices of 32 steps = 352 workgroups 29.0 us, 15 slices of 24 = 480
// workgroups 23.2 us; o-projection 8 x 16 steps 9.7 us, 16 x 8 12.8 us.  HVLM_SKINNY_PER forces a multiple of q (development).
static int skinny_steps_per_split(int N, int steps, int q) {
  static int cus = 0;
  const int forced = config().skinny_per;
  if (!cus) {
    int dev = 0; hipDeviceProp_t prop;
    cus = (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess) ? prop.multiProcessorCount : 256;
  }
  if (forced >= 1 && forced <= 4) return forced * q;
  const int cols = cdiv(N, SK_WAVES * 16);
  int best = q; long best_cost = -1;
  for (int per = 4 * q; per >= q; per -= q) {
    const long wgs = (long)cols * cdiv(steps, per);
    const long cost = cdiv(wgs, cus) * (per * (8 / q) + 4);      // in bf16 steps: an e4m3 step of 64 k costs the same 8 KB
    if (best_cost < 0 || cost < best_cost) { best = per; best_cost = cost; }
  }
  return best;
}
extern "C" int hvlm_gemm_skinny_splits(int N, int K) { return cdiv(K / 32, skinny_steps_per_split(N, K / 32, 8)); }
extern "C" size_t hvlm_gemm_skinny_ws_bytes(int M, int N, int K) { return (size_t)hvlm_gemm_skinny_splits(N, K) * M * N * 4; }
extern "C" int hvlm_gemm_w8a16_skinny_splits(int N, int K) { return cdiv(K / 64, skinny_steps_per_split(N, K / 64, 4)); }
extern "C" size_t hvlm_gemm_w8a16_skinny_ws_bytes(int M, int N, int K) {
  return (size_t)hvlm_gemm_w8a16_skinny_splits(N, K) * M * N * 4;
}

// launches the streaming kernel into the slabs of `ws`; *splits_out = number of slabs
template <bool W8>
static int skinny_launch(const void* A, const void* B, int M, int N, int K, int lda, int ldb, void* ws, void* stream, int* splits_out) {
  constexpr int KPS = W8 ? 64 : 32, Q = W8 ? 4 : 8, DEPTH = W8 ? SK_DEPTH / 2 : SK_DEPTH;
  HVLM_REQUIRE(A && B && ws, "gemm_skinny: null pointer");
  HVLM_REQUIRE(M > 0 && M <= 32 && N > 0 && K > 0, "gemm_skinny: M must be 1 .. 32 (got %d)", M);
  HVLM_REQUIRE(K % KPS == 0, "gemm_skinny: K=%d must be a multiple of %d", K, KPS);
  HVLM_REQUIRE(N % 4 == 0 && lda >= K && ldb >= K && lda % 8 == 0 && ldb % (W8 ? 16 : 8) == 0,
               "gemm_skinny: bad leading dimensions (N %% 4; rows of A and W 16-byte aligned)");
  HVLM_REQUIRE(((uintptr_t)A & 15) == 0 && ((uintptr_t)B & 15) == 0 && ((uintptr_t)ws & 15) == 0, "gemm_skinny: alignment");
  const int per = skinny_steps_per_split(N, K / KPS, Q), splits = cdiv(K / KPS, per);
  const int padded = cdiv(per, DEPTH) * DEPTH;
  const int halves = M <= 16 ? 1 : 2;
  const size_t lds = (size_t)halves * 16 * (padded * KPS + SK_PAD) * 2;
  static bool attr_done = false;
  if (!attr_done) {
    const int cap = 2 * 16 * (1024 + SK_PAD) * 2;
    hipError_t e = hipFuncSetAttribute((const void*)gemm_skinny_kernel<1, W8>, hipFuncAttributeMaxDynamicSharedMemorySize, cap);
    if (e == hipSuccess) e = hipFuncSetAttribute((const void*)gemm_skinny_kernel<2, W8>, hipFuncAttributeMaxDynamicSharedMemorySize, cap);
    if (e != hipSuccess) { set_error("gemm_skinny: hipFuncSetAttribute: %s", hipGetErrorString(e)); return HVLM_ERR_LAUNCH; }
    attr_done = true;
  }
  const dim3 grid(cdiv(N, SK_WAVES * 16), splits), block(SK_WAVES * 64);
  if (halves == 1)
    hipLaunchKernelGGL((gemm_skinny_kernel<1, W8>), grid, block, lds, as_stream(stream), (const bf16_t*)A, (const char*)B,
                       (float*)ws, M, N, K, lda, ldb, per, (size_t)M * N);
  else
    hipLaunchKernelGGL((gemm_skinny_kernel<2, W8>), grid, block, lds, as_stream(stream), (const bf16_t*)A, (const char*)B,
                       (float*)ws, M, N, K, lda, ldb, per, (size_t)M * N);
  *splits_out = splits;
  return check_launch(W8 ? "gemm_w8a16_skinny" : "gemm_skinny");
}

template <bool W8>
static int skinny_impl(const void* A, const void* B, const float* col_scale, void* C, int M, int N, int K, int lda, int ldb,
                       int ldc, const void* bias, const void* residual, int ldr, int act, void* ws, void* stream) {
  HVLM_REQUIRE(C && ldc >= N && ldc % 4 == 0 && ((uintptr_t)C & 7) == 0, "gemm_skinny: bad C / ldc");
  HVLM_REQUIRE(!residual || ldr >= N, "gemm_skinny: bad residual");
  HVLM_REQUIRE(act == HVLM_ACT_NONE || act == HVLM_ACT_QUICK_GELU, "gemm_skinny: unknown act %d", act);
  int splits = 0;
  const int rc = skinny_launch<W8>(A, B, M, N, K, lda, ldb, ws, stream, &splits);
  if (rc) return rc;
  return launch_splitk_finalize((const float*)ws, splits, (size_t)M * N, col_scale, (const bf16_t*)bias, (const bf16_t*)residual,
                                ldr, act, (bf16_t*)C, ldc, M, N, as_stream(stream));
}

// ---- the two projections of the decode step whose consumer rides in the finalize pass ---------------------------------------
extern "C" int hvlm_gemm_skinny_swiglu(const void* A, const void* B, const float* col_scale, void* act, int M, int F, int K, int lda,
                                       int ldb, void* ws, void* stream) {
  HVLM_REQUIRE(act && F > 0 && F % 4 == 0 && ((uintptr_t)act & 7) == 0, "gemm_skinny_swiglu: bad act / F");
  int splits = 0;
  const int rc = col_scale ? skinny_launch<true>(A, B, M, 2 * F, K, lda, ldb, ws, stream, &splits)
                           : skinny_launch<false>(A, B, M, 2 * F, K, lda, ldb, ws, stream, &splits);
  if (rc) return rc;
  const long total = (long)M * (F / 4);
  int fg = (int)((total + 255) / 256); if (fg > 2048) fg = 2048;
  hipLaunchKernelGGL(skinny_finalize_swiglu_kernel, dim3(fg), dim3(256), 0, as_stream(stream), (const float*)ws, splits,
                     (size_t)M * 2 * F, col_scale, (bf16_t*)act, M, F);
  return check_launch("gemm_skinny_swiglu");
}

extern "C" int hvlm_gemm_skinny_qkv(const void* A, const void* B, const float* col_scale, void* qkv, int ldq, void* cache,
                                    long max_len, const int32_t* lens, const void* cos_tab, const void* sin_tab, int M, int H,
                                    int dh, int K, int lda, int ldb, void* ws, void* stream) {
  HVLM_REQUIRE(qkv && cache && lens && cos_tab && sin_tab, "gemm_skinny_qkv: null pointer");
  HVLM_REQUIRE(H > 0 && dh % 8 == 0 && ldq >= 3 * H * dh && ldq % 4 == 0 && max_len > 0, "gemm_skinny_qkv: bad shape");
  const int N = 3 * H * dh;
  int splits = 0;
  const int rc = col_scale ? skinny_launch<true>(A, B, M, N, K, lda, ldb, ws, stream, &splits)
                           : skinny_launch<false>(A, B, M, N, K, lda, ldb, ws, stream, &splits);
  if (rc) return rc;
  const long total = (long)M * (2 * H * (dh / 8) + H * dh / 4);
  int fg = (int)((total + 255) / 256); if (fg > 2048) fg = 2048;
  hipLaunchKernelGGL(skinny_finalize_qkv_kernel, dim3(fg), dim3(256), 0, as_stream(stream), (const float*)ws, splits, (size_t)M * N,
                     col_scale, (bf16_t*)qkv, ldq, (bf16_t*)cache, max_len, lens, (const bf16_t*)cos_tab, (const bf16_t*)sin_tab, M, H, dh);
  return check_launch("gemm_skinny_qkv");
}

extern "C" int hvlm_gemm_bf16_nt_skinny(const void* A, const void* B, void* C, int M, int N, int K, int lda, int ldb, int ldc,
                                        const void* bias, const void* residual, int ldr, int act, void* ws, void* stream) {
  return skinny_impl<false>(A, B, nullptr, C, M, N, K, lda, ldb, ldc, bias, residual, ldr, act, ws, stream);
}

extern "C" int hvlm_gemm_w8a16_nt_skinny(const void* A, const void* B8, const float* col_scale, void* C, int M, int N, int K,
                                         int lda, int ldb, int ldc, const void* bias, const void* residual, int ldr, int act,
                                         void* ws, void* stream) {
  HVLM_REQUIRE(col_scale, "gemm_w8a16_skinny: null col_scale");
  return skinny_impl<true>(A, B8, col_scale, C, M, N, K, lda, ldb, ldc, bias, residual, ldr, act, ws, stream);
}
