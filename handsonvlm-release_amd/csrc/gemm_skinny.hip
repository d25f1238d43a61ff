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
// Bound: HBM.  Algorithmic bytes per launch = N * K * 2 (W once); x is 32 * K * 2 per workgroup column, L2-resident.
#include "common.h"

namespace hvlm {
int launch_splitk_finalize(const float* ws, int splits, size_t slab, const bf16_t* bias, const bf16_t* res, int ldr, int act,
                           bf16_t* C, int ldc, int M, int N, hipStream_t stream);
namespace {

constexpr int SK_WAVES = 8;            // 128 rows of W per workgroup
#ifndef HVLM_SK_DEPTH
#define HVLM_SK_DEPTH 8
#endif
constexpr int SK_DEPTH = HVLM_SK_DEPTH; // W loads in flight per lane (x 16 B)
static_assert(SK_WAVES == 8, "x staging assumes 8 waves");
constexpr int SK_PAD = 8;              // elements of padding per x row in LDS

template <int HALVES>                  // 16-row groups of x: 1 (M <= 16) or 2 (M <= 32)
__global__ __launch_bounds__(SK_WAVES * 64) void gemm_skinny_kernel(const bf16_t* __restrict__ A, const bf16_t* __restrict__ W,
                                                                     float* __restrict__ ws, int M, int N, int K, int lda,
                                                                     int ldb, int steps_per_split, size_t slab) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  bf16_t* xs = reinterpret_cast<bf16_t*>(smem);
  const int t = threadIdx.x, w = t >> 6, l = t & 63, r16 = l & 15, g = l >> 4;
  const int total_steps = K >> 5;
  const int s0 = blockIdx.y * steps_per_split;
  const int nsteps = min(steps_per_split, total_steps - s0);                  // 32-deep k steps of this slice
  const int padded = (steps_per_split + SK_DEPTH - 1) / SK_DEPTH * SK_DEPTH;    // the loop runs whole rings; x is zero beyond nsteps
  const int xstride = padded * 32 + SK_PAD;
  const int kbase = s0 * 32;
  const int nw = blockIdx.x * (SK_WAVES * 16) + w * 16;                         // first of this wave's 16 rows of W
  const int n = nw + r16;                                                       // MFMA column of this lane
  const bf16_t* wp = W + (size_t)min(nw + (l >> 2), N - 1) * ldb + kbase + (l & 3) * 8;   // coalesced load order
  const int perm = (4 * r16 + g) << 2;                                           // fragment lane (r16, g) <- load lane 4 * r16 + g

  u32x4 ring[SK_DEPTH];
#pragma unroll
  for (int i = 0; i < SK_DEPTH; ++i) ring[i] = *reinterpret_cast<const u32x4*>(wp + min(i, nsteps - 1) * 32);

  // x slice -> LDS: wave w fills rows w, w + 8, ...; rows >= M and steps >= nsteps are zero.  A slice is at most 32 steps = 128
  // chunks of 16 B per row, two per lane: all 4 * HALVES loads are issued behind the ring's, straight-line (a loop here
  // gets a vmcnt(0) in its preheader, which drains the ring, and one load -> wait -> write per trip)
  {
    constexpr int XR = 2 * HALVES;
    u32x4 v[XR][2];
#pragma unroll
    for (int mi = 0; mi < XR; ++mi) {
      const bf16_t* ap = A + (size_t)min(w + SK_WAVES * mi, M - 1) * lda + kbase;
#pragma unroll
      for (int u = 0; u < 2; ++u) v[mi][u] = *reinterpret_cast<const u32x4*>(ap + min(l + 64 * u, nsteps * 4 - 1) * 8);
    }
#pragma unroll
    for (int mi = 0; mi < XR; ++mi)
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        const int m = w + SK_WAVES * mi, c = l + 64 * u;
        const u32x4 z = {0u, 0u, 0u, 0u};
        if (c < padded * 4) *reinterpret_cast<u32x4*>(xs + m * xstride + c * 8) = (m < M && c < nsteps * 4) ? v[mi][u] : z;
      }
  }
  __syncthreads();

  f32x4 acc[HALVES];
#pragma unroll
  for (int h = 0; h < HALVES; ++h) acc[h] = f32x4{0.f, 0.f, 0.f, 0.f};
  const bf16_t* xr = xs + r16 * xstride + g * 8;
  for (int s = 0; s < padded; s += SK_DEPTH) {
#pragma unroll
    for (int i = 0; i < SK_DEPTH; ++i) {
      // the reload is issued after the MFMAs that read the slot, so it can land in the same registers (a reload ahead of
      // them makes the compiler rotate the ring with copies at the loop end, behind vmcnt(0))
      u32x4 wq;
#pragma unroll
      for (int j = 0; j < 4; ++j) wq[j] = (unsigned)__builtin_amdgcn_ds_bpermute(perm, (int)ring[i][j]);
      const bf16x8 wf = __builtin_bit_cast(bf16x8, wq);
#pragma unroll
      for (int h = 0; h < HALVES; ++h) {
        const bf16x8 xf = *reinterpret_cast<const bf16x8*>(xr + h * 16 * xstride + (s + i) * 32);
        acc[h] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(xf, wf, acc[h], 0, 0, 0);
      }
      ring[i] = *reinterpret_cast<const u32x4*>(wp + min(s + i + SK_DEPTH, nsteps - 1) * 32);
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

}  // namespace
}  // namespace hvlm

using namespace hvlm;

// Steps of 32 k per K slice for (N, K): 8, 16, 24 or 32 (the x slice of a workgroup lives in LDS: at most 32 x 1024 bf16).
// The kernel is a stream with one workgroup-sized quantum, so the choice is a balance question: with `cols` workgroup
// columns and s slices there are cols * s workgroups dealt over 256 CUs, the busiest CU streams ceil(cols * s / 256)
// of them, and each costs its `per` steps plus about four steps' worth of bytes for its fp32 partial tile (written here,
// read by the finalize pass).  Measured on MI355X (profiles/r02v_skinny_gemm.txt): down-projection 11 slices of 32 steps =
// 352 workgroups 29.0 us, 15 slices of 24 = 480 workgroups 23.2 us; o-projection 8 x 16 steps 9.7 us, 16 x 8 12.8 us.
// HVLM_SKINNY_PER forces a value (development).
static int skinny_steps_per_split(int N, int K) {
  static int forced = -1, cus = 0;
  if (forced < 0) { const char* e = getenv("HVLM_SKINNY_PER"); forced = e ? atoi(e) : 0; }
  if (!cus) {
    int dev = 0; hipDeviceProp_t prop;
    cus = (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess) ? prop.multiProcessorCount : 256;
  }
  if (forced == 8 || forced == 16 || forced == 24 || forced == 32) return forced;
  const int cols = cdiv(N, SK_WAVES * 16), steps = K / 32;
  int best = 8; long best_cost = -1;
  for (int per = 32; per >= 8; per -= 8) {
    const long wgs = (long)cols * cdiv(steps, per);
    const long cost = cdiv(wgs, cus) * (per + 4);
    if (best_cost < 0 || cost < best_cost) { best = per; best_cost = cost; }
  }
  return best;
}
extern "C" int hvlm_gemm_skinny_splits(int N, int K) { return cdiv(K / 32, skinny_steps_per_split(N, K)); }

extern "C" size_t hvlm_gemm_skinny_ws_bytes(int M, int N, int K) { return (size_t)hvlm_gemm_skinny_splits(N, K) * M * N * 4; }

extern "C" int hvlm_gemm_bf16_nt_skinny(const void* A, const void* B, void* C, int M, int N, int K, int lda, int ldb, int ldc,
                                        const void* bias, const void* residual, int ldr, int act, void* ws, void* stream) {
  HVLM_REQUIRE(A && B && C && ws, "gemm_skinny: null pointer");
  HVLM_REQUIRE(M > 0 && M <= 32 && N > 0 && K > 0, "gemm_skinny: M must be 1 .. 32 (got %d)", M);
  HVLM_REQUIRE(K % 32 == 0, "gemm_skinny: K=%d must be a multiple of 32", K);
  HVLM_REQUIRE(N % 4 == 0 && lda >= K && ldb >= K && ldc >= N && lda % 8 == 0 && ldb % 8 == 0 && ldc % 4 == 0,
               "gemm_skinny: bad leading dimensions (N %% 4)");
  HVLM_REQUIRE(((uintptr_t)A & 15) == 0 && ((uintptr_t)B & 15) == 0 && ((uintptr_t)C & 7) == 0 && ((uintptr_t)ws & 15) == 0,
               "gemm_skinny: alignment");
  HVLM_REQUIRE(!residual || ldr >= N, "gemm_skinny: bad residual");
  HVLM_REQUIRE(act == HVLM_ACT_NONE || act == HVLM_ACT_QUICK_GELU, "gemm_skinny: unknown act %d", act);
  const int per = skinny_steps_per_split(N, K), splits = cdiv(K / 32, per);
  const int padded = cdiv(per, SK_DEPTH) * SK_DEPTH;
  const int halves = M <= 16 ? 1 : 2;
  const size_t lds = (size_t)halves * 16 * (padded * 32 + SK_PAD) * 2;
  static bool attr_done = false;
  if (!attr_done) {
    const int cap = 2 * 16 * (32 * 32 + SK_PAD) * 2;
    hipError_t e = hipFuncSetAttribute((const void*)gemm_skinny_kernel<1>, hipFuncAttributeMaxDynamicSharedMemorySize, cap);
    if (e == hipSuccess) e = hipFuncSetAttribute((const void*)gemm_skinny_kernel<2>, hipFuncAttributeMaxDynamicSharedMemorySize, cap);
    if (e != hipSuccess) { set_error("gemm_skinny: hipFuncSetAttribute: %s", hipGetErrorString(e)); return HVLM_ERR_LAUNCH; }
    attr_done = true;
  }
  const dim3 grid(cdiv(N, SK_WAVES * 16), splits), block(SK_WAVES * 64);
  if (halves == 1)
    hipLaunchKernelGGL((gemm_skinny_kernel<1>), grid, block, lds, as_stream(stream), (const bf16_t*)A, (const bf16_t*)B,
                       (float*)ws, M, N, K, lda, ldb, per, (size_t)M * N);
  else
    hipLaunchKernelGGL((gemm_skinny_kernel<2>), grid, block, lds, as_stream(stream), (const bf16_t*)A, (const bf16_t*)B,
                       (float*)ws, M, N, K, lda, ldb, per, (size_t)M * N);
  const int rc = check_launch("gemm_skinny");
  if (rc) return rc;
  return launch_splitk_finalize((const float*)ws, splits, (size_t)M * N, (const bf16_t*)bias, (const bf16_t*)residual, ldr, act,
                                (bf16_t*)C, ldc, M, N, as_stream(stream));
}
