// Host-side launchers of the bf16 / fp8 NT GEMM  C[M,N] = act(A[M,K] @ B[N,K]^T + bias) + residual  and the small kernels
// around it (split-K finalize, fp8 quantisers).  The GEMM kernel itself is the ping-pong kernel of gemm_pp.hip
// (256x256 tiles, persistent workgroups, LDS-DMA operand ring, 4-segment schedule); the first-generation kernels that
// used to live here (single-barrier 2-stage loop and its A/B variants, HVLM_GEMM_VARIANT) were removed in round 2 — their
// measurements stay in DESIGN.md §3 / profiles/r01b_gemm_variants.txt.
//
// Reference arithmetic replaced: every nn.Linear on the path (see include/hvlm.h).
#include "common.h"
#include "gemm_args.h"
#include <stdlib.h>

namespace hvlm {

namespace {

constexpr int BM = 256, BN = 256, BK = 64;
constexpr int GROUP_M = 4;

__device__ __forceinline__ float quick_gelu(float x) { return x / (1.f + __expf(-1.702f * x)); }

}  // namespace

}  // namespace hvlm

using namespace hvlm;

// development ablations of the ping-pong kernel's K loop (wrong results by construction; timing / PMC only):
// 1 = no LDS-DMA in load segment B, 2 = none in load segment A, 4 = staging cursors never advance (cache-hot source),
// 8 = no fragment reads in load segment B (8 of the 24 ds_read_b128 per K-tile and wave)
static int gemm_dbg() { return config().gemm_dbg; }      // (always 0 in a production build: hvlm_configure refuses it)

static int gemm16_impl(const void* A, const void* B, void* C, int M, int N, int K, int lda, int ldb, int ldc,
                       const void* bias, const void* residual, int ldr, int act, int out_f32, bool f16, void* stream,
                       float* stats_out = nullptr, const float* ln_stats = nullptr, const float* ln_c1 = nullptr);

extern "C" size_t hvlm_ln_stats_parts(int N) { return (size_t)cdiv(N, BN) * 4; }

extern "C" int hvlm_gemm_f16_nt_ln(const void* A, const void* B, void* C, int M, int N, int K, int lda, int ldb, int ldc,
                                   const void* bias, const void* residual, int ldr, int act, float* stats_out,
                                   const float* ln_stats, const float* ln_c1, void* stream) {
  HVLM_REQUIRE(!stats_out || (residual && N % BN == 0 && ((uintptr_t)stats_out & 7) == 0),
               "gemm_f16_nt_ln: stats_out needs a residual epilogue and N %% 256 == 0");
  HVLM_REQUIRE((ln_stats == nullptr) == (ln_c1 == nullptr), "gemm_f16_nt_ln: ln_stats and ln_c1 come together");
  HVLM_REQUIRE(!ln_stats || (!residual && N % BN == 0 && ((uintptr_t)ln_stats & 7) == 0 && ((uintptr_t)ln_c1 & 15) == 0),
               "gemm_f16_nt_ln: the LayerNorm-fold epilogue takes no residual and needs N %% 256 == 0");
  HVLM_REQUIRE(ldc % 8 == 0 && (!bias || ((uintptr_t)bias & 15) == 0) && (!residual || (ldr % 8 == 0 && ((uintptr_t)residual & 15) == 0)),
               "gemm_f16_nt_ln: 16-byte aligned rows (the fast epilogue)");
  return gemm16_impl(A, B, C, M, N, K, lda, ldb, ldc, bias, residual, ldr, act, 0, true, stream, stats_out, ln_stats, ln_c1);
}

extern "C" int hvlm_gemm_bf16_nt(const void* A, const void* B, void* C, int M, int N, int K, int lda,
                                 int ldb, int ldc, const void* bias, const void* residual, int ldr,
                                 int act, int out_f32, void* stream) {
  return gemm16_impl(A, B, C, M, N, K, lda, ldb, ldc, bias, residual, ldr, act, out_f32, false, stream);
}

extern "C" int hvlm_gemm_f16_nt(const void* A, const void* B, void* C, int M, int N, int K, int lda,
                                int ldb, int ldc, const void* bias, const void* residual, int ldr,
                                int act, int out_f32, void* stream) {
  return gemm16_impl(A, B, C, M, N, K, lda, ldb, ldc, bias, residual, ldr, act, out_f32, true, stream);
}

static int gemm16_impl(const void* A, const void* B, void* C, int M, int N, int K, int lda, int ldb, int ldc,
                       const void* bias, const void* residual, int ldr, int act, int out_f32, bool f16, void* stream,
                       float* stats_out, const float* ln_stats, const float* ln_c1) {
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
  p.stats_out = stats_out; p.ln_stats = ln_stats; p.ln_c1 = ln_c1;
  p.splits = 1; p.k_split = K; p.slab = 0; p.alpha = 1.f; p.dyn = 0; p.dbg = gemm_dbg(); p.epi_swap = config().gemm_epi_swap; p.tail_half = config().gemm_tail_half; p.sched = pp_sched_ws();
  p.group_m = config().gemm_group_m;
  int grid = p.tiles_m * p.tiles_n;
  static int pgrid = 0;
  if (!pgrid) {
    int dev = 0; hipDeviceProp_t prop;
    if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess) pgrid = prop.multiProcessorCount;
    else pgrid = 256;
    pgrid = (pgrid / 8) * 8;
  }
  // dynamic tile scheduling needs >= 3 units per workgroup: every workgroup commits to three units when it starts,
  // so with fewer the first arrivals would take everything (256 tiles on 256 CUs: 1383 -> 917 TFLOP/s)
  p.dyn = (K / BK >= 3 && pgrid <= 512 && grid >= 3 * pgrid && !config().gemm_static) ? 1 + pp_sched_slot(as_stream(stream)) : 0;
  const int rc = f16 ? launch_gemm_pp_f16(p, grid < pgrid ? grid : pgrid, as_stream(stream))
                     : launch_gemm_pp(p, grid < pgrid ? grid : pgrid, as_stream(stream));
  if (rc) return rc;
  return check_launch(f16 ? "gemm_f16_nt" : "gemm_bf16_nt");
}

// ---- split-K for skinny problems (few output tiles, e.g. the batch-1 prefill M = 867) --------------------------
namespace hvlm {
namespace {
__global__ __launch_bounds__(256) void splitk_finalize_kernel(const float* __restrict__ ws, int splits, size_t slab,
                                                              const float* __restrict__ col_scale,
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
      if (col_scale) o[e] *= col_scale[n + e];         // per-output-channel weight scale (e4m3 weights)
      if (bias) o[e] += bf2f(bias[n + e]);
      if (act == HVLM_ACT_QUICK_GELU) o[e] = quick_gelu(o[e]);
      if (res) o[e] += bf2f(res[m * (long)ldr + n + e]);
    }
    u32x2 pk = {pack_bf2(o[0], o[1]), pack_bf2(o[2], o[3])};
    *reinterpret_cast<u32x2*>(C + m * (long)ldc + n) = pk;
  }
}
}  // namespace

// shared with gemm_skinny.hip
int launch_splitk_finalize(const float* ws, int splits, size_t slab, const float* col_scale, const bf16_t* bias,
                           const bf16_t* res, int ldr, int act, bf16_t* C, int ldc, int M, int N, hipStream_t stream) {
  const long total = (long)M * (N / 4);
  int fg = (int)((total + 255) / 256); if (fg > 2048) fg = 2048;
  hipLaunchKernelGGL(splitk_finalize_kernel, dim3(fg), dim3(256), 0, stream, ws, splits, slab, col_scale, bias, res, ldr, act, C, ldc, M, N);
  return check_launch("gemm_splitk_finalize");
}
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
  p.splits = splits; p.k_split = K / splits; p.slab = (size_t)M * N; p.alpha = 1.f; p.dyn = 0; p.dbg = gemm_dbg(); p.epi_swap = config().gemm_epi_swap; p.tail_half = config().gemm_tail_half; p.sched = pp_sched_ws();
  static int cus = 0;
  if (!cus) {
    int dev = 0; hipDeviceProp_t prop;
    cus = (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess) ? prop.multiProcessorCount : 256;
    cus = (cus / 8) * 8;
  }
  int grid = p.tiles_m * p.tiles_n * splits;
  if (grid > cus) grid = cus;
  p.dyn = (p.k_split / BK >= 3 && cus <= 512 && p.tiles_m * p.tiles_n * splits >= 3 * cus && !config().gemm_static) ? 1 + pp_sched_slot(as_stream(stream)) : 0;
  int rc = launch_gemm_pp(p, grid, as_stream(stream));
  if (rc) return rc;
  rc = check_launch("gemm_splitk");
  if (rc) return rc;
  return launch_splitk_finalize((const float*)ws, splits, (size_t)M * N, nullptr, (const bf16_t*)bias, (const bf16_t*)residual, ldr,
                                act, (bf16_t*)C, ldc, M, N, as_stream(stream));
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
// The same function with the row held in registers (K <= TPR * 8 * QC): one read of x instead of two, and for short rows
// (K <= 2048: the CLIP tower's 1024) ONE WAVE per row, four rows per workgroup, no LDS, no barrier — the kernel above runs
// a 256-thread workgroup with two block reductions per row, half of it idle at K = 1024 (416 -> 2xx us on 411200 x 1024).
template <int TPR, int QC>
__global__ __launch_bounds__(256) void quantize_fp8_rowwise_reg_kernel(const bf16_t* __restrict__ x, unsigned char* __restrict__ out,
                                                                       float* __restrict__ inv_scale, long rows, int K, long ldx,
                                                                       long ldo) {
  __shared__ float red[4];
  constexpr int RPB = 256 / TPR;
  const int t = threadIdx.x % TPR;
  const long row = (long)blockIdx.x * RPB + threadIdx.x / TPR;
  const bool active = row < rows;
  const bf16_t* xr = x + (active ? row : 0) * ldx;
  u32x4 v[QC];
  float mx = 0.f;
#pragma unroll
  for (int c = 0; c < QC; ++c) {
    const int i = (c * TPR + t) * 8;
    v[c] = u32x4{0u, 0u, 0u, 0u};
    if (i < K) {
      v[c] = *reinterpret_cast<const u32x4*>(xr + i);
#pragma unroll
      for (int e = 0; e < 4; ++e) mx = fmaxf(mx, fmaxf(fabsf(bf_lo(v[c][e])), fabsf(bf_hi(v[c][e]))));
    }
  }
  if constexpr (TPR == 64) mx = wave_max_dpp(mx); else mx = block_max<4>(mx, red);
  if (!active) return;
  const float amax = fmaxf(mx, 1e-12f);
  const float sc = 448.f / amax;
  if (t == 0) inv_scale[row] = amax / 448.f;
  unsigned char* orow = out + row * ldo;
#pragma unroll
  for (int c = 0; c < QC; ++c) {
    const int i = (c * TPR + t) * 8;
    if (i < K) {
      float f[8];
#pragma unroll
      for (int e = 0; e < 4; ++e) { f[2 * e] = bf_lo(v[c][e]) * sc; f[2 * e + 1] = bf_hi(v[c][e]) * sc; }
#pragma unroll
      for (int e = 0; e < 8; ++e) f[e] = __builtin_amdgcn_fmed3f(f[e], -448.f, 448.f);
      unsigned lo = 0, hi = 0;
      lo = __builtin_amdgcn_cvt_pk_fp8_f32(f[0], f[1], lo, false);
      lo = __builtin_amdgcn_cvt_pk_fp8_f32(f[2], f[3], lo, true);
      hi = __builtin_amdgcn_cvt_pk_fp8_f32(f[4], f[5], hi, false);
      hi = __builtin_amdgcn_cvt_pk_fp8_f32(f[6], f[7], hi, true);
      const u32x2 o = {lo, hi};
      *reinterpret_cast<u32x2*>(orow + i) = o;
    }
  }
}
}  // namespace
}  // namespace hvlm

extern "C" int hvlm_quantize_fp8_rowwise(const void* x, void* out, float* inv_scale, int rows, int K, int ldx, int ldo,
                                         void* stream) {
  HVLM_REQUIRE(x && out && inv_scale && rows > 0 && K > 0 && K % 8 == 0 && ldx % 8 == 0 && ldo % 8 == 0 && ldx >= K && ldo >= K,
               "quantize_fp8_rowwise: bad args (K, ldx, ldo multiples of 8)");
  HVLM_REQUIRE(((uintptr_t)x & 15) == 0 && ((uintptr_t)out & 7) == 0, "quantize_fp8_rowwise: alignment");
  if (K <= 64 * 8 * 4)
    hipLaunchKernelGGL((quantize_fp8_rowwise_reg_kernel<64, 4>), dim3(cdiv(rows, 4)), dim3(256), 0, as_stream(stream), (const bf16_t*)x,
                       (unsigned char*)out, inv_scale, (long)rows, K, (long)ldx, (long)ldo);
  else if (K <= 256 * 8 * 2)
    hipLaunchKernelGGL((quantize_fp8_rowwise_reg_kernel<256, 2>), dim3(rows), dim3(256), 0, as_stream(stream), (const bf16_t*)x,
                       (unsigned char*)out, inv_scale, (long)rows, K, (long)ldx, (long)ldo);
  else if (K <= 256 * 8 * 7)          // up to 14336: the SwiGLU output of the 7B (11008) and 13B (13824) decoders
    hipLaunchKernelGGL((quantize_fp8_rowwise_reg_kernel<256, 7>), dim3(rows), dim3(256), 0, as_stream(stream), (const bf16_t*)x,
                       (unsigned char*)out, inv_scale, (long)rows, K, (long)ldx, (long)ldo);
  else
    hipLaunchKernelGGL(quantize_fp8_rowwise_kernel, dim3(rows), dim3(256), 0, as_stream(stream), (const bf16_t*)x,
                       (unsigned char*)out, inv_scale, K, (long)ldx, (long)ldo);
  return check_launch("quantize_fp8_rowwise");
}

static int gemm_fp8_impl(const void* A, const void* B, void* C, int M, int N, int K, int lda, int ldb, int ldc,
                         float alpha, const float* row_scale, const float* col_scale, const void* bias,
                         const void* residual, int ldr, int act, int out_f32, void* stream, const float* out_inv = nullptr);

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

extern "C" int hvlm_gemm_fp8_nt_scaled_q8(const void* A, const void* B, void* C8, int M, int N, int K, int lda, int ldb, int ldc,
                                          float alpha, const float* row_scale, const float* col_scale, const void* bias,
                                          int act, const float* out_inv, void* stream) {
  HVLM_REQUIRE(out_inv, "gemm_fp8_q8: null out_inv");
  HVLM_REQUIRE(ldc % 16 == 0, "gemm_fp8_q8: ldc must be a multiple of 16");
  return gemm_fp8_impl(A, B, C8, M, N, K, lda, ldb, ldc, alpha, row_scale, col_scale, bias, nullptr, 0, act, 0, stream, out_inv);
}

static int gemm_fp8_impl(const void* A, const void* B, void* C, int M, int N, int K, int lda, int ldb, int ldc,
                         float alpha, const float* row_scale, const float* col_scale, const void* bias,
                         const void* residual, int ldr, int act, int out_f32, void* stream, const float* out_inv) {
  HVLM_REQUIRE(A && B && C, "gemm_fp8: null operand");
  HVLM_REQUIRE(!out_inv || (!residual && !out_f32), "gemm_fp8: e4m3 output takes no residual and is not fp32");
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
  p.splits = 1; p.k_split = K; p.slab = 0; p.alpha = alpha; p.dyn = 0; p.dbg = 0; p.epi_swap = config().gemm_epi_swap; p.tail_half = config().gemm_tail_half; p.sched = pp_sched_ws();
  p.row_scale = row_scale; p.col_scale = col_scale; p.out_inv = out_inv;
  static int cus = 0;
  if (!cus) {
    int dev = 0; hipDeviceProp_t prop;
    cus = (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess) ? prop.multiProcessorCount : 256;
    cus = (cus / 8) * 8;
  }
  int grid = p.tiles_m * p.tiles_n;
  if (grid > cus) grid = cus;
  p.dyn = (K / 128 >= 3 && cus <= 512 && p.tiles_m * p.tiles_n >= 3 * cus && !config().gemm_static) ? 1 + pp_sched_slot(as_stream(stream)) : 0;
  const int rc = launch_gemm_pp_fp8(p, grid, as_stream(stream));
  if (rc) return rc;
  return check_launch("gemm_fp8_nt");
}
