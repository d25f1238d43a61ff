// Argument block shared by the bf16 NT GEMM kernels (gemm.hip, gemm_pp.hip).
#pragma once
#include "common.h"

namespace hvlm {

struct GemmArgs {
  const bf16_t* A;
  const bf16_t* B;
  void* C;
  const bf16_t* bias;
  const bf16_t* res;
  int M, N, K, lda, ldb, ldc, ldr, act, out_f32, tiles_m, tiles_n, group_m;
  int splits, k_split;          // split-K: `splits` K-slices of k_split columns, slice s writes fp32 slab s of C
  size_t slab;                  // elements between fp32 slabs
  int dyn;                      // gemm_pp: 0 = static tile assignment, s+1 = dynamic scheduling on state slot s (pp_sched_slot)
  void* sched = nullptr;        // gemm_pp: caller-provided scheduler state (the sched_ws given to hvlm_create), NULL = the library's built-in state
  int dbg;                      // development ablations (HVLM_GEMM_DBG), 0 in production
  int epi_swap = 0;             // gemm_pp, development builds: 1 = register-exchange epilogue (v_permlane swaps)
  int tail_half = 0;            // gemm_pp, dynamic scheduling: 1 = an XCD chunk's last partial round of tiles runs as half-tile units
  float alpha;                  // fp8 kernel only: C = alpha * (A B^T) before the epilogue (product of the tensor scales)
  const float* row_scale = nullptr;   // fp8 kernel only, optional: C[m, :] *= row_scale[m]  (per-token activation de-scale)
  const float* col_scale = nullptr;   // fp8 kernel only, optional: C[:, n] *= col_scale[n]  (per-output-channel weight de-scale)
  // LayerNorm folded into the GEMMs of the frozen CLIP tower (fp16 kernel only; hvlm_gemm_f16_nt_ln):
  float* stats_out = nullptr;         //   producer (residual epilogue): partial (sum, sum of squares) of every OUTPUT row over this wave's 64
                                      //   columns, [M][tiles_n * 4][2] fp32, each entry written by exactly one wave (deterministic)
  const float* ln_stats = nullptr;    //   consumer: (mean, rstd) of every A row [M][2]; C = rstd * (A B^T - mean * ln_c1) + bias (B = W (.) ln_w)
  const float* ln_c1 = nullptr;       //   consumer: c1[n] = sum_k B[n, k] (fp32 [N])
  const float* out_inv = nullptr;     // fp8 kernel only, optional: C is e4m3 BYTES, C[m, n] = e4m3(clamp(bf16(v) / out_inv[m])) — the
                                      // next GEMM's A operand with row_scale = out_inv (no residual in this mode)
};

// gemm_pp.hip: 8-phase ping-pong kernel (persistent 256x256 tiles); grid = workgroups to launch.
int launch_gemm_pp(const GemmArgs& p, int grid, hipStream_t stream);
// Dynamic-scheduling state slot of `stream` (launches on one stream are serialised, so they can share a slot; kernels
// on different streams may overlap and must not): the first 8 distinct streams get their own slot, -1 after that.
int pp_sched_slot(hipStream_t stream);
// The scheduler-state block the caller installed for the CURRENT device (hvlm_create), or NULL.
void* pp_sched_ws();
// same kernel, both operands K-strided (A is [K, M], B is [K, N]; lda / ldb = row strides of those): dW = dY^T X
int launch_gemm_pp_tn(const GemmArgs& p, int grid, hipStream_t stream);
// same kernel on IEEE fp16 operands / output (v_mfma_f32_32x32x16_f16): the CLIP tower in the reference's dtype
int launch_gemm_pp_f16(const GemmArgs& p, int grid, hipStream_t stream);
// same kernel on OCP e4m3 operands (A, B point at bytes; lda / ldb in bytes = elements), C = alpha * A B^T (+ epilogue)
int launch_gemm_pp_fp8(const GemmArgs& p, int grid, hipStream_t stream);

}  // namespace hvlm
