/*
 * hvlm.h — C ABI of libhvlm_hip.so: the MI355X (gfx950 / CDNA4) kernels behind the HandsOnVLM hot
 * path (CLIP-ViT-L/14 -> mm_projector -> temporal/spatial pooling -> multimodal splice -> Vicuna
 * decoder -> lm_head / cross-entropy; hand-trajectory gather), forward and backward.
 *
 * The reference (Kami-code/HandsOnVLM-release) is pure Python and has no FFI: its only
 * kernel-swap precedent is llava/train/llama_flash_attn_monkey_patch.py:114-124 (assigning
 * LlamaAttention.forward at import time).  Each entry point below therefore cites the reference
 * call site (or the third-party transformers==4.31.0 function it calls, setup.py:15) whose
 * arithmetic it replaces.  All paths are relative to the reference tree.
 *
 * Conventions
 *   - extern "C", plain pointers and sizes; no torch / STL types cross this boundary.
 *   - every pointer is a DEVICE pointer owned by the caller; the library never allocates, frees
 *     or retains memory; workspaces are passed in explicitly.
 *   - `stream` is a hipStream_t passed as void*; all work is enqueued on it, never synchronised.
 *   - return value: HVLM_OK (0) or a negative HVLM_ERR_*; hvlm_last_error() gives the message of
 *     the most recent failure on the calling thread.  Nothing throws across the ABI.
 *   - "bf16" buffers are raw uint16 bfloat16; row-major; `ld*` are leading dimensions in ELEMENTS.
 *   - tokens are stored token-major: activations are [M, features] with M = B*S (batch-major,
 *     right-padded to S per sample) — the layout of inputs_embeds at handsonvlm.py:416-450.
 */
#ifndef HVLM_H
#define HVLM_H
#include <stddef.h>
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

#define HVLM_OK 0
#define HVLM_ERR_INVALID (-1)
#define HVLM_ERR_LAUNCH (-2)
#define HVLM_ERR_UNSUPPORTED (-3)

#define HVLM_ACT_NONE 0
#define HVLM_ACT_QUICK_GELU 1 /* x*sigmoid(1.702x): transformers CLIP `quick_gelu` (clip_encoder.py:48) */

/* Bumped whenever an exported symbol is added, removed or changes meaning, so that a host built against an older header
 * fails its version check instead of binding a different function set.
 *   1: rounds 1-3.
 *   2: round 4 (hvlm_gemm_set_sched_ws replaced by hvlm_create / hvlm_destroy, hvlm_debug_occupy left production builds,
 *      hvlm_patch_im2col_f16 and hvlm_temporal_spatial_pool_fwd_f16 added) and round 5 (library switch "attn_fwd_asm"). */
#define HVLM_ABI_VERSION 2

int hvlm_abi_version(void);
const char* hvlm_last_error(void);
/* Fills CU count, LDS bytes per workgroup and the gcnArchName of `device`. */
int hvlm_device_info(int device, int* cu_count, int* lds_bytes, char* arch, int arch_len);
#ifdef HVLM_DEV
/* Development builds only (`make DEV=1`): keeps `workgroups` CUs busy (64 KiB LDS each) for about `cycles` shader clocks on
 * `stream` — a stand-in for a collective running beside the compute stream (tools/contention_bench.py). */
int hvlm_debug_occupy(int workgroups, long cycles, void* stream);
/* Development builds only: the hand-scheduled dh = 128 causal forward (hvlm_attn_fwd's default path, csrc/asm/attn_fwd_d128.py) in
 * its cycle-stamped form (variant 1) or as a timing-only ablation (variant >= 2: WRONG results — no softmax stream, no DMA, no
 * fragment reloads, no MFMAs, ...; tools/attn_asm_prof.py lists them).  prof: NULL or a device buffer of
 * min(CUs, B * H * ceil(S / 256)) workgroups x 4 waves x 8 uint32: cycle sums of the item phases {start wait, prologue + step 0,
 * loop, epilogue, item switch}.  ws as for hvlm_attn_fwd.  Padded batch only. */
int hvlm_attn_fwd_asm_dev(const void* qkv, void* out, float* lse, void* ws, void* prof, int B, int S, int H, int ld, float scale,
                          int variant, void* stream);
/* The same for the CLIP tower's attention (csrc/asm/attn_clip_d64.py: fp16, S = 257, dh = 64, non-causal; qkv [B * 257, ld]). */
int hvlm_attn_clip_asm_dev(const void* qkv, void* out, void* prof, int B, int H, int ld, float scale, int variant, void* stream);
#endif

/* Per-device handle — the ONLY mutable state the library keeps between calls: which block of device memory
 * holds the scheduler state of the persistent GEMM kernels (per-XCD tile counters + per-workgroup hand-over slots of the
 * dynamic tile scheduling, one slot per launching stream; hvlm_gemm_sched_ws_bytes() = 75 KiB).
 *   hvlm_create(device, sched_ws, bytes, stream, &h)  `device` must be the calling thread's current HIP device.  sched_ws: a
 *       caller-owned block of >= hvlm_gemm_sched_ws_bytes() bytes, 256-byte aligned, zero-filled by this call on `stream` and
 *       kept alive by the caller until hvlm_destroy; or NULL (bytes 0) for the library's built-in block (a HIP module global,
 *       one copy per device).  One handle per device: a second hvlm_create for a device that has one fails (HVLM_ERR_INVALID).
 *   hvlm_destroy(h)   the device goes back to the built-in block; the caller's block is no longer touched by later launches.
 * Every entry point that launches the persistent GEMM (hvlm_gemm_*_nt*, the lock-step decode GEMMs excepted) reads the handle
 * of the CURRENT device at launch; without a handle it uses the built-in block.  Kernels of different streams use different
 * slots of the block, so launches may overlap across streams.  Not thread-safe against launches in flight on that device. */
typedef struct hvlm_handle hvlm_handle;
int hvlm_create(int device, void* sched_ws, size_t sched_ws_bytes, void* stream, hvlm_handle** out);
int hvlm_destroy(hvlm_handle* handle);
int hvlm_handle_device(const hvlm_handle* handle);
size_t hvlm_gemm_sched_ws_bytes(void);

/* Library switches.  Every A/B switch of the library is one named integer: initialised ONCE, at the library's first use, from
 * the environment variable HVLM_<KEY IN CAPITALS>, changed at run time through hvlm_configure; no entry point reads the
 * environment.  Production switches: "gemm_group_m" (row panels per group of the XCD-aware tile order, default 4),
 * "gemm_static" (1: static tile assignment), "attn_no_resident" (1: CLIP attention on the streaming kernel),
 * "decode_unfused" (1: RoPE / split combine of the native decode step as separate launches), "decode_attn_v1" (1: the decode step's
 * attention on the one-key-per-thread kernel of round 3 instead of the lane-group kernel), "decode_attn_ns" (cap on its key splits, 0 =
 * automatic), "gemv_early" (default 1: the decode GEMVs request x, then their first weight block, and run the RMSNorm prologue while the block
 * is in flight; 0: the round-3 kernel), "decode_engine" (1: the whole batch-1 decode step as ONE persistent launch with grid barriers —
 * bit-identical results, measured slower than the 5-launches-per-layer sequence, default 0), "skinny_per" (forced K-slice of
 * the weight-streaming GEMM), "attn_delta_separate" (1: rowsum(O * dO) of the attention backward by its own pass).  Switches that select a superseded kernel variant or the K-loop ablation kernel ("gemm_dbg",
 * "gemm_seg4", "gemm_m16", "gemm_tn_seg4", "gemm_fp8_m16", "gemm_epi_swap", "gemm_tail_half", "attn_dkv_fused", "attn_fwd_pipe") exist in development builds only
 * (`make DEV=1`; hvlm_is_dev_build() == 1): a production build returns HVLM_ERR_UNSUPPORTED for any non-default value, and
 * does not contain those kernels.  Not thread-safe against concurrent launches (set switches before issuing work). */
int hvlm_configure(const char* key, int value);
int hvlm_config_get(const char* key, int* value);
int hvlm_is_dev_build(void);

/* ------------------------------------------------------------------------------------------ *
 * GEMM (MFMA, bf16 in, fp32 accumulate)
 *   C[M,N] = act(A[M,K] @ B[N,K]^T + bias[N]) + residual[M,N]
 * Replaces every nn.Linear on the path: LlamaAttention q/k/v/o_proj and LlamaMLP gate/up/down
 * (transformers modeling_llama, called from handsonvlm.py:119-128), lm_head (handsonvlm.py:48,132),
 * mm_projector (llava/model/llava_arch.py:33, applied at hoi_forecast/model/visual_to_tokens.py:279)
 * and the CLIP q/k/v/out_proj/fc1/fc2 + patch-embed conv (clip_encoder.py:48).  Backward GEMMs
 * (dX = dY @ W, dW = dY^T @ X) use the same entry point on transposed operands.
 *   K % 64 == 0; lda/ldb/ldc/ldr in elements, multiples of 8; A,B,C 16-byte aligned.
 *   bias (bf16[N]) / residual (bf16[M,ldr]) may be NULL.  out_f32 != 0 -> C is float32.
 * ------------------------------------------------------------------------------------------ */
int hvlm_gemm_bf16_nt(const void* A, const void* B, void* C, int M, int N, int K, int lda,
                      int ldb, int ldc, const void* bias, const void* residual, int ldr, int act,
                      int out_f32, void* stream);

/* Split-K variant for skinny problems (few 256x256 output tiles, e.g. the batch-1 prefill, M = 867): `splits` K-slices
 * are computed by separate workgroups into fp32 slabs of `ws` (hvlm_gemm_splitk_ws_bytes) and summed, deterministically,
 * by a finalize pass that applies bias / activation / residual and writes bf16 C.  K % (64*splits) == 0, N % 4 == 0. */
size_t hvlm_gemm_splitk_ws_bytes(int M, int N, int splits);
int hvlm_gemm_bf16_nt_splitk(const void* A, const void* B, void* C, int M, int N, int K, int lda, int ldb,
                             int ldc, const void* bias, const void* residual, int ldr, int act, void* ws,
                             int splits, void* stream);

/* Weight-streaming variant for M <= 32 (lock-step generation of a batch: every projection of a decode step is
 * x[B, K] @ W[N, K]^T and costs one read of W).  W goes from HBM into registers in coalesced order and is permuted into
 * MFMA fragments there, x is the LDS-resident operand, K is split over workgroups (hvlm_gemm_skinny_splits slices, fp32
 * slabs in `ws` of hvlm_gemm_skinny_ws_bytes) and the split-K finalize pass applies bias / activation / residual.
 * K % 32 == 0, N % 4 == 0, deterministic.  Same function as hvlm_gemm_bf16_nt on the same arguments (reference: the
 * nn.Linear calls of LlamaDecoderLayer inside generate(), handsonvlm_inference.py:99-113, at batch 32 = BASELINE config 5). */
int hvlm_gemm_skinny_splits(int N, int K);
size_t hvlm_gemm_skinny_ws_bytes(int M, int N, int K);
int hvlm_gemm_bf16_nt_skinny(const void* A, const void* B, void* C, int M, int N, int K, int lda, int ldb,
                             int ldc, const void* bias, const void* residual, int ldr, int act, void* ws,
                             void* stream);

/* The same kernel on e4m3 WEIGHTS with bf16 activations (weight-only quantisation for the decode steps of the fp8 evaluation,
 * BASELINE config 5): B8 is [N, K] e4m3 bytes (hvlm_quantize_fp8_rowwise of W), col_scale[n] its per-output-channel inverse
 * scale (W[n, :] ~= col_scale[n] * e4m3[n, :]); the bytes are converted to bf16 in registers (exact), the MFMA and x are
 * bf16, col_scale is applied by the finalize pass before bias / activation / residual.  Half the HBM bytes of the bf16
 * kernel, no activation quantisation pass.  K % 64 == 0, ldb % 16 == 0. */
int hvlm_gemm_w8a16_skinny_splits(int N, int K);
size_t hvlm_gemm_w8a16_skinny_ws_bytes(int M, int N, int K);
int hvlm_gemm_w8a16_nt_skinny(const void* A, const void* B8, const float* col_scale, void* C, int M, int N, int K,
                              int lda, int ldb, int ldc, const void* bias, const void* residual, int ldr, int act,
                              void* ws, void* stream);

/* The two projections of the lock-step decode step whose consumer rides in the finalize pass (bit-identical to the plain
 * calls followed by hvlm_swiglu_fwd / hvlm_rope_qk + the cache row write; three launches fewer per layer).  B is bf16 [N, K]
 * when col_scale == NULL, e4m3 [N, K] with per-output-channel scales otherwise (the W8A16 form); ws as for the plain calls
 * with N = 2F / 3 H dh.
 *   swiglu: act[M, F] = SwiGLU(x W_gate^T, x W_up^T), B = [gate; up] rows (LlamaMLP).
 *   qkv   : qkv[m, :] = RoPE(x W_qkv^T) at position lens[m] (q and k heads; v passes through), and its k | v columns are also
 *           written to cache[m, lens[m], :] of the [M, max_len, 3 H dh] cache (LlamaAttention with past_key_value). */
int hvlm_gemm_skinny_swiglu(const void* A, const void* B, const float* col_scale, void* act, int M, int F, int K, int lda,
                            int ldb, void* ws, void* stream);
int hvlm_gemm_skinny_qkv(const void* A, const void* B, const float* col_scale, void* qkv, int ldq, void* cache,
                         long max_len, const int32_t* lens, const void* cos_tab, const void* sin_tab, int M, int H,
                         int dh, int K, int lda, int ldb, void* ws, void* stream);

/* fp8 (OCP e4m3) operands on the CDNA4 scaled MFMA (v_mfma_scale_f32_32x32x64_f8f6f4, block scales fixed to 1):
 *   C[M,N] = act(alpha * A[M,K] @ B[N,K]^T + bias) + residual, A / B one byte per element, K % 128 == 0,
 *   lda / ldb % 16 == 0; alpha = 1 / (scale_a * scale_b) for per-tensor scaled operands (BASELINE config 5: fp8 forward).
 * hvlm_quantize_fp8: out[i] = e4m3(clamp(x[i] * scale, +-448)), scale from *scale_dev when non-NULL (no host sync)
 * else from `scale`; n % 8 == 0.  Reference arithmetic replaced: none in the reference (it has no fp8 path); the
 * parity test compares against the same product computed in fp32 from the de-quantised operands. */
int hvlm_quantize_fp8(const void* x, void* out, const float* scale_dev, float scale, long n, void* stream);
int hvlm_gemm_fp8_nt(const void* A, const void* B, void* C, int M, int N, int K, int lda, int ldb, int ldc,
                     float alpha, const void* bias, const void* residual, int ldr, int act, int out_f32,
                     void* stream);

/* Finer-grained scaling for the fp8 forward (BASELINE config 5): per-TOKEN activation scales and per-OUTPUT-CHANNEL weight
 * scales instead of one scale per tensor.
 *   hvlm_quantize_fp8_rowwise: for every row r of x [rows, K] (bf16, ldx): amax_r = max |x[r, :]|,
 *       out[r, :] = e4m3(x[r, :] * 448 / amax_r)  (ldo bytes per row), inv_scale[r] = amax_r / 448.
 *       Used on activations (rows = tokens) at run time and on weights (rows = output channels) once.
 *   hvlm_gemm_fp8_nt_scaled: C[m, n] = act(alpha * row_scale[m] * col_scale[n] * sum_k A[m,k] B[n,k] + bias) + residual;
 *       row_scale (f32[M]) / col_scale (f32[N]) may be NULL (= 1).  With A, B from the row-wise quantiser:
 *       row_scale = inv_scale of A, col_scale = inv_scale of B, alpha = 1.
 * Reference arithmetic replaced: none (the reference has no fp8 path); parity = the fp64 product of the de-quantised
 * operands (exact up to fp32 accumulation), accuracy = against the bf16 forward (tests/test_gpu_fp8.py). */
int hvlm_quantize_fp8_rowwise(const void* x, void* out, float* inv_scale, int rows, int K, int ldx, int ldo, void* stream);
int hvlm_gemm_fp8_nt_scaled(const void* A, const void* B, void* C, int M, int N, int K, int lda, int ldb, int ldc,
                            float alpha, const float* row_scale, const float* col_scale, const void* bias,
                            const void* residual, int ldr, int act, int out_f32, void* stream);
/* hvlm_gemm_fp8_nt_scaled whose OUTPUT is e4m3 as well: C8[m, n] = e4m3(clamp(bf16(act(alpha * rs[m] * cs[n] * A B^T + bias)) /
 * out_inv[m], +-448)), one byte per element, ldc in bytes — the A operand of the next GEMM with row_scale = out_inv.  The scale
 * has to be known before the row is complete, so the caller supplies an upper BOUND of the row's |output| (a floating-point
 * format loses nothing to a scale that is a few binades high); used for fc1 -> fc2 of the CLIP tower, where
 * |QuickGELU(y W^T + b)| <= |y|_2 max_n |W_n|_2 + max|b| (hvlm_layernorm_fwd_fp8 produces that bound).  No residual. */
int hvlm_gemm_fp8_nt_scaled_q8(const void* A, const void* B, void* C8, int M, int N, int K, int lda, int ldb, int ldc,
                               float alpha, const float* row_scale, const float* col_scale, const void* bias,
                               int act, const float* out_inv, void* stream);

/* General form with K-strided operands (no transposed copies for the backward GEMMs):
 *   trans_a: A is stored [K, M] (M contiguous, lda >= M); trans_b: B is stored [K, N] (ldb >= N).
 *   dX = dY @ W  -> trans_b with B = W as stored;  dW = dY^T @ X -> trans_a + trans_b with A = dY, B = X as stored.
 * Operand tiles are fetched through buffer descriptors (rows past the end read as zero): when both operands are
 * K-strided K may be any positive value; otherwise K % 64 == 0.  Operands must be < 4 GiB each.
 * trans_a = trans_b = 0 forwards to hvlm_gemm_bf16_nt. */
int hvlm_gemm_bf16(const void* A, const void* B, void* C, int M, int N, int K, int lda, int ldb, int ldc,
                   const void* bias, const void* residual, int ldr, int act, int out_f32, int trans_a,
                   int trans_b, void* stream);

/* out[C,R] = in[R,C]^T (bf16), LDS-tiled.  Rows r >= R_valid of `in` are treated as zero when
 * `R_pad` > R (out has leading dimension R_pad and its columns R..R_pad-1 are zero-filled): this
 * builds the K-padded transposed operands for the dW GEMMs. */
int hvlm_transpose_bf16(const void* in, void* out, int R, int C, int ld_in, int R_pad, void* stream);

/* out[C] (f32) = sum over rows of in[R,C] (bf16): bias gradients (mm_projector.bias). */
int hvlm_colsum_bf16(const void* in, float* out, int R, int C, int ld_in, void* stream);

/* ------------------------------------------------------------------------------------------ *
 * RMSNorm — transformers LlamaRMSNorm.forward (x.float(), variance, rsqrt(var+eps), cast back,
 * times weight), used (2L+1) times per LlamaModel.forward (handsonvlm.py:119).
 *   fwd: y = bf16(bf16(x * rstd) * w), rstd[rows] (f32) saved when non-NULL.
 *   bwd: dx = (add ? add : 0) + d(rmsnorm)/dx ; dw_partial[HVLM_NORM_BWD_BLOCKS, D] f32 partial
 *        sums of dy * xhat, reduced by hvlm_reduce_rows_f32.
 * ------------------------------------------------------------------------------------------ */
#define HVLM_NORM_BWD_BLOCKS 1024
int hvlm_rmsnorm_fwd(const void* x, const void* w, void* y, float* rstd, int rows, int D, float eps,
                     void* stream);
int hvlm_rmsnorm_bwd(const void* dy, const void* x, const void* w, const float* rstd,
                     const void* add, void* dx, float* dw_partial, int rows, int D, void* stream);
/* out[D] = sum_r in[r, D]  (f32 -> bf16 when out_bf16 else f32); optional accumulate into out. */
int hvlm_reduce_rows_f32(const float* in, void* out, int rows, int D, int out_bf16, void* stream);

/* LayerNorm forward (CLIP pre_layrnorm / layer_norm1 / layer_norm2, eps 1e-5; clip_encoder.py:48
 * -> transformers CLIPEncoderLayer).  y = bf16((x-mean)*rstd*w + b), fp32 statistics. */
int hvlm_layernorm_fwd(const void* x, const void* w, const void* b, void* y, int rows, int D,
                       float eps, void* stream);

/* ------------------------------------------------------------------------------------------ *
 * RoPE — transformers apply_rotary_pos_emb / rotate_half (half-split, not interleaved), cited at
 * llava/train/llama_flash_attn_monkey_patch.py:55-58.  In place on the q and k sections of the
 * fused qkv buffer [M, 3*H*dh] (q at column 0, k at H*dh).  cos/sin: bf16 tables [max_pos, dh/2]
 * (the reference casts the fp32 tables to the activation dtype); pos: int32[M].
 * inverse != 0 applies the transposed rotation (backward pass).
 * ------------------------------------------------------------------------------------------ */
int hvlm_rope_qk(void* qkv, const void* cos_tab, const void* sin_tab, const int32_t* pos, int M,
                 int H, int dh, int ld, int inverse, void* stream);

/* ------------------------------------------------------------------------------------------ *
 * Attention (flash-style, MFMA, online softmax in fp32) on the fused qkv buffer.
 * Replaces LlamaAttention's softmax(QK^T/sqrt(dh)+mask)V (eager) and the flash-attn call at
 * llama_flash_attn_monkey_patch.py:81-94 (causal; key padding is at the right end so the causal
 * mask alone decides, handsonvlm.py:441), and CLIPAttention (non-causal, dh=64).
 *   qkv: bf16 [B*S, ld]  q at col 0, k at col H*dh, v at col 2*H*dh (per head dh contiguous)
 *   out: bf16 [B*S, H*dh]; lse: f32 [B, H, S] (log-sum-exp of scaled scores; NULL to skip)
 *   dh in {64, 128}.
 *   ws: caller-provided scratch.  HARD CONTRACT (ABI version >= 2): at least hvlm_attn_{fwd,bwd}_ws_bytes() bytes — the
 *       forward's is 1024, of which the persistent kernel uses the first 512 — and 64-byte aligned.  The backward keeps delta = rowsum(dO*O) there, followed by the eight per-XCD
 *       entry counters of the persistent dK/dV kernel; the forward keeps the eight work-queue counters of its persistent
 *       dh = 128 causal kernel there and WRITES all 512 bytes (both zeroed by the call itself with a stream-ordered memset: no
 *       initialisation is asked of the caller).  The library can only check the alignment: a forward workspace that is NULL or
 *       not 64-byte aligned selects the non-persistent kernel (which needs no workspace) instead of failing; a workspace that
 *       is aligned but SHORTER than hvlm_attn_fwd_ws_bytes() is overrun.  (The transposed operands V^T, K^T, Q^T, dO^T are
 *       gathered inside the kernels with LDS transpose reads.)
 * bwd: dqkv (same layout as qkv, fully written for rows < S) from dout [B*S, H*dh].
 * ------------------------------------------------------------------------------------------ */
size_t hvlm_attn_fwd_ws_bytes(int B, int S, int H, int dh);
size_t hvlm_attn_bwd_ws_bytes(int B, int S, int H, int dh);
int hvlm_attn_fwd(const void* qkv, void* out, float* lse, void* ws, int B, int S, int H, int dh,
                  int ld, int causal, float scale, void* stream);
int hvlm_attn_bwd(const void* qkv, const void* out, const void* dout, const float* lse, void* dqkv,
                  void* ws, int B, int S, int H, int dh, int ld, int causal, float scale,
                  void* stream);

/* Attention backward with the gradient of apply_rotary_pos_emb (transformers modeling_llama, called from LlamaAttention.forward;
 * flash path llava/train/llama_flash_attn_monkey_patch.py:48-52) folded into the dQ / dK epilogues: dqkv is the gradient with
 * respect to the UN-rotated q / k, bit-identical to hvlm_attn_bwd[_varlen] followed by hvlm_rope_qk(dqkv, ..., inverse = 1).
 *   cu_seqlens: int32 [B+1] for packed rows (then max_seqlen / total_tokens as in hvlm_attn_bwd_varlen) or NULL (padded batch).
 *   cos_tab / sin_tab: bf16 [max_pos, dh/2]; pos: int32 position of every token row. */
int hvlm_attn_bwd_rope(const void* qkv, const void* out, const void* dout, const float* lse, void* dqkv, void* ws,
                       const int32_t* cu_seqlens, int B, int max_seqlen, long total_tokens, int H, int dh, int ld,
                       int causal, float scale, const void* cos_tab, const void* sin_tab, const int32_t* pos,
                       void* stream);

/* Variable-length (packed) form — the contract of the reference's kernel swap, llava/train/llama_flash_attn_monkey_patch.py:
 * 81-96 (unpad_input -> flash_attn_varlen_qkvpacked_func(qkv, cu_seqlens, max_seqlen, causal=True) -> pad_input):
 *   qkv / out / dout / dqkv hold the tokens of all B samples PACKED end to end (total_tokens rows, no padding rows),
 *   cu_seqlens int32[B+1] (device) gives the row range [cu[b], cu[b+1]) of sample b, max_seqlen = the longest sample.
 *   lse is f32 [B, H, max_seqlen] (entries past a sample's length are not written), ws as for the padded entry points with
 *   S = max_seqlen.  The engine (hvlm/llama.py) keeps tokens packed through the whole decoder, so no unpad / pad copies exist. */
int hvlm_attn_fwd_varlen(const void* qkv, void* out, float* lse, void* ws, const int32_t* cu_seqlens, int B,
                         int max_seqlen, int H, int dh, int ld, int causal, float scale, void* stream);
int hvlm_attn_bwd_varlen(const void* qkv, const void* out, const void* dout, const float* lse, void* dqkv, void* ws,
                         const int32_t* cu_seqlens, int B, int max_seqlen, long total_tokens, int H, int dh, int ld,
                         int causal, float scale, void* stream);

/* Single-token decode (KV-cached generation, batch 1; the reference's sampling loop handsonvlm.py:543-647
 * re-runs the full prefix per token — a cache is exact because appended tokens' input embeddings never change).
 *   gemv: y[N] = W[N,K] @ x[K] (+ residual[N]); K % 8 == 0, K <= 32768.
 *   attn_decode: q [H*dh] of the new token (already RoPE'd) against caches k/v [S, ld] (head h at column h*dh);
 *                out [H*dh]. */
int hvlm_gemv_bf16(const void* W, const void* x, void* y, const void* residual, int N, int K, int ldw,
                   void* stream);
int hvlm_attn_decode(const void* q, const void* k_cache, const void* v_cache, void* out, int S, int H,
                     int dh, int ld, float scale, void* stream);
/* Batched single-token attention for lock-step generation of B samples (batch evaluation, BASELINE config 5): sample b has
 * query row q + b*ldq (already RoPE'd), cache slabs k_cache / v_cache + b*sample_stride ([max_S, ld] each, fused qkv layout)
 * and its own length lens[b] (device int32; keys 0..lens[b]-1 include the new token's).  out row b at out + b*ldo. */
int hvlm_attn_decode_batched(const void* q, int ldq, const void* k_cache, const void* v_cache, long sample_stride,
                             const int32_t* lens, void* out, int ldo, int B, int max_S, int H, int dh, int ld, float scale,
                             void* stream);
/* One decode step of the whole decoder stack from native code (replaces LlamaModel.forward for one new token at
 * position `pos`): pointer tables are HOST arrays [L] of DEVICE pointers (fused weights as in hvlm/llama.py;
 * kv_cache[l] is [max_len, cache_ld] in the fused qkv layout, row `pos` is written).  pos_dev points at a device
 * int32 equal to pos.  ws: hvlm_llama_decode_ws_bytes(D, F) bytes, ZERO-FILLED once before its first use (it holds the
 * tickets of the fused attention combine; every step leaves them at zero).  hidden_out: final-normed hidden [D]. */
size_t hvlm_llama_decode_ws_bytes(int D, int F);
int hvlm_llama_decode_step(const void* const* w_qkv, const void* const* w_o, const void* const* w_gu,
                           const void* const* w_down, const void* const* ln1, const void* const* ln2,
                           void* const* kv_cache, const void* norm_w, const void* cos_tab, const void* sin_tab,
                           const int32_t* pos_dev, const void* x_in, void* hidden_out, void* ws, int L, int D,
                           int F, int H, int dh, int pos, int cache_ld, float eps, void* stream);
/* The same step with the position kept ON THE DEVICE: every kernel that needs it (the qkv GEMV's output row, RoPE, the
 * attention length) reads pos_dev[0], and the last kernel advances it by one — the launch sequence does not depend on the
 * token, so the caller captures it ONCE in a hipGraph and replays it per generated token (hvlm/llama.py::decode_step;
 * the reference's sample() loop, handsonvlm.py:478-672, re-runs the whole model per token instead).  max_S = capacity of
 * the caches (bounds the attention's LDS score buffer, fixes the key split).  No host synchronisation, no allocation. */
int hvlm_llama_decode_step_dev(const void* const* w_qkv, const void* const* w_o, const void* const* w_gu,
                               const void* const* w_down, const void* const* ln1, const void* const* ln2,
                               void* const* kv_cache, const void* norm_w, const void* cos_tab, const void* sin_tab,
                               int32_t* pos_dev, const void* x_in, void* hidden_out, void* ws, int L, int D, int F,
                               int H, int dh, int max_S, int cache_ld, float eps, void* stream);

/* ------------------------------------------------------------------------------------------ *
 * SwiGLU — transformers LlamaMLP: down(silu(gate(x)) * up(x)).  gu is the fused [M, 2F] output of
 * the concatenated gate|up projection (gate in columns [0,F), up in [F,2F)).
 * ------------------------------------------------------------------------------------------ */
int hvlm_swiglu_fwd(const void* gu, void* act, int M, int F, void* stream);
/* SwiGLU with the row written as e4m3 bytes + one scale per row (== hvlm_quantize_fp8_rowwise of hvlm_swiglu_fwd's output, bit for
 * bit): the input of down_proj in the fp8 prefill (LlamaMLP.act_fn(gate) * up).  F % 8 == 0, F <= 14336. */
int hvlm_swiglu_fwd_fp8(const void* gu, void* act8, float* inv_scale, int M, int F, void* stream);
int hvlm_swiglu_bwd(const void* gu, const void* dact, void* dgu, int M, int F, void* stream);

/* The same two functions with a fused TRANSPOSED second output ([features, R_pad] with the token index contiguous, columns
 * M..R_pad-1 zero) — the operand layout of the weight-gradient GEMMs dW = dY^T X: actT feeds dW_down, dguT feeds dW_gate|up;
 * saves the read half of a separate hvlm_transpose_bf16 pass.  F % 64 == 0, R_pad % 64 == 0, R_pad >= M. */
int hvlm_swiglu_fwd_t(const void* gu, void* act, void* actT, int M, int F, int R_pad, void* stream);
int hvlm_swiglu_bwd_t(const void* gu, const void* dact, void* dgu, void* dguT, int M, int F, int R_pad, void* stream);

/* ------------------------------------------------------------------------------------------ *
 * Visual-token pipeline
 * ------------------------------------------------------------------------------------------ */
/* im2col for the CLIP patch-embed conv (kernel = stride = P, no bias): images bf16
 * [N,3,HW,HW] -> patches bf16 [N*(HW/P)^2, Kpad] (K = 3*P*P, zero-padded to Kpad). */
int hvlm_patch_im2col(const void* images, void* patches, int N, int HW, int P, int Kpad, void* stream);
/* The same with bf16 frames in and IEEE-half patches out: images.to(dtype=self.dtype) of clip_encoder.py:48 applied on the way
 * through (bit-identical to hvlm_cast_bf16_to_f16 followed by hvlm_patch_im2col).  P, HW, Kpad even. */
int hvlm_patch_im2col_f16(const void* images, void* patches, int N, int HW, int P, int Kpad, void* stream);
/* tokens[n, 0, :] = cls + pos[0];  tokens[n, 1+p, :] = patch_out[n*NP+p, :] + pos[1+p]
 * (transformers CLIPVisionEmbeddings). */
int hvlm_clip_embed(const void* patch_out, const void* cls, const void* pos, void* tokens, int N,
                    int NP, int D, void* stream);
/* VisualToTokenHelper.compress_tokens('temporal_spatial_pool')
 * (hoi_forecast/model/visual_to_tokens.py:252-272): tokens [B,T,G*G,D] -> out [B,T+G*G,D];
 * rows [0,T): mean over the G*G patches of each frame; rows [T,T+G*G): frames
 * round(linspace(0,T-1,4)) on their GxG grid, 2x2 average-pooled to (G/2)x(G/2).
 * grid = G = 16 is the reference geometry (224 px: it asserts 256 patches); 24 = the 336-px generalisation.
 * drop_cls != 0: the input has G*G+1 tokens per frame and token 0 (CLS) is skipped
 * (feature_select 'patch', clip_encoder.py:29-37). */
int hvlm_temporal_spatial_pool_fwd(const void* tokens, void* out, int B, int T, int D,
                                   int drop_cls, int grid, void* stream);
/* The same on IEEE-half tokens / output (fp32 accumulation): the pooling applied INSIDE the last layer of the fp16 CLIP tower
 * (config.pool_before_last_fc2: the means commute with CLIPMLP.fc2 and the residual add, transformers modeling_clip through
 * clip_encoder.py:39-51, so fc2 runs on T + G*G pooled rows per sample instead of T*(G*G+1)).
 * frame_index (int32 [B*T], may be NULL): frame (b, t) is token block frame_index[b*T + t] of `tokens` — the tower ran on the
 * DISTINCT frames only (config.dedup_frames: EK100 tiles 10 frames x10, epic_dataset.py:90-95) and the pooling gathers them. */
int hvlm_temporal_spatial_pool_fwd_f16(const void* tokens, void* out, int B, int T, int D,
                                       int drop_cls, int grid, const int32_t* frame_index, void* stream);
/* backward of the above: dtokens [B,T,G*G,D] (fully written) from dout [B,T+G*G,D]. */
int hvlm_temporal_spatial_pool_bwd(const void* dout, void* dtokens, int B, int T, int D, int grid, void* stream);

/* Multimodal splice (handsonvlm.py:212-451): builds inputs_embeds [B,S,D] where
 *   src[b*S+s] >= 0  : row = embed_tokens[src]            (text token id)
 *   src[b*S+s] == -1 : row = 0                             (right padding)
 *   src[b*S+s] <= -2 : row = visual[b, -2-src]             (visual token index)
 * and adds the <hand_traj> sinusoidal encoding (process_traj_positional_embedding,
 * handsonvlm.py:310-338) to rows with hand_slot[b*S+s] >= 0:
 *   + hand_scale[b] * posenc(hands[b, :, hand_slot, :])    (hands f32 [B,2,NH,2]).
 * The scatter semantics of handsonvlm.py:368-375 (missing hand tokens fall on row 0 of the text
 * chunk) are resolved on the host into (hand_slot, hand_scale). */
int hvlm_embed_splice_fwd(const void* embed, const void* visual, const int32_t* src,
                          const int32_t* hand_slot, const float* hands, const float* hand_scale,
                          void* out, int B, int S, int D, int NV, int NH, void* stream);
/* backward: scatter-add dX rows into dembed (f32 [V,D], atomics) and dvisual (bf16 [B,NV,D]). */
int hvlm_embed_splice_bwd(const void* dx, const int32_t* src, float* dembed, void* dvisual, int B,
                          int S, int D, int NV, void* stream);

/* Packed (unpadded) variants: `rows` = total tokens of the batch, no padding rows; row_sample int32[rows] names the sample
 * each row belongs to (it selects visual[b], hands[b], hand_scale[b]).  Same arithmetic as above. */
int hvlm_embed_splice_fwd_packed(const void* embed, const void* visual, const int32_t* src, const int32_t* hand_slot,
                                 const float* hands, const float* hand_scale, const int32_t* row_sample, void* out,
                                 int rows, int D, int NV, int NH, void* stream);
int hvlm_embed_splice_bwd_packed(const void* dx, const int32_t* src, const int32_t* row_sample, float* dembed,
                                 void* dvisual, int rows, int D, int NV, void* stream);

/* Deterministic form of the text-embedding gradient (the `embed_tokens` rows of the splice backward): the host sorts
 * the text rows of the batch by token id (stable) into segments — seg_token[n_seg] the distinct ids, seg_start[n_seg+1]
 * offsets into seg_rows[], seg_rows[] the row indices into dx [B*S, D] — and every segment is summed in that fixed order in
 * fp32 and written ONCE as bf16 into dembed [V, D] (rows of tokens that do not occur are left untouched: the caller zeroes
 * dembed).  No atomics: the SFT step is run-to-run bit-identical (hvlm_embed_splice_bwd with dembed = NULL then only
 * scatters the visual rows).  Reference: autograd of nn.Embedding at handsonvlm.py:245-300. */
int hvlm_embed_grad_segments(const void* dx, const int32_t* seg_rows, const int32_t* seg_start,
                             const int32_t* seg_token, void* dembed, int n_seg, int D, void* stream);

/* ------------------------------------------------------------------------------------------ *
 * Heads
 * ------------------------------------------------------------------------------------------ */
/* Shifted cross-entropy over bf16 logits [M, ldl] with V valid columns (handsonvlm.py:137-145:
 * CrossEntropyLoss(mean over labels != -100)).  targets int32[M] already shifted by the host
 * (target[m] = labels[m+1] within a sample, -100 at the last position / ignored).
 *   loss_sum[0] += sum of -log p[target];  loss_sum[1] += number of valid targets.
 *   when dlogits != NULL (may alias logits): dlogits = (softmax - onehot) * grad_scale[0] for
 *   valid rows, 0 otherwise, zero in the padding columns [V, ldl).  grad_scale is a device f32. */
int hvlm_ce_fwd(const void* logits, const int32_t* targets, float* loss_sum, float* row_lse, int M,
                int V, int ldl, void* stream);
int hvlm_ce_bwd(const void* logits, const int32_t* targets, const float* row_lse,
                const float* grad_scale, void* dlogits, int M, int V, int ldl, void* stream);

/* Hand-token gather (handsonvlm.py:146-187): out[b, hand, j, c] = hidden[row[b*4+j], 2*c + hand]
 * (reshape(4, D/2, 2).permute(2,0,1)); row < 0 -> zeros.  out is f32 or bf16 [B,2,4,D/2]. */
int hvlm_hand_gather_fwd(const void* hidden, const int32_t* rows, void* out, int B, int D,
                         int out_f32, void* stream);
/* backward: dhidden[row, 2*c+hand] += dout[b,hand,j,c] (rows are unique). dhidden is bf16 [M,D]. */
int hvlm_hand_gather_bwd(const void* dout, const int32_t* rows, void* dhidden, int B, int D,
                         int dout_f32, void* stream);

/* ------------------------------------------------------------------------------------------ *
 * fp16 instantiations for the CLIP tower.  The reference runs the frozen tower in IEEE half
 * (handsonvlm/train/train.py:316 `vision_tower.to(dtype=torch.float16)`, handsonvlm/model/builder.py:108) and casts the
 * frames to it and the features back (llava/model/multimodal_encoder/clip_encoder.py:44-48); these entry points are the
 * same kernels as their bf16 namesakes with 16-bit IEEE-half storage and v_mfma_f32_32x32x16_f16 (same MFMA rate):
 *   gemm_f16_nt (A, B, C, bias, residual all fp16), layernorm_fwd_f16, clip_embed_f16, attn_fwd_f16 (non-causal, dh = 64:
 *   CLIPAttention; lse / ws as in hvlm_attn_fwd), and the two 16-bit casts (n % 8 == 0).
 * ------------------------------------------------------------------------------------------ */
int hvlm_gemm_f16_nt(const void* A, const void* B, void* C, int M, int N, int K, int lda, int ldb, int ldc,
                     const void* bias, const void* residual, int ldr, int act, int out_f32, void* stream);
int hvlm_layernorm_fwd_f16(const void* x, const void* w, const void* b, void* y, int rows, int D, float eps, void* stream);
/* RMSNorm with an e4m3 row output (== hvlm_quantize_fp8_rowwise of hvlm_rmsnorm_fwd's output, bit for bit): the input of the
 * q|k|v and gate|up GEMMs of the decoder's fp8 prefill (LlamaRMSNorm input_layernorm / post_attention_layernorm).
 * D = 2048, 4096 (7B) or 5120 (13B). */
int hvlm_rmsnorm_fwd_fp8(const void* x, const void* w, void* out8, float* inv_scale, int rows, int D, float eps, void* stream);
/* LayerNorm (bf16 in, as hvlm_layernorm_fwd) whose output row is written as e4m3 bytes with one scale per row: out8[r, :] and
 * inv_scale[r] are bit-identical to hvlm_quantize_fp8_rowwise of hvlm_layernorm_fwd's bf16 output — the input of the q|k|v
 * and fc1 GEMMs of the CLIP tower in the fp8 forward (BASELINE config 5; CLIPEncoderLayer.layer_norm1 / layer_norm2).
 * next_inv (optional): next_inv[r] = (|y_r|_2 * bound_mul + bound_add) / 448 with y_r the bf16 output row — the output scale of
 * a GEMM that consumes this row and writes e4m3 itself (hvlm_gemm_fp8_nt_scaled_q8). */
int hvlm_layernorm_fwd_fp8(const void* x, const void* w, const void* b, void* out8, float* inv_scale, float* next_inv,
                           float bound_mul, float bound_add, int rows, int D, float eps, void* stream);
/* LayerNorm folded into the GEMMs around it (frozen CLIP tower, fp16): CLIPEncoderLayer computes layer_norm1 -> q|k|v and
 * layer_norm2 -> fc1 (transformers modeling_clip, called from clip_encoder.py:48).  Since LN(x) W^T + b =
 * rstd * (x (W (.) ln_w)^T - mean * c1) + (b + W ln_b) with c1[n] = sum_k (W (.) ln_w)[n, k], the normalised tensor never has to
 * exist: the GEMM that PRODUCES the residual stream x (out_proj / fc2, residual epilogue) also writes, per row, the partial
 * (sum, sum of squares) of its outputs as stored (`stats_out`: fp32 [M, hvlm_ln_stats_parts(N), 2], one writer per entry);
 * hvlm_ln_stats_finalize turns them into (mean, rstd) [M, 2]; the GEMM that CONSUMES x (q|k|v / fc1: B = W (.) ln_w, bias =
 * b + W ln_b prepared once by the host, `ln_c1` = c1 as fp32 [N], 16-byte aligned) applies them in its epilogue.  Differences from the reference's
 * arithmetic: the normalised row is not rounded to fp16 before the product (more exact), variance = E[x^2] - mean^2 in fp32.
 * stats_out requires a residual and N % 256 == 0; ln_stats / ln_c1 come together, take no residual, N % 256 == 0.
 * With all three NULL this is hvlm_gemm_f16_nt. */
size_t hvlm_ln_stats_parts(int N);
int hvlm_gemm_f16_nt_ln(const void* A, const void* B, void* C, int M, int N, int K, int lda, int ldb, int ldc,
                        const void* bias, const void* residual, int ldr, int act, float* stats_out, const float* ln_stats,
                        const float* ln_c1, void* stream);
int hvlm_ln_stats_finalize(const float* part, float* stats, int rows, int n_parts, int D, float eps, void* stream);
int hvlm_clip_embed_f16(const void* patch_out, const void* cls, const void* pos, void* tokens, int N, int NP, int D,
                        void* stream);
int hvlm_attn_fwd_f16(const void* qkv, void* out, float* lse, void* ws, int B, int S, int H, int dh, int ld, int causal,
                      float scale, void* stream);
int hvlm_cast_bf16_to_f16(const void* in, void* out, long n, void* stream);
int hvlm_cast_f16_to_bf16(const void* in, void* out, long n, void* stream);

/* ------------------------------------------------------------------------------------------ *
 * Optimizer / elementwise
 * ------------------------------------------------------------------------------------------ */
/* AdamW on fp32 master weights with bf16 gradients (scripts/finetune.sh:30-33: lr 2e-5, wd 0),
 * writing the bf16 working copy W[R,C] and, when wt != NULL, its transpose WT[C,R] (operand of
 * the dX GEMMs).  grad_scale multiplies the gradient (1/world for DP averaging, loss scaling).
 * step >= 1. */
int hvlm_adamw_step(float* master, float* m, float* v, const void* grad, void* w, void* wt, int R,
                    int C, float lr, float beta1, float beta2, float eps, float weight_decay,
                    float grad_scale, int step, void* stream);
/* Prefill -> KV cache of lock-step generation: the k | v columns of qkv [B * S, 3D] (fused q|k|v rows after RoPE) are copied to
 * cache [B, max_len, 3D] rows 0 .. S-1 of every sample (columns D .. 3D; the q columns of the cache are never read).
 * Replaces the `past_key_value` concatenation of LlamaAttention.forward under use_cache. */
int hvlm_kv_cache_fill(const void* qkv, void* cache, int B, int S, int max_len, int D, void* stream);
/* y = a + b (bf16, n elements). */
int hvlm_add_bf16(const void* a, const void* b, void* y, long n, void* stream);
/* f32 -> bf16 and bf16 -> f32 casts (n elements). */
int hvlm_cast_f32_to_bf16(const float* in, void* out, long n, void* stream);
int hvlm_cast_bf16_to_f32(const void* in, float* out, long n, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* HVLM_H */
