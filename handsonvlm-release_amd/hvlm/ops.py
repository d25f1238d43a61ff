"""Tensor-level wrappers over the C ABI (include/hvlm.h).

PyTorch is used only for device memory and streams: every function takes CUDA(ROCm) tensors, passes
raw device pointers + sizes + the current HIP stream through ctypes, and returns torch tensors that
own the output memory.  No function here computes anything with torch kernels; if the library is
missing, `_lib.load()` raises (there is no fallback).
"""
from __future__ import annotations

import ctypes
import os
from typing import Optional

import torch

from . import _lib
from ._lib import check

ACT_NONE = 0
ACT_QUICK_GELU = 1
NORM_BWD_BLOCKS = 1024
IGNORE_INDEX = -100
BF16 = torch.bfloat16


_GEMM_CUS = 256                      # persistent GEMM grid (MI355X: 256 CUs); only used to decide the tail split below
_GEMM_TAIL = os.environ.get("HVLM_GEMM_TAIL", "1") == "1"
_GEMM_SKINNY = os.environ.get("HVLM_GEMM_SKINNY", "1") == "1"     # 0: M <= 32 goes through the 256 x 256 split-K path (A/B)
_GEMM_TAIL_MIN_K = int(os.environ.get("HVLM_GEMM_TAIL_MIN_K", "8192"))

# Optional instrumentation (bench.py): hook(M, N, K, True) before / hook(M, N, K, False[, n_kernels]) after each GEMM call.
_gemm_hook = None


def set_gemm_hook(hook):
    global _gemm_hook
    _gemm_hook = hook


# Optional instrumentation of the HBM-bound kernels (bench.py `hbm` object): hook(kernel_name, algorithmic_bytes, True) before /
# hook(kernel_name, algorithmic_bytes, False) after the launch, on the launch stream.
_hbm_hook = None


def set_hbm_hook(hook):
    global _hbm_hook
    _hbm_hook = hook


def _p(t: Optional[torch.Tensor]):
    return None if t is None else ctypes.c_void_p(t.data_ptr())


def _stream():
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def _chk(t: torch.Tensor, dtype, name: str):
    if not t.is_cuda:
        raise _lib.HvlmError(f"{name}: expected a GPU tensor (the HIP path has no CPU fallback)")
    if t.dtype != dtype:
        raise _lib.HvlmError(f"{name}: expected {dtype}, got {t.dtype}")


def device_info(device: int = 0):
    cu, lds = ctypes.c_int(0), ctypes.c_int(0)
    arch = ctypes.create_string_buffer(64)
    check(_lib.load().hvlm_device_info(device, ctypes.byref(cu), ctypes.byref(lds), arch, 64), "device_info")
    return {"cu_count": cu.value, "lds_bytes": lds.value, "arch": arch.value.decode()}


# ------------------------------------------------------------------------------------------------
# GEMM
# ------------------------------------------------------------------------------------------------
def gemm_nt(a: torch.Tensor, b: torch.Tensor, bias: Optional[torch.Tensor] = None,
            residual: Optional[torch.Tensor] = None, act: int = ACT_NONE,
            out: Optional[torch.Tensor] = None, out_dtype=BF16) -> torch.Tensor:
    """out[M,N] = act(a[M,K] @ b[N,K]^T + bias) + residual.  a/b may be row-strided 2-D views."""
    _chk(a, BF16, "gemm.a"); _chk(b, BF16, "gemm.b")
    assert a.dim() == 2 and b.dim() == 2 and a.shape[1] == b.shape[1], (a.shape, b.shape)
    assert a.stride(1) == 1 and b.stride(1) == 1
    M, K = a.shape
    N = b.shape[0]
    if out is None:
        out = torch.empty((M, N), dtype=out_dtype, device=a.device)
    assert out.shape == (M, N) and out.stride(1) == 1
    ldr = 0
    if residual is not None:
        _chk(residual, BF16, "gemm.residual")
        assert residual.shape == (M, N) and residual.stride(1) == 1
        ldr = residual.stride(0)
    if bias is not None:
        _chk(bias, BF16, "gemm.bias")
        assert bias.numel() == N and bias.is_contiguous()
    hook = _gemm_hook
    if hook is not None:
        hook(M, N, K, True)
    # M <= 32 (lock-step decode of a batch): the weight-streaming kernel, W straight from HBM into registers
    if M <= 32 and out.dtype == BF16 and N % 4 == 0 and K % 32 == 0 and K >= 256 and _GEMM_SKINNY:
        lib = _lib.load()
        ws = torch.empty((lib.hvlm_gemm_skinny_ws_bytes(M, N, K),), dtype=torch.uint8, device=a.device)
        check(lib.hvlm_gemm_bf16_nt_skinny(_p(a), _p(b), _p(out), M, N, K, a.stride(0), b.stride(0), out.stride(0),
                                           _p(bias), _p(residual), ldr, act, _p(ws), _stream()), "gemm_skinny")
        if hook is not None:
            hook(M, N, K, False)
        return out
    # skinny problems (fewer output tiles than half the CUs): split K across workgroups
    tiles = ((M + 255) // 256) * ((N + 255) // 256)
    if tiles <= 96 and out.dtype == BF16 and N % 4 == 0 and K >= 1024:
        splits = 1
        for cand in (8, 4, 2):
            if tiles * cand <= 288 and K % (64 * cand) == 0 and K // cand >= 512:
                splits = cand
                break
        if splits > 1:
            lib = _lib.load()
            ws = torch.empty((lib.hvlm_gemm_splitk_ws_bytes(M, N, splits),), dtype=torch.uint8, device=a.device)
            check(lib.hvlm_gemm_bf16_nt_splitk(_p(a), _p(b), _p(out), M, N, K, a.stride(0), b.stride(0), out.stride(0),
                                               _p(bias), _p(residual), ldr, act, _p(ws), splits, _stream()), "gemm_splitk")
            if hook is not None:
                hook(M, N, K, False)
            return out
    # Tail of the last partial round of 256x256 tiles (e.g. 880 tiles on 256 CUs = 3.44 rounds): when it fills at most half of
    # the chip and K is long enough to pay for two extra launches, the bottom row panels that make up that partial round run
    # as a split-K GEMM (its K-slices fill the idle CUs: 3 + 1/s tile-times instead of 4), the rest as exactly R full rounds.
    lib = _lib.load()
    tn = (N + 255) // 256
    if tiles > _GEMM_CUS and K >= _GEMM_TAIL_MIN_K and out.dtype == BF16 and N % 4 == 0 and _GEMM_TAIL:
        T = tiles % _GEMM_CUS
        if 0 < T <= _GEMM_CUS // 2 and T % tn == 0:
            rows_tail_tiles = T // tn
            m_body = ((M + 255) // 256 - rows_tail_tiles) * 256
            splits = 0
            for cand in (4, 3, 2):
                if T * cand <= _GEMM_CUS and K % (64 * cand) == 0:
                    splits = cand
                    break
            if splits and 0 < m_body < M:
                mt = M - m_body
                res_b = None if residual is None else residual[:m_body]
                res_t = None if residual is None else residual[m_body:]
                check(lib.hvlm_gemm_bf16_nt(_p(a[:m_body]), _p(b), _p(out[:m_body]), m_body, N, K, a.stride(0), b.stride(0),
                                            out.stride(0), _p(bias), _p(res_b), ldr, act, 0, _stream()), "gemm_bf16_nt")
                ws = torch.empty((lib.hvlm_gemm_splitk_ws_bytes(mt, N, splits),), dtype=torch.uint8, device=a.device)
                check(lib.hvlm_gemm_bf16_nt_splitk(_p(a[m_body:]), _p(b), _p(out[m_body:]), mt, N, K, a.stride(0), b.stride(0),
                                                   out.stride(0), _p(bias), _p(res_t), ldr, act, _p(ws), splits, _stream()),
                      "gemm_splitk(tail)")
                if hook is not None:
                    hook(M, N, K, False, 2)             # two launches of the GEMM kernel inside this interval
                return out
    # Few row panels (the batch-1 prefill: M = 867 -> 4 panels) and a tile count just above one round of the chip (gate|up: 4 x 86 = 344
    # tiles on 256 CUs = two rounds for 1.34 rounds of work): the leftmost column panels that make up exactly ONE round run as one launch and
    # the remaining columns as a split-K launch whose K-slices fill the chip — 1 + 1/s tile-times instead of 2 (round 6; the row-panel
    # form above needs K >= 8192 and many row panels).  Deterministic (fixed slab order), every output column computed once.
    tm = (M + 255) // 256
    if (_GEMM_TAIL and tm <= 8 and _GEMM_CUS < tiles < 2 * _GEMM_CUS and out.dtype == BF16 and N % 256 == 0 and K >= 2048):
        body_cols = _GEMM_CUS // tm                       # column panels of the body: tm * body_cols <= one round
        T = (tn - body_cols) * tm
        splits = 0
        for cand in (2, 3, 4):
            if T * cand <= _GEMM_CUS and K % (64 * cand) == 0 and K // cand >= 1024:
                splits = cand
        if splits and 0 < body_cols < tn:
            nb = body_cols * 256
            b_b, b_t = b[:nb], b[nb:]
            bias_b = None if bias is None else bias[:nb]
            bias_t = None if bias is None else bias[nb:]
            res_b = None if residual is None else residual[:, :nb]
            res_t = None if residual is None else residual[:, nb:]
            o_b, o_t = out[:, :nb], out[:, nb:]
            check(lib.hvlm_gemm_bf16_nt(_p(a), _p(b_b), _p(o_b), M, nb, K, a.stride(0), b.stride(0), out.stride(0), _p(bias_b),
                                        _p(res_b), ldr, act, 0, _stream()), "gemm_bf16_nt(body columns)")
            ws = torch.empty((lib.hvlm_gemm_splitk_ws_bytes(M, N - nb, splits),), dtype=torch.uint8, device=a.device)
            check(lib.hvlm_gemm_bf16_nt_splitk(_p(a), _p(b_t), _p(o_t), M, N - nb, K, a.stride(0), b.stride(0), out.stride(0),
                                               _p(bias_t), _p(res_t), ldr, act, _p(ws), splits, _stream()), "gemm_splitk(tail columns)")
            if hook is not None:
                hook(M, N, K, False, 2)
            return out
    check(lib.hvlm_gemm_bf16_nt(_p(a), _p(b), _p(out), M, N, K, a.stride(0), b.stride(0),
                                out.stride(0), _p(bias), _p(residual), ldr, act,
                                1 if out.dtype == torch.float32 else 0, _stream()), "gemm_bf16_nt")
    if hook is not None:
        hook(M, N, K, False)
    return out


def gemm(a: torch.Tensor, b: torch.Tensor, trans_a: bool = False, trans_b: bool = False,
         bias: Optional[torch.Tensor] = None, residual: Optional[torch.Tensor] = None, act: int = ACT_NONE,
         out: Optional[torch.Tensor] = None, out_dtype=BF16) -> torch.Tensor:
    """out[M,N] = act(opA(a) @ opB(b)^T + bias) + residual; trans_a: a is [K,M]; trans_b: b is [K,N]."""
    if not trans_a and not trans_b:
        return gemm_nt(a, b, bias=bias, residual=residual, act=act, out=out, out_dtype=out_dtype)
    _chk(a, BF16, "gemm.a"); _chk(b, BF16, "gemm.b")
    assert a.dim() == 2 and b.dim() == 2 and a.stride(1) == 1 and b.stride(1) == 1
    K, M = (a.shape[0], a.shape[1]) if trans_a else (a.shape[1], a.shape[0])
    Kb, N = (b.shape[0], b.shape[1]) if trans_b else (b.shape[1], b.shape[0])
    assert K == Kb, (a.shape, b.shape, trans_a, trans_b)
    if out is None:
        out = torch.empty((M, N), dtype=out_dtype, device=a.device)
    assert out.shape == (M, N) and out.stride(1) == 1
    ldr = residual.stride(0) if residual is not None else 0
    hook = _gemm_hook
    if hook is not None:
        hook(M, N, K, True)
    check(_lib.load().hvlm_gemm_bf16(_p(a), _p(b), _p(out), M, N, K, a.stride(0), b.stride(0), out.stride(0), _p(bias),
                                     _p(residual), ldr, act, 1 if out.dtype == torch.float32 else 0,
                                     1 if trans_a else 0, 1 if trans_b else 0, _stream()), "gemm_bf16")
    if hook is not None:
        hook(M, N, K, False)
    return out


def quantize_fp8(x: torch.Tensor, scale) -> torch.Tensor:
    """bf16 -> OCP e4m3 bytes (uint8 storage): out = e4m3(clamp(x * scale, +-448)).  `scale` is a python float or a
    1-element fp32 GPU tensor (read on the device: no host sync)."""
    _chk(x, BF16, "quantize_fp8.x")
    assert x.is_contiguous() and x.numel() % 8 == 0
    out = torch.empty(x.shape, dtype=torch.uint8, device=x.device)
    if isinstance(scale, torch.Tensor):
        assert scale.dtype == torch.float32 and scale.numel() == 1 and scale.is_cuda
        check(_lib.load().hvlm_quantize_fp8(_p(x), _p(out), _p(scale), 0.0, x.numel(), _stream()), "quantize_fp8")
    else:
        check(_lib.load().hvlm_quantize_fp8(_p(x), _p(out), None, float(scale), x.numel(), _stream()), "quantize_fp8")
    return out


def gemm_fp8_nt(a8: torch.Tensor, b8: torch.Tensor, alpha: float, bias: Optional[torch.Tensor] = None,
                residual: Optional[torch.Tensor] = None, act: int = ACT_NONE, out: Optional[torch.Tensor] = None,
                out_dtype=BF16) -> torch.Tensor:
    """out[M,N] = act(alpha * a8[M,K] @ b8[N,K]^T + bias) + residual on e4m3 operands (uint8 storage), K % 128 == 0."""
    assert a8.dtype == torch.uint8 and b8.dtype == torch.uint8 and a8.is_cuda and b8.is_cuda
    assert a8.dim() == 2 and b8.dim() == 2 and a8.shape[1] == b8.shape[1] and a8.stride(1) == 1 and b8.stride(1) == 1
    M, K = a8.shape
    N = b8.shape[0]
    if out is None:
        out = torch.empty((M, N), dtype=out_dtype, device=a8.device)
    ldr = residual.stride(0) if residual is not None else 0
    hook = _gemm_hook
    if hook is not None:
        hook(M, N, K, True)
    check(_lib.load().hvlm_gemm_fp8_nt(_p(a8), _p(b8), _p(out), M, N, K, a8.stride(0), b8.stride(0), out.stride(0),
                                       float(alpha), _p(bias), _p(residual), ldr, act,
                                       1 if out.dtype == torch.float32 else 0, _stream()), "gemm_fp8_nt")
    if hook is not None:
        hook(M, N, K, False)
    return out


def quantize_fp8_rowwise(x: torch.Tensor):
    """bf16 [R, K] -> (e4m3 bytes [R, K], inv_scale f32 [R]) with one scale per row (token / output channel):
    x[r, :] ~= inv_scale[r] * e4m3[r, :]."""
    _chk(x, BF16, "quantize_fp8_rowwise.x")
    assert x.dim() == 2 and x.stride(1) == 1
    R, K = x.shape
    out = torch.empty((R, K), dtype=torch.uint8, device=x.device)
    inv = torch.empty((R,), dtype=torch.float32, device=x.device)
    check(_lib.load().hvlm_quantize_fp8_rowwise(_p(x), _p(out), _p(inv), R, K, x.stride(0), out.stride(0), _stream()),
          "quantize_fp8_rowwise")
    return out, inv


def gemm_fp8_nt_scaled(a8, b8, row_scale, col_scale, alpha: float = 1.0, bias=None, residual=None, act: int = ACT_NONE,
                       out=None, out_dtype=BF16):
    """out[M,N] = act(alpha * row_scale[m] * col_scale[n] * a8[M,K] @ b8[N,K]^T + bias) + residual (e4m3 operands)."""
    assert a8.dtype == torch.uint8 and b8.dtype == torch.uint8 and a8.is_cuda and b8.is_cuda
    assert a8.dim() == 2 and b8.dim() == 2 and a8.shape[1] == b8.shape[1] and a8.stride(1) == 1 and b8.stride(1) == 1
    M, K = a8.shape
    N = b8.shape[0]
    assert row_scale is None or (row_scale.dtype == torch.float32 and row_scale.numel() == M)
    assert col_scale is None or (col_scale.dtype == torch.float32 and col_scale.numel() == N)
    if out is None:
        out = torch.empty((M, N), dtype=out_dtype, device=a8.device)
    ldr = residual.stride(0) if residual is not None else 0
    hook = _gemm_hook
    if hook is not None:
        hook(M, N, K, True)
    check(_lib.load().hvlm_gemm_fp8_nt_scaled(_p(a8), _p(b8), _p(out), M, N, K, a8.stride(0), b8.stride(0), out.stride(0),
                                              float(alpha), _p(row_scale), _p(col_scale), _p(bias), _p(residual), ldr, act,
                                              1 if out.dtype == torch.float32 else 0, _stream()), "gemm_fp8_nt_scaled")
    if hook is not None:
        hook(M, N, K, False)
    return out


def gemm_w8a16_nt(a: torch.Tensor, b8: torch.Tensor, col_scale: torch.Tensor, bias=None, residual=None, act: int = ACT_NONE,
                  out=None) -> torch.Tensor:
    """out[M,N] = act(col_scale[n] * a[M,K] @ e4m3(b8)[N,K]^T + bias) + residual for M <= 32: bf16 activations against e4m3
    weights (hvlm_quantize_fp8_rowwise of W) on the weight-streaming kernel — half the HBM bytes of gemm_nt, no
    activation quantisation."""
    _chk(a, BF16, "gemm_w8a16.a")
    assert b8.dtype == torch.uint8 and b8.is_cuda and a.dim() == 2 and b8.dim() == 2 and a.shape[1] == b8.shape[1]
    assert a.stride(1) == 1 and b8.stride(1) == 1 and col_scale.dtype == torch.float32 and col_scale.numel() == b8.shape[0]
    M, K = a.shape
    N = b8.shape[0]
    assert M <= 32 and K % 64 == 0 and N % 4 == 0, (M, N, K)
    if out is None:
        out = torch.empty((M, N), dtype=BF16, device=a.device)
    ldr = 0
    if residual is not None:
        _chk(residual, BF16, "gemm_w8a16.residual")
        assert residual.shape == (M, N) and residual.stride(1) == 1
        ldr = residual.stride(0)
    lib = _lib.load()
    hook = _gemm_hook
    if hook is not None:
        hook(M, N, K, True)
    ws = torch.empty((lib.hvlm_gemm_w8a16_skinny_ws_bytes(M, N, K),), dtype=torch.uint8, device=a.device)
    check(lib.hvlm_gemm_w8a16_nt_skinny(_p(a), _p(b8), _p(col_scale), _p(out), M, N, K, a.stride(0), b8.stride(0), out.stride(0),
                                        _p(bias), _p(residual), ldr, act, _p(ws), _stream()), "gemm_w8a16_skinny")
    if hook is not None:
        hook(M, N, K, False)
    return out


def _skinny_w(w):
    """(weight tensor, col_scale or None, ws-size function) for a bf16 weight or an (e4m3 bytes, inv_scale) pair."""
    lib = _lib.load()
    if isinstance(w, tuple):
        return w[0], w[1], lib.hvlm_gemm_w8a16_skinny_ws_bytes
    return w, None, lib.hvlm_gemm_skinny_ws_bytes


def gemm_skinny_swiglu(a: torch.Tensor, w_gu) -> torch.Tensor:
    """act[M, F] = SwiGLU(a @ w_gate^T, a @ w_up^T) for M <= 32; w_gu = [gate; up] rows as bf16 [2F, K] or (e4m3, inv_scale).
    == swiglu_fwd(gemm_nt(a, w_gu)) bit for bit (the SwiGLU rides in the split-K finalize pass)."""
    _chk(a, BF16, "gemm_skinny_swiglu.a")
    w, cs, wsb = _skinny_w(w_gu)
    M, K = a.shape
    F = w.shape[0] // 2
    act = torch.empty((M, F), dtype=BF16, device=a.device)
    ws = torch.empty((wsb(M, 2 * F, K),), dtype=torch.uint8, device=a.device)
    check(_lib.load().hvlm_gemm_skinny_swiglu(_p(a), _p(w), _p(cs), _p(act), M, F, K, a.stride(0), w.stride(0), _p(ws), _stream()),
          "gemm_skinny_swiglu")
    return act


def gemm_skinny_qkv(a: torch.Tensor, w_qkv, cache: torch.Tensor, lens: torch.Tensor, cos: torch.Tensor, sin: torch.Tensor,
                    H: int, dh: int) -> torch.Tensor:
    """qkv[M, 3D] = RoPE(a @ w_qkv^T) at positions lens[m], k | v columns also written to cache[m, lens[m]] ([M, max_len, 3D]).
    == gemm_nt + rope_qk_ + the cache row write, bit for bit."""
    _chk(a, BF16, "gemm_skinny_qkv.a"); _chk(cache, BF16, "gemm_skinny_qkv.cache"); _chk(lens, torch.int32, "gemm_skinny_qkv.lens")
    w, cs, wsb = _skinny_w(w_qkv)
    M, K = a.shape
    N = 3 * H * dh
    assert w.shape[0] == N and cache.is_contiguous() and cache.shape[0] == M and cache.shape[2] == N
    qkv = torch.empty((M, N), dtype=BF16, device=a.device)
    ws = torch.empty((wsb(M, N, K),), dtype=torch.uint8, device=a.device)
    check(_lib.load().hvlm_gemm_skinny_qkv(_p(a), _p(w), _p(cs), _p(qkv), N, _p(cache), cache.shape[1], _p(lens), _p(cos), _p(sin),
                                           M, H, dh, K, a.stride(0), w.stride(0), _p(ws), _stream()), "gemm_skinny_qkv")
    return qkv


def transpose(x: torch.Tensor, r_pad: Optional[int] = None, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """x[R,C] -> out[C,R_pad] (columns R..R_pad-1 zero)."""
    _chk(x, BF16, "transpose.x")
    assert x.dim() == 2 and x.stride(1) == 1
    R, C = x.shape
    r_pad = R if r_pad is None else r_pad
    if out is None:
        out = torch.empty((C, r_pad), dtype=BF16, device=x.device)
    assert out.shape == (C, r_pad) and out.is_contiguous()
    hk = _hbm_hook
    if hk is not None:
        hk("transpose_kernel", 2.0 * (R * C + C * r_pad), True)
    check(_lib.load().hvlm_transpose_bf16(_p(x), _p(out), R, C, x.stride(0), r_pad, _stream()), "transpose")
    if hk is not None:
        hk("transpose_kernel", 2.0 * (R * C + C * r_pad), False)
    return out


def colsum(x: torch.Tensor) -> torch.Tensor:
    _chk(x, BF16, "colsum.x")
    R, C = x.shape
    out = torch.empty((C,), dtype=torch.float32, device=x.device)
    check(_lib.load().hvlm_colsum_bf16(_p(x), _p(out), R, C, x.stride(0), _stream()), "colsum")
    return out


# ------------------------------------------------------------------------------------------------
# norms
# ------------------------------------------------------------------------------------------------
def rmsnorm_fwd(x: torch.Tensor, w: torch.Tensor, eps: float, save_rstd: bool = True,
                out: Optional[torch.Tensor] = None):
    _chk(x, BF16, "rmsnorm.x"); _chk(w, BF16, "rmsnorm.w")
    assert x.is_contiguous()
    D = x.shape[-1]
    rows = x.numel() // D
    y = torch.empty_like(x) if out is None else out
    rstd = torch.empty((rows,), dtype=torch.float32, device=x.device) if save_rstd else None
    hk = _hbm_hook
    if hk is not None:
        hk("rmsnorm_fwd_kernel", 4.0 * rows * D, True)
    check(_lib.load().hvlm_rmsnorm_fwd(_p(x), _p(w), _p(y), _p(rstd), rows, D, eps, _stream()), "rmsnorm_fwd")
    if hk is not None:
        hk("rmsnorm_fwd_kernel", 4.0 * rows * D, False)
    return y, rstd


def rmsnorm_bwd(dy, x, w, rstd, add=None, out=None, dw_out=None):
    """Returns (dx [like x], dw [D] bf16).  dw_out: the gradient buffer to write dw into (a view of the layer's flat bucket:
    no 12-us device copy of 8 KB per norm afterwards)."""
    _chk(dy, BF16, "rmsnorm_bwd.dy")
    D = x.shape[-1]
    rows = x.numel() // D
    dx = torch.empty_like(x) if out is None else out
    part = torch.empty((NORM_BWD_BLOCKS, D), dtype=torch.float32, device=x.device)
    lib = _lib.load()
    hk = _hbm_hook
    nbytes = 2.0 * rows * D * (4 if add is not None else 3)          # dy, x (, add) read, dx written
    if hk is not None:
        hk("rmsnorm_bwd_kernel (+ reduce_rows)", nbytes, True)
    check(lib.hvlm_rmsnorm_bwd(_p(dy), _p(x), _p(w), _p(rstd), _p(add), _p(dx), _p(part), rows, D, _stream()),
          "rmsnorm_bwd")
    dw = torch.empty((D,), dtype=BF16, device=x.device) if dw_out is None else dw_out
    assert dw.dtype == BF16 and dw.numel() == D and dw.is_contiguous()
    check(lib.hvlm_reduce_rows_f32(_p(part), _p(dw), NORM_BWD_BLOCKS, D, 1, _stream()), "reduce_rows")
    if hk is not None:
        hk("rmsnorm_bwd_kernel (+ reduce_rows)", nbytes, False)
    return dx, dw


def layernorm_fwd(x, w, b, eps: float, out=None):
    _chk(x, BF16, "layernorm.x")
    assert x.is_contiguous()
    D = x.shape[-1]
    rows = x.numel() // D
    y = torch.empty_like(x) if out is None else out
    check(_lib.load().hvlm_layernorm_fwd(_p(x), _p(w), _p(b), _p(y), rows, D, eps, _stream()), "layernorm_fwd")
    return y


def rmsnorm_fwd_fp8(x: torch.Tensor, w: torch.Tensor, eps: float):
    """RMSNorm of bf16 rows (D = 2048 / 4096 / 5120) written as (e4m3 bytes, inv_scale) == quantize_fp8_rowwise(rmsnorm_fwd(x)[0])."""
    _chk(x, BF16, "rmsnorm_fp8.x"); _chk(w, BF16, "rmsnorm_fp8.w")
    assert x.is_contiguous()
    D = x.shape[-1]
    rows = x.numel() // D
    out = torch.empty((rows, D), dtype=torch.uint8, device=x.device)
    inv = torch.empty((rows,), dtype=torch.float32, device=x.device)
    check(_lib.load().hvlm_rmsnorm_fwd_fp8(_p(x), _p(w), _p(out), _p(inv), rows, D, eps, _stream()), "rmsnorm_fwd_fp8")
    return out, inv


def layernorm_fwd_fp8(x, w, b, eps: float, bound=None):
    """LayerNorm of bf16 rows, written as (e4m3 bytes [rows, D], inv_scale f32 [rows]) == quantize_fp8_rowwise(layernorm_fwd(x)).
    bound = (mul, add): also returns next_inv[r] = (|y_r|_2 * mul + add) / 448, the e4m3 output scale of the GEMM that consumes
    these rows (gemm_fp8_nt_scaled_q8)."""
    _chk(x, BF16, "layernorm_fp8.x")
    assert x.is_contiguous()
    D = x.shape[-1]
    rows = x.numel() // D
    out = torch.empty((rows, D), dtype=torch.uint8, device=x.device)
    inv = torch.empty((rows,), dtype=torch.float32, device=x.device)
    nxt = torch.empty((rows,), dtype=torch.float32, device=x.device) if bound is not None else None
    mul, add = bound if bound is not None else (0.0, 0.0)
    check(_lib.load().hvlm_layernorm_fwd_fp8(_p(x), _p(w), _p(b), _p(out), _p(inv), _p(nxt), float(mul), float(add), rows, D, eps,
                                             _stream()), "layernorm_fwd_fp8")
    return (out, inv) if bound is None else (out, inv, nxt)


def gemm_fp8_nt_scaled_q8(a8, b8, row_scale, col_scale, out_inv, bias=None, act: int = ACT_NONE):
    """e4m3 in, e4m3 out: (bytes [M, N], out_inv) with out[m, n] = e4m3(bf16(act(rs[m] cs[n] a8 b8^T + bias)) / out_inv[m]);
    out_inv must bound the row (see hvlm_gemm_fp8_nt_scaled_q8)."""
    assert a8.dtype == torch.uint8 and b8.dtype == torch.uint8 and a8.dim() == 2 and b8.dim() == 2 and a8.shape[1] == b8.shape[1]
    M, K = a8.shape
    N = b8.shape[0]
    assert out_inv.dtype == torch.float32 and out_inv.numel() == M and N % 16 == 0
    out = torch.empty((M, N), dtype=torch.uint8, device=a8.device)
    hook = _gemm_hook
    if hook is not None:
        hook(M, N, K, True)
    check(_lib.load().hvlm_gemm_fp8_nt_scaled_q8(_p(a8), _p(b8), _p(out), M, N, K, a8.stride(0), b8.stride(0), out.stride(0), 1.0,
                                                 _p(row_scale), _p(col_scale), _p(bias), act, _p(out_inv), _stream()), "gemm_fp8_q8")
    if hook is not None:
        hook(M, N, K, False)
    return out


# ------------------------------------------------------------------------------------------------
# rope / attention / swiglu
# ------------------------------------------------------------------------------------------------
def rope_qk_(qkv: torch.Tensor, cos: torch.Tensor, sin: torch.Tensor, pos: torch.Tensor, H: int, dh: int,
             inverse: bool = False) -> torch.Tensor:
    """In place on the q,k sections of qkv [M, ld]."""
    _chk(qkv, BF16, "rope.qkv"); _chk(cos, BF16, "rope.cos"); _chk(sin, BF16, "rope.sin")
    _chk(pos, torch.int32, "rope.pos")
    assert qkv.dim() == 2 and qkv.stride(1) == 1
    M = qkv.shape[0]
    check(_lib.load().hvlm_rope_qk(_p(qkv), _p(cos), _p(sin), _p(pos), M, H, dh, qkv.stride(0),
                                   1 if inverse else 0, _stream()), "rope_qk")
    return qkv


def attn_fwd(qkv: torch.Tensor, B: int, S: int, H: int, dh: int, causal: bool, scale: float,
             need_lse: bool = True, out: Optional[torch.Tensor] = None, cu_seqlens: Optional[torch.Tensor] = None):
    """Padded batch: qkv [B*S, ld].  With cu_seqlens (int32 [B+1], device): the tokens are PACKED (qkv [total, ld]) and
    S is max_seqlen — flash_attn_varlen_qkvpacked_func's contract (llama_flash_attn_monkey_patch.py:81-94)."""
    _chk(qkv, BF16, "attn.qkv")
    assert qkv.dim() == 2 and qkv.stride(1) == 1
    lib = _lib.load()
    rows = qkv.shape[0]
    assert cu_seqlens is not None or rows == B * S
    o = torch.empty((rows, H * dh), dtype=BF16, device=qkv.device) if out is None else out
    lse = torch.empty((B, H, S), dtype=torch.float32, device=qkv.device) if need_lse else None
    ws = torch.empty((lib.hvlm_attn_fwd_ws_bytes(B, S, H, dh),), dtype=torch.uint8, device=qkv.device)
    if cu_seqlens is None:
        check(lib.hvlm_attn_fwd(_p(qkv), _p(o), _p(lse), _p(ws), B, S, H, dh, qkv.stride(0), 1 if causal else 0,
                                scale, _stream()), "attn_fwd")
    else:
        _chk(cu_seqlens, torch.int32, "attn.cu_seqlens")
        assert cu_seqlens.numel() == B + 1
        check(lib.hvlm_attn_fwd_varlen(_p(qkv), _p(o), _p(lse), _p(ws), _p(cu_seqlens), B, S, H, dh, qkv.stride(0),
                                       1 if causal else 0, scale, _stream()), "attn_fwd_varlen")
    return o, lse


def attn_bwd(qkv, out, dout, lse, B, S, H, dh, causal: bool, scale: float, dqkv=None, cu_seqlens=None, rope=None):
    """rope = (cos, sin, pos): the gradient of the rotary embedding is applied inside the dQ / dK epilogues — dqkv is then
    the gradient w.r.t. the un-rotated q / k, bit-identical to attn_bwd followed by rope_qk_(dqkv, ..., inverse=True)."""
    _chk(dout, BF16, "attn_bwd.dout")
    assert dout.is_contiguous() and out.is_contiguous()
    lib = _lib.load()
    if dqkv is None:
        dqkv = torch.empty_like(qkv)
    ws = torch.empty((lib.hvlm_attn_bwd_ws_bytes(B, S, H, dh),), dtype=torch.uint8, device=qkv.device)
    if rope is not None:
        cos, sin, pos = rope
        _chk(cos, BF16, "attn_bwd.cos"); _chk(sin, BF16, "attn_bwd.sin"); _chk(pos, torch.int32, "attn_bwd.pos")
        assert pos.numel() == qkv.shape[0] and cos.shape[-1] == dh // 2
        check(lib.hvlm_attn_bwd_rope(_p(qkv), _p(out), _p(dout), _p(lse), _p(dqkv), _p(ws), _p(cu_seqlens), B, S, qkv.shape[0],
                                     H, dh, qkv.stride(0), 1 if causal else 0, scale, _p(cos), _p(sin), _p(pos), _stream()),
              "attn_bwd_rope")
        return dqkv
    if cu_seqlens is None:
        check(lib.hvlm_attn_bwd(_p(qkv), _p(out), _p(dout), _p(lse), _p(dqkv), _p(ws), B, S, H, dh, qkv.stride(0),
                                1 if causal else 0, scale, _stream()), "attn_bwd")
    else:
        check(lib.hvlm_attn_bwd_varlen(_p(qkv), _p(out), _p(dout), _p(lse), _p(dqkv), _p(ws), _p(cu_seqlens), B, S,
                                       qkv.shape[0], H, dh, qkv.stride(0), 1 if causal else 0, scale, _stream()), "attn_bwd_varlen")
    return dqkv


def gemv(w: torch.Tensor, x: torch.Tensor, residual: Optional[torch.Tensor] = None, out=None) -> torch.Tensor:
    """y[N] = w[N,K] @ x[K] (+ residual)."""
    _chk(w, BF16, "gemv.w"); _chk(x, BF16, "gemv.x")
    N, K = w.shape
    assert x.numel() == K and w.stride(1) == 1
    y = torch.empty((N,), dtype=BF16, device=w.device) if out is None else out
    check(_lib.load().hvlm_gemv_bf16(_p(w), _p(x), _p(y), _p(residual), N, K, w.stride(0), _stream()), "gemv")
    return y


def attn_decode(q: torch.Tensor, k_cache: torch.Tensor, v_cache: torch.Tensor, S: int, H: int, dh: int, scale: float):
    _chk(q, BF16, "attn_decode.q")
    assert k_cache.stride(1) == 1 and k_cache.stride(0) == v_cache.stride(0)
    out = torch.empty((H * dh,), dtype=BF16, device=q.device)
    check(_lib.load().hvlm_attn_decode(_p(q), _p(k_cache), _p(v_cache), _p(out), S, H, dh, k_cache.stride(0), scale,
                                       _stream()), "attn_decode")
    return out


def attn_decode_batched(q: torch.Tensor, cache: torch.Tensor, lens: torch.Tensor, H: int, dh: int, scale: float):
    """q [B, >= H*dh] (row b = the new token of sample b, RoPE'd; may be a view of the fused qkv rows), cache [B, max_S, 3*H*dh]
    (fused qkv layout: k at column H*dh, v at 2*H*dh), lens int32 [B] on the device -> out [B, H*dh]."""
    _chk(q, BF16, "attn_decode_batched.q"); _chk(cache, BF16, "attn_decode_batched.cache"); _chk(lens, torch.int32, "attn_decode_batched.lens")
    B, max_S, ld = cache.shape
    assert q.shape[0] == B and q.stride(1) == 1 and cache.is_contiguous() and ld == 3 * H * dh
    D = H * dh
    out = torch.empty((B, D), dtype=BF16, device=q.device)
    kc = ctypes.c_void_p(cache.data_ptr() + 2 * D)
    vc = ctypes.c_void_p(cache.data_ptr() + 4 * D)
    check(_lib.load().hvlm_attn_decode_batched(_p(q), q.stride(0), kc, vc, cache.stride(0), _p(lens), _p(out), D, B, max_S, H, dh,
                                               ld, scale, _stream()), "attn_decode_batched")
    return out


def kv_cache_fill(qkv: torch.Tensor, cache: torch.Tensor, B: int, S: int) -> None:
    """cache[b, s, D:] = qkv[b * S + s, D:] for the [B, max_len, 3D] cache of lock-step generation (k | v columns only)."""
    _chk(qkv, BF16, "kv_cache_fill.qkv"); _chk(cache, BF16, "kv_cache_fill.cache")
    assert qkv.is_contiguous() and cache.is_contiguous() and cache.dim() == 3 and cache.shape[0] == B and cache.shape[2] == qkv.shape[1]
    assert qkv.shape[0] == B * S and cache.shape[1] >= S and qkv.shape[1] % 3 == 0
    check(_lib.load().hvlm_kv_cache_fill(_p(qkv), _p(cache), B, S, cache.shape[1], qkv.shape[1] // 3, _stream()), "kv_cache_fill")


def swiglu_fwd_fp8(gu: torch.Tensor):
    """gu [M, 2F] bf16 -> (e4m3 bytes [M, F], inv_scale f32 [M]) == quantize_fp8_rowwise(swiglu_fwd(gu))."""
    _chk(gu, BF16, "swiglu_fp8.gu")
    assert gu.is_contiguous()
    M, F2 = gu.shape
    out = torch.empty((M, F2 // 2), dtype=torch.uint8, device=gu.device)
    inv = torch.empty((M,), dtype=torch.float32, device=gu.device)
    check(_lib.load().hvlm_swiglu_fwd_fp8(_p(gu), _p(out), _p(inv), M, F2 // 2, _stream()), "swiglu_fwd_fp8")
    return out, inv


def swiglu_fwd(gu: torch.Tensor, out=None) -> torch.Tensor:
    _chk(gu, BF16, "swiglu.gu")
    assert gu.is_contiguous()
    M, F2 = gu.shape
    act = torch.empty((M, F2 // 2), dtype=BF16, device=gu.device) if out is None else out
    check(_lib.load().hvlm_swiglu_fwd(_p(gu), _p(act), M, F2 // 2, _stream()), "swiglu_fwd")
    return act


def swiglu_fwd_t(gu: torch.Tensor, r_pad: int):
    """(act [M, F], actT [F, r_pad]) — the transposed copy is the dW_down operand (columns M.. zero)."""
    _chk(gu, BF16, "swiglu.gu")
    assert gu.is_contiguous()
    M, F2 = gu.shape
    act = torch.empty((M, F2 // 2), dtype=BF16, device=gu.device)
    actT = torch.empty((F2 // 2, r_pad), dtype=BF16, device=gu.device)
    hk = _hbm_hook
    if hk is not None:
        hk("swiglu_fwd_t_kernel", 2.0 * (M * F2 + M * (F2 // 2) + (F2 // 2) * r_pad), True)
    check(_lib.load().hvlm_swiglu_fwd_t(_p(gu), _p(act), _p(actT), M, F2 // 2, r_pad, _stream()), "swiglu_fwd_t")
    if hk is not None:
        hk("swiglu_fwd_t_kernel", 2.0 * (M * F2 + M * (F2 // 2) + (F2 // 2) * r_pad), False)
    return act, actT


def swiglu_bwd_t(gu, dact, r_pad: int):
    """(dgu [M, 2F], dguT [2F, r_pad])."""
    M, F2 = gu.shape
    dgu = torch.empty_like(gu)
    dguT = torch.empty((F2, r_pad), dtype=BF16, device=gu.device)
    hk = _hbm_hook
    if hk is not None:
        hk("swiglu_bwd_t_kernel", 2.0 * (M * F2 + M * (F2 // 2) + M * F2 + F2 * r_pad), True)
    check(_lib.load().hvlm_swiglu_bwd_t(_p(gu), _p(dact), _p(dgu), _p(dguT), M, F2 // 2, r_pad, _stream()), "swiglu_bwd_t")
    if hk is not None:
        hk("swiglu_bwd_t_kernel", 2.0 * (M * F2 + M * (F2 // 2) + M * F2 + F2 * r_pad), False)
    return dgu, dguT


def swiglu_bwd(gu, dact, out=None):
    M, F2 = gu.shape
    dgu = torch.empty_like(gu) if out is None else out
    check(_lib.load().hvlm_swiglu_bwd(_p(gu), _p(dact), _p(dgu), M, F2 // 2, _stream()), "swiglu_bwd")
    return dgu


# ------------------------------------------------------------------------------------------------
# visual pipeline
# ------------------------------------------------------------------------------------------------
def patch_im2col(images: torch.Tensor, P: int, kpad: int, to_f16: bool = False) -> torch.Tensor:
    """bf16 frames [N, 3, HW, HW] -> patch rows [N * (HW/P)^2, kpad]: bf16, or (to_f16) IEEE half — the frames' cast to the
    fp16 tower's dtype applied on the way through."""
    _chk(images, BF16, "im2col.images")
    assert images.is_contiguous() and images.dim() == 4 and images.shape[1] == 3
    N, _, HW, _ = images.shape
    G = HW // P
    out = torch.empty((N * G * G, kpad), dtype=torch.float16 if to_f16 else BF16, device=images.device)
    fn = _lib.load().hvlm_patch_im2col_f16 if to_f16 else _lib.load().hvlm_patch_im2col
    check(fn(_p(images), _p(out), N, HW, P, kpad, _stream()), "patch_im2col")
    return out


def clip_embed(patch_out, cls, pos, N: int, NP: int, D: int) -> torch.Tensor:
    tokens = torch.empty((N, NP + 1, D), dtype=BF16, device=patch_out.device)
    check(_lib.load().hvlm_clip_embed(_p(patch_out), _p(cls), _p(pos), _p(tokens), N, NP, D, _stream()), "clip_embed")
    return tokens


def temporal_spatial_pool_fwd(tokens: torch.Tensor, B: int, T: int, drop_cls: bool, grid: int = 16, out=None, frame_index=None) -> torch.Tensor:
    """grid = patches per side (16: the reference's 224-px geometry; 24: 336 px).  bf16 or fp16 tokens (output in the same dtype).
    frame_index (int32 [B*T], fp16 tokens only): frame (b, t) = token block frame_index[b*T + t] (frame de-duplication)."""
    if tokens.dtype != torch.float16:
        _chk(tokens, BF16, "pool.tokens")
    assert tokens.is_cuda and tokens.is_contiguous()
    D = tokens.shape[-1]
    if out is None:
        out = torch.empty((B, T + grid * grid, D), dtype=tokens.dtype, device=tokens.device)
    assert out.shape == (B, T + grid * grid, D) and out.dtype == tokens.dtype and out.is_contiguous()
    if tokens.dtype == torch.float16:
        if frame_index is not None:
            _chk(frame_index, torch.int32, "pool.frame_index")
            assert frame_index.numel() == B * T
        check(_lib.load().hvlm_temporal_spatial_pool_fwd_f16(_p(tokens), _p(out), B, T, D, 1 if drop_cls else 0, grid, _p(frame_index), _stream()), "pool_fwd_f16")
    else:
        assert frame_index is None, "frame_index: fp16 tokens only"
        check(_lib.load().hvlm_temporal_spatial_pool_fwd(_p(tokens), _p(out), B, T, D, 1 if drop_cls else 0, grid, _stream()), "pool_fwd")
    return out


def temporal_spatial_pool_bwd(dout: torch.Tensor, B: int, T: int, grid: int = 16) -> torch.Tensor:
    D = dout.shape[-1]
    dtok = torch.empty((B, T, grid * grid, D), dtype=BF16, device=dout.device)
    check(_lib.load().hvlm_temporal_spatial_pool_bwd(_p(dout), _p(dtok), B, T, D, grid, _stream()), "pool_bwd")
    return dtok


def embed_splice_fwd(embed, visual, src, hand_slot, hands, hand_scale, B, S, D, NV, NH, out=None, row_sample=None):
    """Padded: rows = B*S, sample of a row = row // S.  Packed (row_sample int32 [rows]): rows = src.numel()."""
    _chk(embed, BF16, "splice.embed"); _chk(src, torch.int32, "splice.src")
    rows = src.numel()
    x = torch.empty((rows, D), dtype=BF16, device=embed.device) if out is None else out
    if row_sample is None:
        assert rows == B * S
        check(_lib.load().hvlm_embed_splice_fwd(_p(embed), _p(visual), _p(src), _p(hand_slot), _p(hands), _p(hand_scale),
                                                _p(x), B, S, D, NV, NH, _stream()), "embed_splice_fwd")
    else:
        _chk(row_sample, torch.int32, "splice.row_sample")
        check(_lib.load().hvlm_embed_splice_fwd_packed(_p(embed), _p(visual), _p(src), _p(hand_slot), _p(hands),
                                                       _p(hand_scale), _p(row_sample), _p(x), rows, D, NV, NH, _stream()),
              "embed_splice_fwd_packed")
    return x


def embed_splice_bwd(dx, src, dembed, dvisual, B, S, D, NV, row_sample=None):
    if row_sample is None:
        check(_lib.load().hvlm_embed_splice_bwd(_p(dx), _p(src), _p(dembed), _p(dvisual), B, S, D, NV, _stream()),
              "embed_splice_bwd")
    else:
        check(_lib.load().hvlm_embed_splice_bwd_packed(_p(dx), _p(src), _p(row_sample), _p(dembed), _p(dvisual),
                                                       src.numel(), D, NV, _stream()), "embed_splice_bwd_packed")


def embed_grad_segments(dx, seg_rows, seg_start, seg_token, dembed):
    """Deterministic text-embedding gradient: dembed[seg_token[i]] = sum of dx rows seg_rows[seg_start[i]:seg_start[i+1]]."""
    _chk(dx, BF16, "embed_grad.dx"); _chk(dembed, BF16, "embed_grad.dembed")
    assert dx.is_contiguous() and dembed.is_contiguous() and dx.shape[1] == dembed.shape[1]
    check(_lib.load().hvlm_embed_grad_segments(_p(dx), _p(seg_rows), _p(seg_start), _p(seg_token), _p(dembed),
                                               seg_token.numel(), dx.shape[1], _stream()), "embed_grad_segments")


# ------------------------------------------------------------------------------------------------
# heads / optimizer
# ------------------------------------------------------------------------------------------------
def ce_fwd(logits: torch.Tensor, targets: torch.Tensor, V: int):
    """Returns (loss_sum[2] f32 = [sum nll, n_valid], row_lse[M])."""
    _chk(logits, BF16, "ce.logits"); _chk(targets, torch.int32, "ce.targets")
    M, ldl = logits.shape[0], logits.stride(0)
    acc = torch.zeros((2,), dtype=torch.float32, device=logits.device)
    row_lse = torch.empty((M,), dtype=torch.float32, device=logits.device)
    check(_lib.load().hvlm_ce_fwd(_p(logits), _p(targets), _p(acc), _p(row_lse), M, V, ldl, _stream()), "ce_fwd")
    return acc, row_lse


def ce_bwd_(logits, targets, row_lse, grad_scale: torch.Tensor, V: int):
    """In place: logits <- dlogits."""
    M, ldl = logits.shape[0], logits.stride(0)
    check(_lib.load().hvlm_ce_bwd(_p(logits), _p(targets), _p(row_lse), _p(grad_scale), _p(logits), M, V, ldl,
                                  _stream()), "ce_bwd")
    return logits


def hand_gather_fwd(hidden: torch.Tensor, rows: torch.Tensor, B: int, D: int, out_dtype=torch.float32):
    _chk(hidden, BF16, "hand_gather.hidden"); _chk(rows, torch.int32, "hand_gather.rows")
    out = torch.empty((B, 2, 4, D // 2), dtype=out_dtype, device=hidden.device)
    check(_lib.load().hvlm_hand_gather_fwd(_p(hidden), _p(rows), _p(out), B, D,
                                           1 if out_dtype == torch.float32 else 0, _stream()), "hand_gather_fwd")
    return out


def hand_gather_bwd_(dout: torch.Tensor, rows: torch.Tensor, dhidden: torch.Tensor, B: int, D: int):
    assert dout.is_contiguous()
    check(_lib.load().hvlm_hand_gather_bwd(_p(dout), _p(rows), _p(dhidden), B, D,
                                           1 if dout.dtype == torch.float32 else 0, _stream()), "hand_gather_bwd")
    return dhidden


def adamw_step_(master, m, v, grad, w, wt, lr, beta1, beta2, eps, wd, grad_scale, step):
    R, C = (w.shape[0], w.shape[1]) if w.dim() == 2 else (1, w.numel())
    hk = _hbm_hook
    nbytes = float(R * C) * (30 if wt is not None else 28)      # fp32 master / m / v read + written, bf16 g read, bf16 w (and w^T) written
    if hk is not None:
        hk("adamw_kernel", nbytes, True)
    check(_lib.load().hvlm_adamw_step(_p(master), _p(m), _p(v), _p(grad), _p(w), _p(wt), R, C, lr, beta1, beta2, eps,
                                      wd, grad_scale, step, _stream()), "adamw_step")
    if hk is not None:
        hk("adamw_kernel", nbytes, False)


def add(a, b, out=None):
    y = torch.empty_like(a) if out is None else out
    check(_lib.load().hvlm_add_bf16(_p(a), _p(b), _p(y), a.numel(), _stream()), "add")
    return y


def cast_f32_to_bf16(x, out=None):
    y = torch.empty(x.shape, dtype=BF16, device=x.device) if out is None else out
    check(_lib.load().hvlm_cast_f32_to_bf16(_p(x), _p(y), x.numel(), _stream()), "cast")
    return y


# ------------------------------------------------------------------------------------------------
# fp16 instantiations (CLIP tower in the reference's dtype, handsonvlm/train/train.py:316)
# ------------------------------------------------------------------------------------------------
F16 = torch.float16


def gemm_f16_nt(a, b, bias=None, residual=None, act: int = ACT_NONE, out=None):
    """fp16 operands / output: out[M,N] = act(a[M,K] @ b[N,K]^T + bias) + residual."""
    _chk(a, F16, "gemm_f16.a"); _chk(b, F16, "gemm_f16.b")
    assert a.dim() == 2 and b.dim() == 2 and a.shape[1] == b.shape[1] and a.stride(1) == 1 and b.stride(1) == 1
    M, K = a.shape
    N = b.shape[0]
    if out is None:
        out = torch.empty((M, N), dtype=F16, device=a.device)
    assert out.dtype == F16 and out.shape == (M, N)
    ldr = 0
    if residual is not None:
        _chk(residual, F16, "gemm_f16.residual")
        ldr = residual.stride(0)
    if bias is not None:
        _chk(bias, F16, "gemm_f16.bias")
    hook = _gemm_hook
    if hook is not None:
        hook(M, N, K, True)
    check(_lib.load().hvlm_gemm_f16_nt(_p(a), _p(b), _p(out), M, N, K, a.stride(0), b.stride(0), out.stride(0), _p(bias),
                                       _p(residual), ldr, act, 0, _stream()), "gemm_f16_nt")
    if hook is not None:
        hook(M, N, K, False)
    return out


def gemm_f16_nt_ln(a, b, bias=None, residual=None, act: int = ACT_NONE, out=None, stats_out=None, ln=None):
    """gemm_f16_nt with the LayerNorm-fold epilogues (hvlm_gemm_f16_nt_ln).  stats_out: fp32 [M, ln_stats_parts(N), 2] to receive
    the per-row partial sums of the output (needs `residual`); ln = (stats [M, 2] fp32 = (mean, rstd) of a's rows, c1 fp32 [N])."""
    _chk(a, F16, "gemm_f16_ln.a"); _chk(b, F16, "gemm_f16_ln.b")
    assert a.dim() == 2 and b.dim() == 2 and a.shape[1] == b.shape[1] and a.stride(1) == 1 and b.stride(1) == 1
    M, K = a.shape
    N = b.shape[0]
    if out is None:
        out = torch.empty((M, N), dtype=F16, device=a.device)
    ldr = 0
    if residual is not None:
        _chk(residual, F16, "gemm_f16_ln.residual")
        ldr = residual.stride(0)
    st, c1 = (None, None) if ln is None else ln
    if st is not None:
        _chk(st, torch.float32, "gemm_f16_ln.stats"); _chk(c1, torch.float32, "gemm_f16_ln.c1")
        assert st.shape == (M, 2) and st.is_contiguous() and c1.numel() == N
    if stats_out is not None:
        _chk(stats_out, torch.float32, "gemm_f16_ln.stats_out")
        assert stats_out.is_contiguous() and stats_out.numel() == M * ln_stats_parts(N) * 2
    hook = _gemm_hook
    if hook is not None:
        hook(M, N, K, True)
    check(_lib.load().hvlm_gemm_f16_nt_ln(_p(a), _p(b), _p(out), M, N, K, a.stride(0), b.stride(0), out.stride(0), _p(bias), _p(residual),
                                          ldr, act, _p(stats_out), _p(st), _p(c1), _stream()), "gemm_f16_nt_ln")
    if hook is not None:
        hook(M, N, K, False)
    return out


def ln_stats_parts(N: int) -> int:
    return int(_lib.load().hvlm_ln_stats_parts(N))


def ln_stats_finalize(part: torch.Tensor, rows: int, n_parts: int, D: int, eps: float, out=None) -> torch.Tensor:
    """Per-wave partial (sum, sum of squares) [rows, n_parts, 2] -> (mean, rstd) [rows, 2] (fp32)."""
    _chk(part, torch.float32, "ln_stats_finalize.part")
    st = torch.empty((rows, 2), dtype=torch.float32, device=part.device) if out is None else out
    check(_lib.load().hvlm_ln_stats_finalize(_p(part), _p(st), rows, n_parts, D, eps, _stream()), "ln_stats_finalize")
    return st


def layernorm_fwd_f16(x, w, b, eps: float, out=None):
    _chk(x, F16, "layernorm_f16.x")
    assert x.is_contiguous()
    D = x.shape[-1]
    y = torch.empty_like(x) if out is None else out
    check(_lib.load().hvlm_layernorm_fwd_f16(_p(x), _p(w), _p(b), _p(y), x.numel() // D, D, eps, _stream()), "layernorm_fwd_f16")
    return y


def clip_embed_f16(patch_out, cls, pos, N: int, NP: int, D: int):
    tokens = torch.empty((N, NP + 1, D), dtype=F16, device=patch_out.device)
    check(_lib.load().hvlm_clip_embed_f16(_p(patch_out), _p(cls), _p(pos), _p(tokens), N, NP, D, _stream()), "clip_embed_f16")
    return tokens


def attn_fwd_f16(qkv, B: int, S: int, H: int, dh: int, scale: float, out=None):
    _chk(qkv, F16, "attn_f16.qkv")
    assert qkv.dim() == 2 and qkv.shape[0] == B * S and qkv.stride(1) == 1
    lib = _lib.load()
    o = torch.empty((B * S, H * dh), dtype=F16, device=qkv.device) if out is None else out
    assert o.shape == (B * S, H * dh) and o.dtype == F16 and o.is_contiguous()
    ws = torch.empty((lib.hvlm_attn_fwd_ws_bytes(B, S, H, dh),), dtype=torch.uint8, device=qkv.device)
    check(lib.hvlm_attn_fwd_f16(_p(qkv), _p(o), None, _p(ws), B, S, H, dh, qkv.stride(0), 0, scale, _stream()), "attn_fwd_f16")
    return o


def cast16(x, to_dtype, out=None):
    """bf16 <-> fp16 (round-to-nearest-even), the casts of clip_encoder.py:44-48."""
    assert x.is_contiguous() and x.numel() % 8 == 0 and {x.dtype, to_dtype} == {BF16, F16}
    if out is None:
        out = torch.empty(x.shape, dtype=to_dtype, device=x.device)
    assert out.is_contiguous() and out.dtype == to_dtype and out.numel() == x.numel()
    fn = _lib.load().hvlm_cast_bf16_to_f16 if to_dtype == F16 else _lib.load().hvlm_cast_f16_to_bf16
    check(fn(_p(x), _p(out), x.numel(), _stream()), "cast16")
    return out
