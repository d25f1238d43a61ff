#!/usr/bin/env python3
"""The CLIP tower's four GEMM shapes in isolation (fp16, 400 frames x 257 tokens), with and without their epilogues:
qkv (bias), out_proj (bias + residual), fc1 (bias + QuickGELU), fc2 (bias + residual)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "handsonvlm-release_amd"))
import torch
from hvlm import ops
F16 = torch.float16
def t(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    a = torch.cuda.Event(enable_timing=True); b = torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n
M = 102800
for name, N, K, use_res, act in (("qkv", 3072, 1024, False, ops.ACT_NONE), ("out", 1024, 1024, True, ops.ACT_NONE),
                                 ("fc1", 4096, 1024, False, ops.ACT_QUICK_GELU), ("fc2", 1024, 4096, True, ops.ACT_NONE)):
    a = torch.rand(M, K, device="cuda").sub_(0.5).to(F16); w = torch.rand(N, K, device="cuda").sub_(0.5).mul_(0.05).to(F16)
    bias = torch.rand(N, device="cuda").sub_(0.5).to(F16); res = torch.rand(M, N, device="cuda").sub_(0.5).to(F16) if use_res else None
    out = torch.empty(M, N, dtype=F16, device="cuda")
    fl = 2.0 * M * N * K
    t_plain = t(lambda: ops.gemm_f16_nt(a, w, out=out))
    t_bias = t(lambda: ops.gemm_f16_nt(a, w, bias=bias, out=out))
    t_full = t(lambda: ops.gemm_f16_nt(a, w, bias=bias, residual=res, act=act, out=out))
    print(f"{name:4s} {M}x{N}x{K}: plain {fl/t_plain/1e9:6.0f}  +bias {fl/t_bias/1e9:6.0f}  full epilogue {fl/t_full/1e9:6.0f} TF/s  ({t_full*1e3:.0f} us)")
