#!/usr/bin/env python3
"""The CLIP tower's four GEMMs exactly as the folded fp16 tower issues them in the SFT step (1600 frames x 257 tokens, M = 411200):
q|k|v (LayerNorm-fold correction + bias), out_proj (bias + residual + row statistics), fc1 (LayerNorm-fold + bias + QuickGELU),
fc2 (bias + residual + row statistics), each also without its epilogue work — TFLOP/s per shape (HIP events, 10 launches).
A/B of two library builds: HVLM_LIB=handsonvlm-release_amd/lib/libhvlm_base.so python3 tools/clip_epi_bench.py"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "handsonvlm-release_amd"))
import torch
from hvlm import ops
F16 = torch.float16
M = int(sys.argv[1]) if len(sys.argv) > 1 else 411200


def t(fn, n=10):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    a = torch.cuda.Event(enable_timing=True); b = torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n


g = torch.Generator(device="cuda").manual_seed(0)
rn = lambda *s, sc=1.0: (torch.randn(*s, device="cuda", generator=g) * sc).to(F16)  # noqa: E731
x = rn(M, 1024)
h = rn(M, 4096, sc=0.5)
stats = torch.stack([x.float().mean(1), torch.rsqrt(x.float().var(1, unbiased=False) + 1e-5)], 1).contiguous()
res = {}
for name, N, K, kind in (("qkv", 3072, 1024, "ln"), ("out", 1024, 1024, "res"), ("fc1", 4096, 1024, "ln_gelu"), ("fc2", 1024, 4096, "res")):
    a = x if K == 1024 else h
    w = rn(N, K, sc=K ** -0.5)
    bias = rn(N, sc=0.1)
    c1 = w.float().sum(1).contiguous()
    out = torch.empty(M, N, dtype=F16, device="cuda")
    part = torch.empty((M, ops.ln_stats_parts(N), 2), dtype=torch.float32, device="cuda") if kind == "res" else None
    fl = 2.0 * M * N * K
    t_plain = t(lambda: ops.gemm_f16_nt(a, w, out=out))
    if kind == "res":
        t_full = t(lambda: ops.gemm_f16_nt_ln(a, w, bias=bias, residual=x, out=out, stats_out=part))
    else:
        t_full = t(lambda: ops.gemm_f16_nt_ln(a, w, bias=bias, act=ops.ACT_QUICK_GELU if kind == "ln_gelu" else ops.ACT_NONE, out=out, ln=(stats, c1)))
    res[name] = (fl / t_plain / 1e9, fl / t_full / 1e9, t_full)
    print(f"{name:4s} {M}x{N}x{K}: no epilogue {fl / t_plain / 1e9:6.0f}   as issued by the tower {fl / t_full / 1e9:6.0f} TF/s  ({t_full * 1e3:.0f} us)", flush=True)
print(f"layer total as issued: {sum(v[2] for v in res.values()) * 1e3:.0f} us")
