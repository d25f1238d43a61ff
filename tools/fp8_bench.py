#!/usr/bin/env python3
"""fp8 (e4m3, scaled MFMA) GEMM vs the bf16 kernel on the forward shapes (GPU box)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "handsonvlm-release_amd"))
import torch
from hvlm import ops
dev = torch.device("cuda:0"); BF = torch.bfloat16

def timeit(fn, iters=10, warmup=3):
    for _ in range(warmup): fn()
    torch.cuda.synchronize()
    s = torch.cuda.Event(enable_timing=True); e = torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / iters

for (M, N, K) in [(13872, 12288, 4096), (13872, 4096, 4096), (13872, 22016, 4096), (13872, 4096, 11008), (8192, 8192, 8192),
                  (27744, 12288, 4096), (102800, 4096, 1024)]:
    a = torch.rand(M, K, device=dev).sub_(0.5).to(BF); b = torch.rand(N, K, device=dev).sub_(0.5).to(BF)
    a8 = ops.quantize_fp8(a, 448.0 / 0.5); b8 = ops.quantize_fp8(b, 448.0 / 0.5)
    out = torch.empty(M, N, dtype=BF, device=dev)
    ms8 = min(timeit(lambda: ops.gemm_fp8_nt(a8, b8, 1.0, out=out)) for _ in range(2))
    ms16 = min(timeit(lambda: ops.gemm_nt(a, b, out=out)) for _ in range(2))
    msq = timeit(lambda: ops.quantize_fp8(a, 896.0))
    print(f"{M}x{N}x{K}: fp8 {2.0*M*N*K/ms8/1e9:.0f} TF/s ({ms8*1e3:.0f} us)  bf16 {2.0*M*N*K/ms16/1e9:.0f} TF/s ({ms16*1e3:.0f} us)  "
          f"quantize A {msq*1e3:.0f} us ({3*M*K/msq/1e6:.0f} GB/s)", flush=True)
