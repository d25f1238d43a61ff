#!/usr/bin/env python3
"""The four projections of one decode step at batch 32 (M = 32, 7B dimensions) and the lm_head: the weight-streaming kernel
(hvlm_gemm_bf16_nt_skinny) against the 256 x 256 split-K path, in TB/s of W streamed.  Each shape cycles over 8 distinct
weight matrices so that nothing is served from L2 / Infinity Cache.  HVLM_SKINNY_PER = 1 .. 4 forces the K-slice length (x 256 k)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "handsonvlm-release_amd"))
import torch
from hvlm import ops, _lib

M = int(os.environ.get("M", "32"))
shapes = [("qkv", 12288, 4096, False), ("o", 4096, 4096, True), ("gate|up", 22016, 4096, False), ("down", 4096, 11008, True),
          ("lm_head", 32000, 4096, False)]
lib = _lib.load()
print(f"M={M}  HVLM_SKINNY_PER={os.environ.get('HVLM_SKINNY_PER', 'auto')}")
tot = {True: 0.0, False: 0.0}
for name, N, K, res in shapes:
    ws = [torch.randn(N, K, device="cuda").to(torch.bfloat16) for _ in range(8)]
    x = torch.randn(M, K, device="cuda").to(torch.bfloat16)
    r = torch.randn(M, N, device="cuda").to(torch.bfloat16) if res else None
    line = f"{name:8s} N={N:6d} K={K:6d} splits={lib.hvlm_gemm_skinny_splits(N, K):3d}"
    for skinny in (True, False):
        ops._GEMM_SKINNY = skinny
        for w in ws: ops.gemm_nt(x, w, residual=r)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(5):
            for w in ws: ops.gemm_nt(x, w, residual=r)
        torch.cuda.synchronize(); us = (time.perf_counter() - t0) / 40 * 1e6
        if name != "lm_head": tot[skinny] += us
        line += f"   {'stream' if skinny else 'tile  '} {us:6.1f} us {N * K * 2 / us / 1e6:5.2f} TB/s"
    print(line)
print(f"per layer: weight-streaming {tot[True]:.1f} us   tile split-K {tot[False]:.1f} us   (wall, finalize pass included)")
