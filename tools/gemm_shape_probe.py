#!/usr/bin/env python3
"""Isolated rate of the persistent bf16 GEMM on a list of shapes (1.2 s loops: the chip settles at its power cap), to separate shape effects
from in-situ effects: the N = 11008 GEMMs of the decoder (dact = dx Wd, dW_down) run 15-24 % below their N = 12288 / 4096 peers inside the step
(profiles/r04b_gemm_by_shape.json).   python3 tools/gemm_shape_probe.py [MxNxK ...]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "handsonvlm-release_amd"))
import torch
from hvlm import ops

shapes = [tuple(int(v) for v in s.split("x")) for s in sys.argv[1:]] or [
    (13872, 12288, 4096), (13872, 11008, 4096), (13872, 11264, 4096), (13872, 10240, 4096), (13872, 22016, 4096),
    (4096, 11008, 13888), (4096, 11264, 13888), (4096, 12288, 13888), (12288, 4096, 13888), (13872, 4096, 11008), (13872, 4096, 12288)]
g = torch.Generator(device="cuda").manual_seed(0)
for M, N, K in shapes:
    a = (torch.randn(M, K, device="cuda", generator=g) * 0.5).to(torch.bfloat16)
    b = (torch.randn(N, K, device="cuda", generator=g) * 0.05).to(torch.bfloat16)
    c = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
    for _ in range(5): ops.gemm_nt(a, b, out=c)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); ops.gemm_nt(a, b, out=c); e1.record(); torch.cuda.synchronize()
    n = max(10, int(1200.0 / max(1e-3, e0.elapsed_time(e1))))
    e0.record()
    for _ in range(n): ops.gemm_nt(a, b, out=c)
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / n
    tiles = ((M + 255) // 256) * ((N + 255) // 256)
    print(f"{M:6d} x {N:6d} x {K:6d}   {us:8.1f} us   {2.0 * M * N * K / us / 1e6:7.0f} TFLOP/s   tiles {tiles:5d} = {tiles / 256:5.2f} rounds of 256", flush=True)
    del a, b, c
