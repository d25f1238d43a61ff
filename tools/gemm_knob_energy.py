#!/usr/bin/env python3
"""Settled-chip sweep (1.2 s loops, time + joules per call) of the run-time GEMM knobs on the step's main shapes: gemm_group_m (row panels per XCD-aware tile group)
and the static / dynamic tile scheduler.  The round-4 sweep (profiles/r04m) used short loops."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "handsonvlm-release_amd"))
import torch
from hvlm import _lib, ops
import bench
ec = bench.EnergyCounter(0)
BF, F16 = torch.bfloat16, torch.float16
shapes = [(13872, 12288, 4096, BF), (13872, 22016, 4096, BF), (13872, 4096, 11008, BF), (13872, 4096, 4096, BF), (22016, 4096, 13888, BF), (12288, 4096, 13888, BF),
          (411200, 3072, 1024, F16), (411200, 4096, 1024, F16), (411200, 1024, 1024, F16), (411200, 1024, 4096, F16)]
g = torch.Generator(device="cuda").manual_seed(0)
def measure(fn, seconds=1.2):
    for _ in range(4): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); fn(); e1.record(); torch.cuda.synchronize()
    n = max(8, int(seconds * 1e3 / max(1e-3, e0.elapsed_time(e1))))
    j0 = ec.joules(); e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize(); j1 = ec.joules()
    return e0.elapsed_time(e1) * 1e3 / n, (j1 - j0) / n * 1e3
for M, N, K, dt in shapes:
    a = (torch.randn(M, K, device="cuda", generator=g) * 0.5).to(dt); b = (torch.randn(N, K, device="cuda", generator=g) * 0.05).to(dt)
    c = torch.empty(M, N, device="cuda", dtype=dt)
    bias = torch.zeros(N, device="cuda", dtype=dt)
    fn = (lambda: ops.gemm_nt(a, b, out=c)) if dt == BF else (lambda: ops.gemm_f16_nt(a, b, bias=bias, out=c))
    row = []
    for gm in (4, 2, 8, 16, 4):
        _lib.configure("gemm_group_m", gm)
        us, mj = measure(fn)
        row.append(f"gm={gm}: {us:7.1f} us {mj:7.1f} mJ {2.0 * M * N * K / us / 1e6:6.0f} TF/s")
    _lib.configure("gemm_group_m", 4)
    print(f"{M:6d}x{N:5d}x{K:5d} {'bf16' if dt == BF else 'f16 '} | " + " | ".join(row), flush=True)
    del a, b, c
