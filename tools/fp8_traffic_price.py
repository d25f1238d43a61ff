#!/usr/bin/env python3
"""price the e4m3 GEMM's fabric traffic (3.9 x its algorithmic bytes on the batch-32 prefill shapes) the way
profiles/r05n_traffic.txt priced the bf16 one: the run-time knob `gemm_group_m` (row panels per XCD-aware tile group) moves the L2-miss
traffic of the same kernel on the same problem; 1.5 s settled loops give time AND joules per call per setting.  If joules follow the
traffic, the bytes are worth removing; if they do not, they are memory-side-cache hits.  (FETCH_SIZE per setting: tools/gemm_traffic.py
under rocprofv3 --pmc; this tool needs no profiler.)  Shapes: config 5's prefill, M = 32 x 867."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "handsonvlm-release_amd"))
import torch
from hvlm import _lib, ops
import bench
ec = bench.EnergyCounter(0)
M = 32 * 867
shapes = [(M, 12288, 4096), (M, 4096, 4096), (M, 22016, 4096), (M, 4096, 11008)]
g = torch.Generator(device="cuda").manual_seed(0)


def measure(fn, seconds=1.5):
    for _ in range(4): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); fn(); e1.record(); torch.cuda.synchronize()
    n = max(8, int(seconds * 1e3 / max(1e-3, e0.elapsed_time(e1))))
    j0 = ec.joules(); e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize(); j1 = ec.joules()
    return e0.elapsed_time(e1) * 1e3 / n, ((j1 - j0) / n * 1e3) if (j0 is not None and j1 is not None) else float("nan")


for Mx, N, K in shapes:
    a = torch.randint(0, 120, (Mx, K), device="cuda", generator=g, dtype=torch.uint8)       # e4m3 bytes: finite, mixed magnitudes
    b = torch.randint(0, 120, (N, K), device="cuda", generator=g, dtype=torch.uint8)
    rs = torch.rand(Mx, device="cuda", generator=g) + 0.5; cs = torch.rand(N, device="cuda", generator=g) + 0.5
    c = torch.empty(Mx, N, device="cuda", dtype=torch.bfloat16)
    fn = lambda: ops.gemm_fp8_nt_scaled(a, b, rs, cs, out=c)  # noqa: E731
    row = []
    for gm in (4, 2, 8, 1, 4):
        _lib.configure("gemm_group_m", gm)
        us, mj = measure(fn)
        row.append(f"group_m={gm}: {us:7.1f} us {mj:7.1f} mJ {2.0 * Mx * N * K / us / 1e6:6.0f} TF/s")
    _lib.configure("gemm_group_m", 4)
    alg = (Mx * K + N * K + 2 * Mx * N) / 1e9
    print(f"{Mx:6d}x{N:5d}x{K:5d} e4m3 (algorithmic {alg:.2f} GB) | " + " | ".join(row), flush=True)
    del a, b, c
