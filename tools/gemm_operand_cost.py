#!/usr/bin/env python3
"""VERDICT r5 item 1, the measurement it asks for before the 256 x 128 re-tiling is built: what do 1.5 x the operand bytes per flop cost this
GEMM, in time and in joules, at equal MFMA work?  DEV library (HVLM_LIB=handsonvlm-release_amd/lib/libhvlm_hip_dev.so): the ablation kernel
gemm_pp4_dbg_kernel (the 32x32x16 form of the production schedule) with gemm_dbg = 32 (nothing changed) against gemm_dbg = 48 (the B half-tiles
staged twice: 12 instead of 8 LDS-DMA pieces per K-tile and wave = the L2 -> LDS stream of a 256 x 128 tile; same slots, same bytes, results
unchanged), 1.5 s settled loops with the socket's energy accumulator, A B A B.  bf16 operands (the ablation kernel is the bf16 instantiation);
the tower's K = 1024 shapes at M = 411 200 and two decoder shapes."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "handsonvlm-release_amd"))
import torch
from hvlm import _lib, ops
import bench
assert _lib.load().hvlm_is_dev_build() == 1, "needs the DEV library (make DEV=1; HVLM_LIB=.../libhvlm_hip_dev.so)"
ec = bench.EnergyCounter(0)
BF = torch.bfloat16
shapes = [(411200, 3072, 1024), (411200, 1024, 1024), (411200, 4096, 1024), (13872, 12288, 4096), (13872, 4096, 4096)]
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


for M, N, K in shapes:
    a = (torch.randn(M, K, device="cuda", generator=g) * 0.5).to(BF); b = (torch.randn(N, K, device="cuda", generator=g) * 0.05).to(BF)
    c = torch.empty(M, N, device="cuda", dtype=BF)
    fn = lambda: ops.gemm_nt(a, b, out=c)  # noqa: E731
    _lib.configure("gemm_dbg", 0); fn(); ref = c.clone()
    row = []
    for dbg in (32, 48, 32, 48, 0):
        _lib.configure("gemm_dbg", dbg)
        us, mj = measure(fn)
        same = bool(torch.equal(c, ref)) if dbg else True
        row.append((dbg, us, mj, same))
    _lib.configure("gemm_dbg", 0)
    base = [r for r in row if r[0] == 32]; dup = [r for r in row if r[0] == 48]
    bu, bm = sum(r[1] for r in base) / 2, sum(r[2] for r in base) / 2
    du, dm = sum(r[1] for r in dup) / 2, sum(r[2] for r in dup) / 2
    fl = 2.0 * M * N * K
    print(f"{M:6d}x{N:5d}x{K:5d} bf16 | 1.0x operands: {bu:8.1f} us {bm:8.1f} mJ {fl / bu / 1e6:6.0f} TF/s | 1.5x (B staged twice): {du:8.1f} us {dm:8.1f} mJ "
          f"{fl / du / 1e6:6.0f} TF/s | time x{du / bu:.3f}  joules x{dm / bm:.3f} | production 16x16x32 kernel: {row[-1][1]:8.1f} us {row[-1][2]:8.1f} mJ | "
          f"results unchanged: {all(r[3] for r in row)}", flush=True)
    del a, b, c
