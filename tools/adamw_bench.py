"""hvlm_adamw_step at the decoder's weight shapes: microseconds and HBM bytes/s (30 B per parameter with the transposed bf16
copy: master, m, v read + written in fp32, gradient read, weight and transposed weight written in bf16; 28 B without).
    python tools/adamw_bench.py [--no-wt]"""
import argparse, os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "handsonvlm-release_amd"))
import torch
from hvlm import ops

ap = argparse.ArgumentParser()
ap.add_argument("--no-wt", action="store_true")
ap.add_argument("--iters", type=int, default=20)
a = ap.parse_args()
shapes = [(4096, 4096), (12288, 4096), (11008, 4096), (22016, 4096), (4096, 11008), (32101, 4096), (5120, 5120), (13824, 5120), (1, 4096)]
for R, C in shapes:
    n = R * C
    master = torch.randn(n, device="cuda").view(R, C)
    m = torch.zeros_like(master); v = torch.zeros_like(master)
    g = torch.randn(R, C, device="cuda").bfloat16()
    w = master.bfloat16()
    wt = None if a.no_wt or R == 1 else torch.empty(C, R, dtype=torch.bfloat16, device="cuda")
    for _ in range(3):
        ops.adamw_step_(master, m, v, g, w, wt, 1e-4, 0.9, 0.999, 1e-8, 0.0, 1.0, 1)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(a.iters):
        ops.adamw_step_(master, m, v, g, w, wt, 1e-4, 0.9, 0.999, 1e-8, 0.0, 1.0, 2 + i)
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / a.iters
    bpp = 30 if wt is not None else 28
    print(f"{R:6d} x {C:6d}  wt={'yes' if wt is not None else 'no '}  {us:8.1f} us   {n * bpp / us / 1e6:7.2f} TB/s")
