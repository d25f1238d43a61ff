#!/usr/bin/env python3
"""A/B of the GEMM epilogue (DEV build: HVLM_LIB=handsonvlm-release_amd/lib/libhvlm_hip_dev.so): the register-exchange epilogue
(v_permlane swaps, gemm_epi_swap = 1, production) against the round-2 epilogue through the wave-private LDS buffers
(gemm_epi_swap = 0) on the GEMM shapes of the 7B SFT step — outputs must be bit-identical, times are printed per shape.
Optional argv[1]: another integer switch to A/B instead (e.g. gemm_tail_half)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "handsonvlm-release_amd"))
import torch
from hvlm import ops, _lib
KEY = sys.argv[1] if len(sys.argv) > 1 else "gemm_tail_half"
dev = "cuda"
BF, F16 = torch.bfloat16, torch.float16
# (M, N, K, dtype, bias, act, residual, label)
SHAPES = [
    (411200, 3072, 1024, F16, True, 0, False, "CLIP qkv"),
    (411200, 1024, 1024, F16, True, 0, True, "CLIP out_proj"),
    (411200, 4096, 1024, F16, True, 1, False, "CLIP fc1+QuickGELU"),
    (411200, 1024, 4096, F16, True, 0, True, "CLIP fc2"),
    (13872, 12288, 4096, BF, False, 0, False, "qkv"),
    (13872, 4096, 4096, BF, False, 0, True, "o_proj"),
    (13872, 22016, 4096, BF, False, 0, False, "gate|up"),
    (13872, 4096, 11008, BF, False, 0, True, "down"),
    (13872, 11008, 4096, BF, False, 0, False, "dX down"),
    (12288, 4096, 13888, BF, False, 0, False, "dW qkv"),
    (22016, 4096, 13888, BF, False, 0, False, "dW gate|up"),
    (4096, 4096, 13888, BF, False, 0, False, "dW o"),
    (13872, 32128, 4096, BF, False, 0, False, "lm_head"),
]


def timeit(fn, iters=6, warmup=2):
    for _ in range(warmup):
        fn()
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(2):
        t0 = time.perf_counter()
        for _ in range(iters):
            fn()
        torch.cuda.synchronize()
        best = min(best, (time.perf_counter() - t0) / iters)
    return best


tot = {0: 0.0, 1: 0.0}
for (M, N, K, dt, hb, act, hr, label) in SHAPES:
    g = torch.Generator(device=dev).manual_seed(M + N + K)
    a = torch.rand(M, K, device=dev, generator=g).sub_(0.5).to(dt)
    b = torch.rand(N, K, device=dev, generator=g).sub_(0.5).to(dt)
    bias = torch.rand(N, device=dev, generator=g).to(dt) if hb else None
    res = torch.rand(M, N, device=dev, generator=g).to(dt) if hr else None
    fn = ops.gemm_f16_nt if dt == F16 else ops.gemm_nt
    outs, ts = {}, {}
    for v in (0, 1, 0, 1):
        _lib.configure(KEY, v)
        out = torch.empty(M, N, dtype=dt, device=dev)
        t = timeit(lambda: fn(a, b, bias=bias, residual=res, act=act, out=out))
        ts[v] = min(ts.get(v, 1e9), t)
        outs[v] = out
    same = bool(torch.equal(outs[0], outs[1]))
    fl = 2.0 * M * N * K
    tot[0] += ts[0]; tot[1] += ts[1]
    print(f"{label:20s} {M}x{N}x{K}: {KEY}=0 {ts[0]*1e6:8.0f} us {fl/ts[0]/1e12:7.1f} TF/s | =1 {ts[1]*1e6:8.0f} us {fl/ts[1]/1e12:7.1f} TF/s | "
          f"{(ts[0]/ts[1]-1)*100:+.1f} %  bit-identical={same}", flush=True)
    del a, b, out, outs
print(f"sum: {tot[0]*1e3:.2f} ms -> {tot[1]*1e3:.2f} ms")
