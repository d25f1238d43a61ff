#!/usr/bin/env python3
"""How the persistent GEMM behaves when a few CUs are taken by another stream (an RCCL all-reduce in data-parallel
training): time of one GEMM alone vs beside `hvlm_debug_occupy` holding N CUs for the whole duration (GPU box; the occupier
exists in development builds only: run with HVLM_LIB=handsonvlm-release_amd/lib/libhvlm_hip_dev.so)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "handsonvlm-release_amd"))
import ctypes
import torch
from hvlm import _lib, ops
dev = torch.device("cuda:0"); BF = torch.bfloat16
lib = _lib.load()
side = torch.cuda.Stream()
for (M, N, K) in [(13872, 12288, 4096), (13872, 4096, 11008), (4096, 4096, 13888)]:
    a = torch.rand(M, K, device=dev).sub_(0.5).to(BF); b = torch.rand(N, K, device=dev).sub_(0.5).to(BF)
    out = torch.empty(M, N, dtype=BF, device=dev)
    for _ in range(3): ops.gemm_nt(a, b, out=out)
    torch.cuda.synchronize()
    row = {}
    for occ in (0, 8, 16, 32, 64):
        ts = []
        for rep in range(5):
            torch.cuda.synchronize()
            if occ:
                _lib.check(lib.hvlm_debug_occupy(occ, 20_000_000, ctypes.c_void_p(side.cuda_stream)), "occupy")   # ~10 ms
                torch.cuda._sleep(200_000)      # let the occupier get its CUs first
            s = torch.cuda.Event(enable_timing=True); e = torch.cuda.Event(enable_timing=True)
            s.record(); ops.gemm_nt(a, b, out=out); e.record()
            torch.cuda.synchronize()
            ts.append(s.elapsed_time(e))
        row[occ] = round(sorted(ts)[len(ts) // 2] * 1e3)
    print(f"{M}x{N}x{K}: GEMM us with N CUs occupied elsewhere: {row}", flush=True)
