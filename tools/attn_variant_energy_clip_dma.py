#!/usr/bin/env python3
"""Settled-chip A/B (1.5 s loops, time + joules per call) of the CLIP kernel's DMA spread: the next item's 18 LDS-DMA pieces over 1 / 2 / 3 / 5 (production) / 9 steps."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "handsonvlm-release_amd"))
import torch
from hvlm import _lib
import bench
lib = _lib.load(); assert lib.hvlm_is_dev_build()
ec = bench.EnergyCounter(0)
CLIP = ["production", "prof", "nosm", "nodma", "nolds", "nomfma", "dma1", "dma2", "dma5", "dma9", "split", "dotsum", "eagermax"]
st = torch.cuda.current_stream().cuda_stream
B, H = 1600, 16
qkv = torch.randn(B * 257, 3 * H * 64, device="cuda").to(torch.float16)
out = torch.empty(B * 257, H * 64, device="cuda", dtype=torch.float16)
for rep in range(2):
    for name, label in (("production", "5 steps (production)"), ("dma1", "1 step"), ("dma2", "2 steps"), ("dma5", "3 steps"), ("dma9", "9 steps")):
        v = CLIP.index(name)
        fn = lambda: lib.hvlm_attn_clip_asm_dev(qkv.data_ptr(), out.data_ptr(), None, B, H, qkv.stride(0), 64 ** -0.5, v, st)
        for _ in range(5): fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        n = 1600
        j0 = ec.joules(); e0.record()
        for _ in range(n): fn()
        e1.record(); torch.cuda.synchronize(); j1 = ec.joules()
        print(f"pass {rep}  DMA over {label:22s} {e0.elapsed_time(e1) * 1e3 / n:8.1f} us/call  {(j1 - j0) / n * 1e3:8.1f} mJ/call", flush=True)
