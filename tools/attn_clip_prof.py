#!/usr/bin/env python3
"""GPU box, development library: where the cycles of the hand-scheduled CLIP attention go — kernel time of the production schedule
and of every timing-only ablation (wrong results by construction), and the cycle sums of the stamped schedule per wave
(item start wait / item body / epilogue)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.environ.setdefault("HVLM_LIB", os.path.join(ROOT, "handsonvlm-release_amd", "lib", "libhvlm_hip_dev.so"))
sys.path.insert(0, os.path.join(ROOT, "handsonvlm-release_amd"))
import torch
import numpy as np
from hvlm import ops, _lib
lib = _lib.load()
VARIANTS = ["production", "prof", "nosm", "nodma", "nolds", "nomfma", "dma1", "dma2", "dma5", "dma9", "split", "dotsum", "eagermax"]
def timeit(fn, n=10):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    a = torch.cuda.Event(enable_timing=True); b = torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3
B, H = 1600, 16
qkv = (torch.randn(B * 257, 3 * H * 64, device="cuda") * 0.5).to(torch.float16)
out = torch.empty(B * 257, H * 64, device="cuda", dtype=torch.float16)
prof = torch.zeros(256 * 4 * 8, device="cuda", dtype=torch.int32)
st = torch.cuda.current_stream().cuda_stream
def run(v, p=None):
    rc = lib.hvlm_attn_clip_asm_dev(qkv.data_ptr(), out.data_ptr(), p.data_ptr() if p is not None else None, B, H, qkv.stride(0), 64 ** -0.5, v, st)
    assert rc == 0, _lib.last_error()
fl = 4.0 * 257 * 257 * 64 * H * B
_lib.configure("attn_clip_asm", 0)
t = timeit(lambda: ops.attn_fwd_f16(qkv, B, 257, H, 64, 64 ** -0.5))
print(flush=True, *[f"c++ resident kernel   {t:8.1f} us  {fl / t / 1e6:6.0f} TF/s"])
for v, name in enumerate(VARIANTS):
    t = timeit(lambda: run(v))
    print(flush=True, *[f"asm {name:12s}      {t:8.1f} us  {fl / t / 1e6:6.0f} TF/s"])
names = ["item start wait (vmcnt + barrier)", "setup + item body", "epilogue + buffer swap"]
for v, label in ((1, "production (class row on wave 3)"), (VARIANTS.index("split"), "split (class row shared by the four waves)")):
    prof.zero_(); run(v, prof); torch.cuda.synchronize()
    p = (prof.cpu().numpy().astype("int64") & 0xffffffff).reshape(256 * 4, 8)
    print(f"cycle stamps, {label}:")
    for w in range(4):
        pw = p[w::4]
        print(flush=True, *[f"  wave {w}: " + "  ".join(f"{names[k]} {pw[:, k].mean() / 100:8.0f}" for k in range(3)) + f"   per item {pw[:, :3].sum(1).mean() / 100:8.0f} cycles"])
