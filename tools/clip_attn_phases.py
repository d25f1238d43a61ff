#!/usr/bin/env python3
"""Per-wave cycle split of the resident CLIP attention kernel (needs a library built with -DHVLM_ATTN_PROF, where the `lse`
argument receives 8 counters per wave: barrier wait, tile loop, epilogue, total, QK^T / softmax / P V phases of the four full tiles).  HVLM_LIB=... python tools/clip_attn_phases.py"""
import os, sys, ctypes
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "handsonvlm-release_amd"))
import torch
from hvlm import ops, _lib

S, H, dh = 257, 16, 64
B = int(sys.argv[1]) if len(sys.argv) > 1 else 1600
qkv = (torch.randn(B * S, 3 * H * dh, device="cuda") * 0.7).half()
lib = _lib.load()
o = torch.empty(B * S, H * dh, dtype=torch.float16, device="cuda")
ws = torch.empty(max(1, lib.hvlm_attn_fwd_ws_bytes(B, S, H, dh)), dtype=torch.uint8, device="cuda")
prof = torch.zeros(256 * 10 * 8, dtype=torch.int64, device="cuda")
st = torch.cuda.current_stream().cuda_stream
for _ in range(3):
    rc = lib.hvlm_attn_fwd_f16(qkv.data_ptr(), o.data_ptr(), prof.data_ptr(), ws.data_ptr(), B, S, H, dh, qkv.stride(0), 0, dh ** -0.5, st)
    assert rc == 0
torch.cuda.synchronize()
a = torch.cuda.Event(enable_timing=True); b = torch.cuda.Event(enable_timing=True)
a.record()
lib.hvlm_attn_fwd_f16(qkv.data_ptr(), o.data_ptr(), prof.data_ptr(), ws.data_ptr(), B, S, H, dh, qkv.stride(0), 0, dh ** -0.5, st)
b.record(); torch.cuda.synchronize()
us = a.elapsed_time(b) * 1e3
p = prof.view(256, 10, 8).double().cpu()
items = B * H / 256
print(f"B={B}: {us:.0f} us, {items:.0f} items per workgroup; cycles per item (mean over workgroups), counter ticks per us = {p[:, 0, 3].mean() / us:.0f}")
print("wave   barrier    tiles  epilogue     total |  per full tile: QK^T  softmax     P V")
for w in range(10):
    m = p[:, w].mean(0) / items
    print(f"{w:4d} {m[0]:9.0f} {m[1]:8.0f} {m[2]:9.0f} {m[3]:9.0f} | {m[4] / 4:19.0f} {m[5] / 4:8.0f} {m[6] / 4:7.0f}")
