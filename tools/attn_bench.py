#!/usr/bin/env python3
"""Attention forward / backward timing at the decoder's SFT shape (GPU box): B=16, S=867, H=32, dh=128, causal."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "handsonvlm-release_amd"))
import torch
from hvlm import ops
B, S, H, dh = 16, 867, 32, 128
qkv = (torch.randn(B * S, 3 * H * dh, device="cuda") * 0.5).to(torch.bfloat16)
o, lse = ops.attn_fwd(qkv, B, S, H, dh, True, dh ** -0.5)
do = torch.randn_like(o)
def t(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    a = torch.cuda.Event(enable_timing=True); b = torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n
print("fwd %.1f us" % (1e3 * t(lambda: ops.attn_fwd(qkv, B, S, H, dh, True, dh ** -0.5))))
print("bwd %.1f us" % (1e3 * t(lambda: ops.attn_bwd(qkv, o, do, lse, B, S, H, dh, True, dh ** -0.5))))
