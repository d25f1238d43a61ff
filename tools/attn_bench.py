#!/usr/bin/env python3
"""Attention forward / backward timing at the decoder's SFT shapes (GPU box): B=16, S=867, H=32, dh=128 (7B headline) and
B=4, S=2048, H=40 (13B), causal and — to separate the cost of the diagonal — non-causal.  TFLOP/s are algorithmic
(4*S^2*dh*H*B per forward, x0.5 causal; backward = 2.5x forward)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "handsonvlm-release_amd"))
import torch
from hvlm import ops
def t(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    a = torch.cuda.Event(enable_timing=True); b = torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n
shapes = [(16, 867, 32, 128)] if "--all" not in sys.argv else [(16, 867, 32, 128), (4, 2048, 40, 128), (16, 512, 32, 128)]
for (B, S, H, dh) in shapes:
    qkv = (torch.randn(B * S, 3 * H * dh, device="cuda") * 0.5).to(torch.bfloat16)
    for causal in ((True, False) if "--all" in sys.argv else (True,)):
        o, lse = ops.attn_fwd(qkv, B, S, H, dh, causal, dh ** -0.5)
        do = torch.randn_like(o)
        fl = 4.0 * S * S * dh * H * B * (0.5 if causal else 1.0)
        tf = t(lambda: ops.attn_fwd(qkv, B, S, H, dh, causal, dh ** -0.5))
        tb = t(lambda: ops.attn_bwd(qkv, o, do, lse, B, S, H, dh, causal, dh ** -0.5))
        print("B=%d S=%d H=%d causal=%d: fwd %.1f us (%.0f TF/s)  bwd %.1f us (%.0f TF/s)" % (B, S, H, causal, 1e3 * tf, fl / tf / 1e9, 1e3 * tb, 2.5 * fl / tb / 1e9))
