#!/usr/bin/env python3
"""CLIP attention (fp16, 400 frames x 257 tokens, 16 heads, dh = 64) forward in isolation: the resident-K/V kernel in its
software-pipelined form (default) and the plain form (HVLM_ATTN_RES_PIPE=0), and the error of both against fp32 torch."""
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
for (B, S, H, dh) in [(400, 257, 16, 64), (64, 200, 16, 64), (64, 33, 16, 64)]:
    qkv = (torch.randn(B * S, 3 * H * dh, device="cuda") * 0.5).to(torch.float16)
    ms = t(lambda: ops.attn_fwd_f16(qkv, B, S, H, dh, dh ** -0.5))
    o = ops.attn_fwd_f16(qkv, B, S, H, dh, dh ** -0.5).float().view(B, S, H, dh)
    q, k, v = (x.float().view(B, S, H, dh).permute(0, 2, 1, 3)[:8] for x in qkv.split(H * dh, dim=1))
    ref = torch.softmax(q @ k.transpose(-1, -2) * dh ** -0.5, -1) @ v
    err = (o[:8].permute(0, 2, 1, 3) - ref).norm() / ref.norm()
    print(f"B={B} S={S}: {ms*1e3:.0f} us  {4*S*S*dh*H*B/ms/1e9:.0f} TF/s (algorithmic)  rel err vs fp32 {err:.2e}")
