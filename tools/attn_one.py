#!/usr/bin/env python3
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "handsonvlm-release_amd"))
import torch
from hvlm import ops
B, S, H, dh, causal = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]), bool(int(sys.argv[5]))
it = int(sys.argv[6]) if len(sys.argv) > 6 else 3
qkv = (torch.randn(B * S, 3 * H * dh, device="cuda") * 0.5).to(torch.bfloat16)
for _ in range(it): o, lse = ops.attn_fwd(qkv, B, S, H, dh, causal, dh ** -0.5)
if len(sys.argv) > 7:
    do = torch.randn_like(o)
    for _ in range(it): ops.attn_bwd(qkv, o, do, lse, B, S, H, dh, causal, dh ** -0.5)
torch.cuda.synchronize()
