#!/usr/bin/env python3
"""CLIP attention forward (non-causal, dh = 64, S = 257, 16 heads) at the SFT step's size (1600 frames per pass) and a smaller one,
fp16 (the tower's dtype) and bf16, checked against an fp32 softmax on a subset of (frame, head) items.  TFLOP/s are algorithmic
(4 S^2 dh H B).  Two timing passes per case (the clock of an otherwise idle GPU ramps during the first launches; for an in-situ
figure use the rocprofv3 kernel stats of bench.py: tools/collect_round.sh <tag> stats).
    python tools/clip_attn_bench.py            HVLM_LIB=<other build> python tools/clip_attn_bench.py   (same-box A/B)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "handsonvlm-release_amd"))
import torch
from hvlm import ops


def t(fn, n=10):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    a = torch.cuda.Event(enable_timing=True); b = torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n


S, H, dh = 257, 16, 64
for B in (400, 1600):
    for dt in (torch.float16, torch.bfloat16):
        qkv = (torch.randn(B * S, 3 * H * dh, device="cuda") * 0.7).to(dt)
        fn = (lambda: ops.attn_fwd_f16(qkv, B, S, H, dh, dh ** -0.5)) if dt == torch.float16 else \
             (lambda: ops.attn_fwd(qkv, B, S, H, dh, False, dh ** -0.5, need_lse=False)[0])
        fl = 4.0 * S * S * dh * H * B
        ms = [t(fn), t(fn)]
        o = fn()
        x = qkv.view(B, S, 3, H, dh)[:3].float().permute(2, 0, 3, 1, 4)          # [3, b, H, S, dh]
        ref = torch.softmax((x[0] @ x[1].transpose(-1, -2)) * dh ** -0.5, -1) @ x[2]
        ref = ref.permute(0, 2, 1, 3).reshape(3 * S, H * dh)
        e = float((o[:3 * S].float() - ref).norm() / ref.norm())
        print(f"B={B} {str(dt)[6:]}: {ms[0]*1e3:.0f} / {ms[1]*1e3:.0f} us ({fl/min(ms)/1e9:.0f} TF/s)   rel-L2 vs fp32 (3 frames): {e:.2e}   "
              f"finite: {bool(torch.isfinite(o.float()).all())}", flush=True)
