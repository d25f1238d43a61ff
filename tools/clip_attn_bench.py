#!/usr/bin/env python3
"""CLIP attention forward (non-causal, dh = 64, S = 257, 16 heads) at the SFT step's size (1600 frames per pass) and a smaller one,
fp16 (the tower's dtype) and bf16, checked against an fp32 softmax on a subset of (frame, head) items.  TFLOP/s are algorithmic
(4 S^2 dh H B).  With a library that has the switch `attn_clip_pipelined` (round 3's software-pipelined 32-key-block loop, measured
2-4.5 % slower and removed: profiles/r03f_clip_attention_pipelined_ab.txt) both variants are timed."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "handsonvlm-release_amd"))
import torch
from hvlm import ops, _lib


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
        res = {}
        try:
            _lib.configure("attn_clip_pipelined", 1)
            has = True
        except _lib.HvlmError:
            has = False
        for pipe in (0, 1, 0, 1):
            if has:
                _lib.configure("attn_clip_pipelined", pipe)
            ms = t(fn)
            res[pipe] = min(res.get(pipe, 1e9), ms)
        o1 = fn()
        if has:
            _lib.configure("attn_clip_pipelined", 0)
        o0 = fn()
        if has:
            _lib.configure("attn_clip_pipelined", 1)
        x = qkv.view(B, S, 3, H, dh)[:3].float().permute(2, 0, 3, 1, 4)          # [3, b, H, S, dh]
        ref = torch.softmax((x[0] @ x[1].transpose(-1, -2)) * dh ** -0.5, -1) @ x[2]
        ref = ref.permute(0, 2, 1, 3).reshape(3 * S, H * dh)
        e1 = float((o1[:3 * S].float() - ref).norm() / ref.norm()); e0 = float((o0[:3 * S].float() - ref).norm() / ref.norm())
        d = float((o1.float() - o0.float()).abs().max())
        print(f"B={B} {str(dt)[6:]}: tile loop {res[0]*1e3:.0f} us ({fl/res[0]/1e9:.0f} TF/s)  pipelined {res[1]*1e3:.0f} us ({fl/res[1]/1e9:.0f} TF/s)  "
              f"{(res[0]/res[1]-1)*100:+.1f} %   rel-L2 vs fp32: {e0:.2e} / {e1:.2e}   max |a-b| {d:.2e}", flush=True)
