#!/usr/bin/env python3
"""GPU box: the hand-scheduled CLIP attention (csrc/asm/attn_clip_d64.py: fp16, S = 257, dh = 64, non-causal) against the C++ resident
kernel and an fp32 torch reference; then timing of both at the tower's shape (1600 frames x 16 heads).  [--quick] [--time]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "handsonvlm-release_amd"))
import torch
from hvlm import ops, _lib
F16 = torch.float16

def ref(qkv, B, H):
    x = qkv.float().view(B, 257, 3, H, 64)
    q, k, v = x[:, :, 0].transpose(1, 2), x[:, :, 1].transpose(1, 2), x[:, :, 2].transpose(1, 2)
    s = (q @ k.transpose(-1, -2)) * 64 ** -0.5
    return (torch.softmax(s, -1) @ v).transpose(1, 2).reshape(B * 257, H * 64)

def run(B, H, seed=0, spike=False, scale=1.0):
    g = torch.Generator(device="cuda").manual_seed(seed)
    qkv = (torch.randn(B * 257, 3 * H * 64, device="cuda", generator=g) * scale).to(F16)
    if spike:
        x = qkv.view(B, 257, 3, H, 64)
        for t in range(70, 257, 61): x[:, t, 1] *= 5.0
    _lib.configure("attn_clip_asm", 0); o0 = ops.attn_fwd_f16(qkv, B, 257, H, 64, 64 ** -0.5)
    _lib.configure("attn_clip_asm", 1); o1 = ops.attn_fwd_f16(qkv, B, 257, H, 64, 64 ** -0.5)
    torch.cuda.synchronize()
    _lib.configure("attn_clip_asm", 0)
    orf = ref(qkv, B, H)
    e0, e1 = (o0.float() - orf).abs().max().item(), (o1.float() - orf).abs().max().item()
    import math
    ulp = 2.0 ** (math.floor(math.log2(orf.abs().max().item())) - 10)       # fp16 ulp of the largest output (tests/test_gpu_r5.py)
    ok = e1 <= max(2.0 * e0, 2e-3, 1.5 * ulp) and torch.isfinite(o1.float()).all().item()
    print("B=%4d H=%2d spike=%d: O err asm %.3e (c++ %.3e) %s" % (B, H, spike, e1, e0, "ok" if ok else "FAIL"), flush=True)
    if not ok:
        d = (o1.float() - orf).abs().view(B, 257, H, 64)
        print("   bad rows:", (d.amax(dim=(0, 2, 3)) > max(2 * e0, 2e-3)).nonzero().flatten().tolist()[:40])
        print("   bad frames:", (d.amax(dim=(1, 2, 3)) > max(2 * e0, 2e-3)).nonzero().flatten().tolist()[:20], " bad heads:", (d.amax(dim=(0, 1, 3)) > max(2 * e0, 2e-3)).nonzero().flatten().tolist()[:20])
        print("   bad dims:", (d.amax(dim=(0, 1, 2)) > max(2 * e0, 2e-3)).nonzero().flatten().tolist()[:64])
    return ok

def timeit(fn, n=10):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    a = torch.cuda.Event(enable_timing=True); b = torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n

ok = True
for (B, H) in [(1, 1), (1, 2), (2, 3), (3, 16), (17, 16)] + ([] if "--quick" in sys.argv else [(300, 16), (1600, 16)]):
    ok &= run(B, H, seed=B)
    if not ok and "--quick" in sys.argv: break
if ok:
    ok &= run(5, 16, seed=3, spike=True, scale=1.5)
print("ALL OK" if ok else "SOME FAILED", flush=True)
if "--time" in sys.argv and ok:
    B, H = 1600, 16
    qkv = (torch.randn(B * 257, 3 * H * 64, device="cuda") * 0.5).to(F16)
    fl = 4.0 * 257 * 257 * 64 * H * B
    for asm in (0, 1):
        _lib.configure("attn_clip_asm", asm)
        t = timeit(lambda: ops.attn_fwd_f16(qkv, B, 257, H, 64, 64 ** -0.5))
        print("clip attention 1600 x 16, %s: %.1f us (%.0f TF/s)" % ("asm" if asm else "c++", 1e3 * t, fl / t / 1e9), flush=True)
    _lib.configure("attn_clip_asm", 0)
sys.exit(0 if ok else 1)
