#!/usr/bin/env python3
"""GPU box: the hand-scheduled dh = 128 causal forward (attention_asm.hip) against the C++ kernel it replaces and against an
fp32 torch reference, shape by shape (small first: a wrong kernel should fail on a 32-row problem, not hang a big one), then
timing of both.  Usage: attn_asm_check.py [--quick] [--time]"""
import os, sys, ctypes
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "handsonvlm-release_amd"))
import torch
from hvlm import ops, _lib

lib = _lib.load()

def set_asm(v):
    assert lib.hvlm_configure(b"attn_fwd_asm", int(v)) == 0

def ref(qkv, B, S, H, dh, lens=None):
    """fp32 reference on the GPU (per sample)."""
    x = qkv.float().view(B, S, 3, H, dh)
    q, k, v = x[:, :, 0].transpose(1, 2), x[:, :, 1].transpose(1, 2), x[:, :, 2].transpose(1, 2)
    s = (q @ k.transpose(-1, -2)) * dh ** -0.5
    mask = torch.ones(S, S, device=qkv.device, dtype=torch.bool).tril()
    s = s.masked_fill(~mask, float("-inf"))
    lse = torch.logsumexp(s, -1)
    o = torch.softmax(s, -1) @ v
    return o.transpose(1, 2).reshape(B * S, H * dh), lse

def run(B, S, H, dh=128, seed=0, spike=False):
    g = torch.Generator(device="cuda").manual_seed(seed)
    qkv = (torch.randn(B * S, 3 * H * dh, device="cuda", generator=g) * 0.7).to(torch.bfloat16)
    if spike:      # late, large maxima: forces the deferred-rescale slow path
        x = qkv.view(B, S, 3, H, dh)
        for t in range(40, S, 97):
            x[:, t, 1] *= 6.0
    set_asm(0); o0, l0 = ops.attn_fwd(qkv, B, S, H, dh, True, dh ** -0.5)
    set_asm(1); o1, l1 = ops.attn_fwd(qkv, B, S, H, dh, True, dh ** -0.5)
    torch.cuda.synchronize()
    orf, lrf = ref(qkv, B, S, H, dh)
    e_cpp = (o0.float() - orf).abs().max().item(); e_asm = (o1.float() - orf).abs().max().item()
    l_cpp = (l0 - lrf).abs().max().item(); l_asm = (l1 - lrf).abs().max().item()
    ok = e_asm <= max(2.0 * e_cpp, 4e-3) and l_asm <= max(2.0 * l_cpp, 2e-3) and torch.isfinite(o1.float()).all().item()
    print("B=%d S=%4d H=%2d spike=%d: O err asm %.3e (c++ %.3e)  lse err asm %.3e (c++ %.3e)  %s" % (B, S, H, spike, e_asm, e_cpp, l_asm, l_cpp, "ok" if ok else "FAIL"), flush=True)
    if not ok:
        d = (o1.float() - orf).abs().view(B, S, H, dh)
        per_row = d.amax(dim=(0, 2, 3))
        bad = (per_row > max(2.0 * e_cpp, 4e-3)).nonzero().flatten().tolist()
        print("   bad rows (first 40):", bad[:40], " count", len(bad))
        per_d = d.amax(dim=(0, 1, 2)); badd = (per_d > max(2.0 * e_cpp, 4e-3)).nonzero().flatten().tolist()
        print("   bad head-dims (first 40):", badd[:40], " count", len(badd))
        dl = (l1 - lrf).abs().amax(dim=(0, 1)); badl = (dl > max(2.0 * l_cpp, 2e-3)).nonzero().flatten().tolist()
        print("   bad lse rows (first 40):", badl[:40], " count", len(badl))
    return ok

def run_varlen(lens, H, dh=128, seed=1):
    B = len(lens); S = max(lens)
    g = torch.Generator(device="cuda").manual_seed(seed)
    full = (torch.randn(B * S, 3 * H * dh, device="cuda", generator=g) * 0.7).to(torch.bfloat16)
    rows = torch.cat([torch.arange(b * S, b * S + n, device="cuda") for b, n in enumerate(lens)])
    packed = full[rows].contiguous()
    cu = torch.tensor([0] + list(torch.tensor(lens).cumsum(0)), dtype=torch.int32, device="cuda")
    set_asm(1)
    op, lp = ops.attn_fwd(packed, B, S, H, dh, True, dh ** -0.5, cu_seqlens=cu)
    of, lf = ops.attn_fwd(full, B, S, H, dh, True, dh ** -0.5)
    torch.cuda.synchronize()
    same_o = torch.equal(op, of[rows])
    same_l = all(torch.equal(lp[b, :, :n], lf[b, :, :n]) for b, n in enumerate(lens))
    print("varlen lens=%s H=%d: packed == padded bit for bit: O %s, lse %s" % (lens, H, same_o, same_l), flush=True)
    return same_o and same_l

def timeit(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    a = torch.cuda.Event(enable_timing=True); b = torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n

ok = True
quick = "--quick" in sys.argv
shapes = [(1, 32, 1), (1, 64, 1), (1, 96, 1), (1, 128, 1), (1, 160, 2), (2, 256, 2), (1, 257, 1), (1, 300, 3), (2, 512, 2), (1, 867, 4)]
if not quick: shapes += [(2, 1024, 4), (1, 2048, 5), (16, 867, 32)]
for (B, S, H) in shapes:
    ok &= run(B, S, H)
    if not ok and quick: break
if ok:
    ok &= run(2, 867, 4, spike=True)
    ok &= run(1, 2048, 3, spike=True)
    ok &= run_varlen([100, 867, 33, 512], 4)
    ok &= run_varlen([867, 862, 830, 747], 3)
print("ALL OK" if ok else "SOME FAILED", flush=True)
if "--time" in sys.argv and ok:
    for (B, S, H) in [(16, 867, 32), (4, 2048, 40), (16, 512, 32)]:
        qkv = (torch.randn(B * S, 3 * H * 128, device="cuda") * 0.5).to(torch.bfloat16)
        fl = 4.0 * S * S * 128 * H * B * 0.5
        res = []
        for asm in (0, 1):
            set_asm(asm)
            t = timeit(lambda: ops.attn_fwd(qkv, B, S, H, 128, True, 128 ** -0.5))
            res.append("%s %.1f us (%.0f TF/s)" % ("asm" if asm else "c++", 1e3 * t, fl / t / 1e9))
        print("B=%d S=%d H=%d: %s" % (B, S, H, "   ".join(res)), flush=True)
sys.exit(0 if ok else 1)
