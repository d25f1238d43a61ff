#!/usr/bin/env python3
"""GPU box, development library (make DEV=1; HVLM_LIB=lib/libhvlm_hip_dev.so): where the cycles of the hand-scheduled attention
forward go.  (a) kernel time of the production schedule and of every timing-only ablation (results of the ablations are wrong
by construction); (b) the cycle stamps of the stamped production schedule: per wave {prologue loads, prologue compute + step 0,
main loop, epilogue}, averaged per workgroup class j (the 256-row block index), with cycles per 32-key step."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.environ.setdefault("HVLM_LIB", os.path.join(ROOT, "handsonvlm-release_amd", "lib", "libhvlm_hip_dev.so"))
sys.path.insert(0, os.path.join(ROOT, "handsonvlm-release_amd"))
import torch
import numpy as np
from hvlm import ops, _lib
lib = _lib.load()
VARIANTS = ["production", "prof", "nosm", "nodma", "nolds", "nomfma", "mfmaonly", "kgran", "dotsum", "kgrandot", "lazymax"]
PHASES = ["item start wait", "prologue + step 0", "main loop", "flush + epilogue", "item switch"]

def timeit(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    a = torch.cuda.Event(enable_timing=True); b = torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3

shapes = [(16, 867, 32), (4, 2048, 40)]
for (B, S, H) in shapes:
    qkv = (torch.randn(B * S, 3 * H * 128, device="cuda") * 0.5).to(torch.bfloat16)
    out = torch.empty(B * S, H * 128, device="cuda", dtype=torch.bfloat16)
    lse = torch.empty(B, H, S, device="cuda", dtype=torch.float32)
    ws = torch.zeros(1024, device="cuda", dtype=torch.uint8)
    nwg = min(256, B * H * ((S + 255) // 256))
    prof = torch.zeros(nwg * 4 * 8, device="cuda", dtype=torch.int32)
    st = torch.cuda.current_stream().cuda_stream
    def run(v, p=None):
        rc = lib.hvlm_attn_fwd_asm_dev(qkv.data_ptr(), out.data_ptr(), lse.data_ptr(), ws.data_ptr(), p.data_ptr() if p is not None else None,
                                       B, S, H, qkv.stride(0), 128 ** -0.5, v, st)
        assert rc == 0, _lib.last_error()
    fl = 4.0 * S * S * 128 * H * B * 0.5
    print(f"== B={B} S={S} H={H}  ({nwg} persistent workgroups)")
    assert lib.hvlm_configure(b"attn_fwd_asm", 0) == 0
    t = timeit(lambda: ops.attn_fwd(qkv, B, S, H, 128, True, 128 ** -0.5))
    print(f"   c++ kernel            {t:8.1f} us  {fl / t / 1e6:7.0f} TF/s")
    for v, name in enumerate(VARIANTS):
        t = timeit(lambda: run(v))
        print(f"   asm {name:12s}      {t:8.1f} us  {fl / t / 1e6:7.0f} TF/s (algorithmic)")
    # the row-sum variant (v_dot2 over the rounded pairs) against production and against an fp64 reference on a slice
    hs = min(H, 4)
    xq = qkv.view(B, S, 3, H, 128)[:, :, :, :hs].double()
    q_, k_, v_ = xq[:, :, 0].transpose(1, 2), xq[:, :, 1].transpose(1, 2), xq[:, :, 2].transpose(1, 2)
    sc_ = (q_ @ k_.transpose(-1, -2)) * 128 ** -0.5
    sc_ = sc_.masked_fill(torch.triu(torch.ones(S, S, dtype=torch.bool, device="cuda"), 1), float("-inf"))
    ref = (torch.softmax(sc_, -1) @ v_).transpose(1, 2)                                   # [B, S, hs, 128]
    for name in ("production", "dotsum"):
        run(VARIANTS.index(name)); torch.cuda.synchronize()
        e = out.view(B, S, H, 128)[:, :, :hs].double() - ref
        print(f"   error vs fp64 ({name:10s}): mean e {e.mean():+.2e}  mean |e| {e.abs().mean():.3e}  max |e| {e.abs().max():.2e}")
    del xq, q_, k_, v_, sc_, ref
    prof.zero_(); run(1, prof); torch.cuda.synchronize()
    p = (prof.cpu().numpy().astype("int64") & 0xffffffff).reshape(nwg * 4, 8)
    tot = p[:, :5].sum(1)
    print("   cycle sums per wave (mean over the workgroups), whole kernel:")
    for k, nm in enumerate(PHASES):
        print(f"      {nm:20s} {p[:, k].mean():10.0f}  ({100.0 * p[:, k].mean() / tot.mean():5.1f} %)")
    print(f"      total                {tot.mean():10.0f}   min {tot.min()}  max {tot.max()}")
