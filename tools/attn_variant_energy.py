#!/usr/bin/env python3
"""Settled-chip A/B of DEV variants of the two hand-scheduled attention kernels: ~1.5 s of back-to-back launches per variant (the power manager settles),
time per call and joules per call from the socket's energy accumulator (as tools/energy_probe.py).  Needs HVLM_LIB=.../libhvlm_hip_dev.so."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "handsonvlm-release_amd"))
import torch
from hvlm import _lib, ops
import bench
lib = _lib.load()
assert lib.hvlm_is_dev_build(), "development library needed (HVLM_LIB)"
ec = bench.EnergyCounter(0)
FWD = ["production", "prof", "nosm", "nodma", "nolds", "nomfma", "mfmaonly", "kgran", "dotsum", "kgrandot", "lazymax"]
CLIP = ["production", "prof", "nosm", "nodma", "nolds", "nomfma", "dma1", "dma2", "dma5", "dma9", "split", "dotsum", "eagermax"]
st = torch.cuda.current_stream().cuda_stream

def measure(label, fn, seconds=1.5):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); fn(); e1.record(); torch.cuda.synchronize()
    n = max(10, int(seconds * 1e3 / max(1e-3, e0.elapsed_time(e1))))
    torch.cuda.synchronize(); j0 = ec.joules()
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize(); j1 = ec.joules()
    us = e0.elapsed_time(e1) * 1e3 / n
    mj = (j1 - j0) / n * 1e3 if (j0 is not None and j1 is not None) else float("nan")
    print(f"   {label:28s} {us:8.1f} us/call  {mj:8.1f} mJ/call", flush=True)

for rep in range(2):
    for (B, S, H) in ((16, 867, 32), (4, 2048, 40)):
        qkv = (torch.randn(B * S, 3 * H * 128, device="cuda") * 0.5).to(torch.bfloat16)
        out = torch.empty(B * S, H * 128, device="cuda", dtype=torch.bfloat16)
        lse = torch.empty(B, H, S, device="cuda", dtype=torch.float32)
        ws = torch.zeros(1024, device="cuda", dtype=torch.uint8)
        print(f"== decoder forward B={B} S={S} H={H} (pass {rep})")
        if rep == 0:
            for spike in (False, True):
                qq = qkv.clone()
                if spike: qq.view(B, S, 3, H, 128)[:, S // 2::97, 1] *= 6.0           # late large keys: the deferred-maximum slow path
                o2, l2 = torch.empty_like(out), torch.empty_like(lse)
                lib.hvlm_attn_fwd_asm_dev(qq.data_ptr(), out.data_ptr(), lse.data_ptr(), ws.data_ptr(), None, B, S, H, qq.stride(0), 128 ** -0.5, FWD.index("production"), st)
                lib.hvlm_attn_fwd_asm_dev(qq.data_ptr(), o2.data_ptr(), l2.data_ptr(), ws.data_ptr(), None, B, S, H, qq.stride(0), 128 ** -0.5, FWD.index("lazymax"), st)
                torch.cuda.synchronize()
                print(f"   lazymax == production bit for bit (spike={int(spike)}): O {torch.equal(out, o2)}, lse {torch.equal(lse, l2)}", flush=True)
        _lib.configure("attn_fwd_asm", 0)
        measure("c++ kernel", lambda: ops.attn_fwd(qkv, B, S, H, 128, True, 128 ** -0.5))
        _lib.configure("attn_fwd_asm", 1)
        for name in ("production", "lazymax", "kgran", "dotsum"):
            v = FWD.index(name)
            measure("asm " + name, lambda: lib.hvlm_attn_fwd_asm_dev(qkv.data_ptr(), out.data_ptr(), lse.data_ptr(), ws.data_ptr(), None, B, S, H, qkv.stride(0), 128 ** -0.5, v, st))
        del qkv, out, lse
    B, H = 1600, 16
    qkv = torch.randn(B * 257, 3 * H * 64, device="cuda").to(torch.float16)
    out = torch.empty(B * 257, H * 64, device="cuda", dtype=torch.float16)
    print(f"== CLIP attention 1600 x 16 (pass {rep})")
    if rep == 0:
        for spike in (False, True):
            qq = qkv.clone()
            if spike: qq.view(B, 257, 3, H, 64)[:, 70::61, 1] *= 5.0
            o2 = torch.empty_like(out)
            lib.hvlm_attn_clip_asm_dev(qq.data_ptr(), out.data_ptr(), None, B, H, qq.stride(0), 64 ** -0.5, CLIP.index("production"), st)
            lib.hvlm_attn_clip_asm_dev(qq.data_ptr(), o2.data_ptr(), None, B, H, qq.stride(0), 64 ** -0.5, CLIP.index("eagermax"), st)
            torch.cuda.synchronize()
            print(f"   eagermax == production bit for bit (spike={int(spike)}): O {torch.equal(out, o2)}  finite {bool(torch.isfinite(o2.float()).all())}", flush=True)
    _lib.configure("attn_clip_asm", 0)
    measure("c++ resident kernel", lambda: ops.attn_fwd_f16(qkv, B, 257, H, 64, 64 ** -0.5))
    _lib.configure("attn_clip_asm", 1)
    for name in ("production", "eagermax", "dotsum", "split"):
        v = CLIP.index(name)
        measure("asm " + name, lambda: lib.hvlm_attn_clip_asm_dev(qkv.data_ptr(), out.data_ptr(), None, B, H, qkv.stride(0), 64 ** -0.5, v, st))
    del qkv, out
