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
VARIANTS = ["production", "prof", "nosm", "nodma", "nolds", "nomfma", "mfmaonly", "noqk", "nopv", "nodmak", "nodmav"]

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
    nwg = B * H * ((S + 255) // 256)
    prof = torch.zeros(nwg * 4 * (8 + 256), device="cuda", dtype=torch.int32)
    st = torch.cuda.current_stream().cuda_stream
    def run(v, p=None):
        rc = lib.hvlm_attn_fwd_asm_dev(qkv.data_ptr(), out.data_ptr(), lse.data_ptr(), p.data_ptr() if p is not None else None, B, S, H,
                                       qkv.stride(0), 128 ** -0.5, v, st)
        assert rc == 0, _lib.last_error()
    fl = 4.0 * S * S * 128 * H * B * 0.5
    print(f"== B={B} S={S} H={H}  ({nwg} workgroups)")
    assert lib.hvlm_configure(b"attn_fwd_asm", 0) == 0
    t = timeit(lambda: ops.attn_fwd(qkv, B, S, H, 128, True, 128 ** -0.5))
    print(f"   c++ kernel            {t:8.1f} us  {fl / t / 1e6:7.0f} TF/s")
    for v, name in enumerate(VARIANTS):
        t = timeit(lambda: run(v))
        print(f"   asm {name:12s}      {t:8.1f} us  {fl / t / 1e6:7.0f} TF/s (algorithmic)")
    prof.zero_(); run(1, prof); torch.cuda.synchronize()
    raw = prof.cpu().numpy().astype("int64") & 0xffffffff
    p = raw[: nwg * 4 * 8].reshape(nwg * 4, 8)
    its = raw[nwg * 4 * 8:].reshape(nwg * 4, 4, 64)          # [wave, stamp kind, iteration]: top, after barrier, after first, after second
    names = ["prologue loads", "prologue compute + step 0", "main loop", "epilogue"]
    for j in sorted(set(p[:, 7].tolist())):
        rows = p[(p[:, 7] == j) & (p[:, 6] > 0)]
        if len(rows) == 0: continue
        nT = rows[:, 6].mean(); steps = 2 * nT - 1
        desc = "  ".join(f"{names[i]} {rows[:, i].mean():8.0f}" for i in range(4))
        act = rows[rows[:, 5] > 0]
        print(f"   j={j} nT={nT:4.1f} waves={len(rows)}: {desc}   loop cycles per step {rows[:, 2].mean() / max(1, steps):7.0f}  total {rows[:, :4].sum(1).mean():8.0f}")
        # iteration anatomy per wave index (rows are ordered (workgroup, wave); the modes differ near the diagonal)
        sel = (p[:, 7] == j) & (p[:, 6] > 0)
        I4 = its[sel]; n_it = int(nT)
        if n_it >= 2:
            d = lambda a, b: ((b - a) & 0xffffffff).astype("float64")
            fmt = lambda v: " ".join(f"{x:5.0f}" for x in v)
            W = np.arange(len(p))[sel] % 4
            for w in range(4):
                m = W == w
                if m.any():
                    print(f"      wave {w}: wait   {fmt(d(I4[m][:, 0, :n_it], I4[m][:, 1, :n_it]).mean(0))}")
                    print(f"              first  {fmt(d(I4[m][:, 1, :n_it], I4[m][:, 2, :n_it]).mean(0))}")
                    print(f"              second {fmt(d(I4[m][:, 2, :n_it], I4[m][:, 3, :n_it]).mean(0))}")
                    print(f"              ctl    {fmt(d(I4[m][:, 3, : n_it - 1], I4[m][:, 0, 1:n_it]).mean(0))}")
