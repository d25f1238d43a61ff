#!/usr/bin/env python3
"""Joules per call: the socket's energy accumulator (rocm-smi --showenergycounter) read around ~1.5 s of back-to-back launches of ONE
kernel, for the kernels whose cycle counts and in-situ times disagree (profiles/r05i: the hand-scheduled attention kernels need 15-20 %
fewer cycles than the C++ ones and take the same time inside the step).  If the step is ENERGY-limited — time = energy / (cap - base
power) — then joules per call, not cycles per call, predict what a kernel costs in situ.

    python3 tools/energy_probe.py            # prints one line per kernel: us / call back to back, W, mJ / call, mJ above idle / call"""
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "handsonvlm-release_amd"))
import torch  # noqa: E402
from hvlm import _lib, ops  # noqa: E402


def energy_uj():
    """accumulated socket energy in microjoules (None if the tool does not answer)"""
    try:
        out = subprocess.run(["rocm-smi", "--showenergycounter", "--json"], capture_output=True, text=True, timeout=20).stdout
        d = json.loads(out[out.index("{"):])
        card = d[sorted(d)[0]]
        for k, v in card.items():
            if "Accumulated Energy" in k:
                return float(v)
    except Exception as ex:  # noqa
        sys.stderr.write(f"energy counter: {ex}\n")
    return None


def power_w():
    try:
        out = subprocess.run(["rocm-smi", "--showpower", "--json"], capture_output=True, text=True, timeout=20).stdout
        d = json.loads(out[out.index("{"):])
        card = d[sorted(d)[0]]
        for k, v in card.items():
            if "Power" in k:
                return float(v)
    except Exception:  # noqa
        pass
    return None


def measure(name, fn, seconds=1.5, flop=None):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); fn(); e1.record(); torch.cuda.synchronize()
    n = max(10, int(seconds * 1e3 / max(1e-3, e0.elapsed_time(e1))))
    torch.cuda.synchronize()
    u0, t0 = energy_uj(), time.perf_counter()
    e0.record()
    for _ in range(n): fn()
    e1.record()
    torch.cuda.synchronize()
    t1, u1 = time.perf_counter(), energy_uj()
    us = e0.elapsed_time(e1) * 1e3 / n
    if u0 is None or u1 is None:
        print(f"{name:44s} {us:9.1f} us/call   (no energy counter)", flush=True)
        return
    wall = t1 - t0
    joule = (u1 - u0) * 1e-6
    # the reads bracket `wall` seconds of which n * us are kernel time: the rest is the idle floor
    busy = n * us * 1e-6
    mj = (joule - IDLE_W * max(0.0, wall - busy)) / n * 1e3
    line = f"{name:44s} {us:9.1f} us/call  {mj / (us * 1e-3):7.0f} W  {mj:9.2f} mJ/call  {mj - IDLE_W * us * 1e-3:9.2f} mJ above idle"
    if flop: line += f"  {flop / us / 1e6:7.0f} TF/s  {flop / (mj * 1e-3) / 1e12:6.2f} TFLOP/J"
    print(line, flush=True)


torch.cuda.init()
_ = torch.zeros(1, device="cuda")
u0, t0 = energy_uj(), time.perf_counter()
time.sleep(2.0)
u1, t1 = energy_uj(), time.perf_counter()
IDLE_W = ((u1 - u0) * 1e-6 / (t1 - t0)) if (u0 is not None and u1 is not None) else 0.0
print(f"idle (context alive, nothing running): {IDLE_W:.0f} W   (rocm-smi power reading: {power_w()})", flush=True)

BF, F16 = torch.bfloat16, torch.float16
g = torch.Generator(device="cuda").manual_seed(0)
# decoder attention forward, the step's shape and config 4's
for B, S, H in ((16, 867, 32), (4, 2048, 40)):
    qkv = (torch.randn(B * S, 3 * H * 128, device="cuda", generator=g) * 0.5).to(BF)
    fl = 4.0 * B * H * 128 * S * (S + 1) / 2
    for asm in (0, 1):
        _lib.configure("attn_fwd_asm", asm)
        measure(f"attn fwd {'asm' if asm else 'c++'} B{B} S{S} H{H}", lambda: ops.attn_fwd(qkv, B, S, H, 128, True, 128 ** -0.5, need_lse=True), flop=fl)
    if S == 867:
        o, lse = ops.attn_fwd(qkv, B, S, H, 128, True, 128 ** -0.5, need_lse=True)
        do = torch.randn_like(o)
        measure(f"attn bwd (dq + dkv) B{B} S{S} H{H}", lambda: ops.attn_bwd(qkv, o, do, lse, B, S, H, 128, True, 128 ** -0.5), flop=2.5 * fl)
        del o, lse, do
    del qkv
_lib.configure("attn_fwd_asm", 1)
# CLIP attention at the tower's shape
B, H = 1600, 16
qkv = torch.randn(B * 257, 3 * H * 64, device="cuda", generator=g).to(F16)
fl = 4.0 * 257 * 257 * 64 * H * B
for asm in (0, 1):
    _lib.configure("attn_clip_asm", asm)
    measure(f"clip attn {'asm' if asm else 'c++'} 1600x16 S257", lambda: ops.attn_fwd_f16(qkv, B, 257, H, 64, 64 ** -0.5), flop=fl)
_lib.configure("attn_clip_asm", 1)
del qkv
# the GEMMs around them
for M, N, K, dt in ((13872, 12288, 4096, BF), (13872, 22016, 4096, BF), (13872, 4096, 11008, BF), (411200, 3072, 1024, F16), (411200, 1024, 1024, F16)):
    a = (torch.randn(M, K, device="cuda", generator=g) * 0.5).to(dt); b = (torch.randn(N, K, device="cuda", generator=g) * 0.05).to(dt)
    c = torch.empty(M, N, device="cuda", dtype=dt)
    fn = (lambda: ops.gemm_nt(a, b, out=c)) if dt == BF else (lambda: ops.gemm_f16_nt(a, b, out=c))
    measure(f"gemm {'bf16' if dt == BF else 'f16'} {M}x{N}x{K}", fn, flop=2.0 * M * N * K)
    del a, b, c
# HBM-bound passes
x = torch.randn(13872, 22016, device="cuda").to(BF); d = torch.randn(13872, 11008, device="cuda").to(BF)
measure("swiglu_bwd_t 13872x22016", lambda: ops.swiglu_bwd_t(x, d, 13888))
measure("swiglu_fwd 13872x22016", lambda: ops.swiglu_fwd(x))
