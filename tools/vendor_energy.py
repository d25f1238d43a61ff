#!/usr/bin/env python3
"""is the GEMM's "energy ceiling" a ceiling?  The eight production GEMM shapes of the SFT step — four decoder
shapes in bf16, four CLIP-tower shapes in fp16 (M = 411 200 = 1600 frames x 257 tokens) — run back to back for ~1.5 s each (settled
power manager) on this build's persistent kernel and on the vendor library (torch.matmul = hipBLASLt), alternating, with the socket's
energy accumulator read around each loop: us per call, W, mJ per call, TFLOP/s, TFLOP/J.  MEASUREMENT ONLY: the vendor kernel is not
part of the product (libhvlm_hip.so links libamdhip64 alone).

    python3 tools/vendor_energy.py            # the table
    python3 tools/vendor_energy.py --once     # one call of each (run under `rocprofv3 --kernel-trace --stats` for the vendor kernels' names)"""
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "handsonvlm-release_amd"))
import torch  # noqa: E402
from hvlm import ops  # noqa: E402

ONCE = "--once" in sys.argv


def energy_uj():
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


def measure(fn, seconds=1.5):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); fn(); e1.record(); torch.cuda.synchronize()
    n = max(10, int(seconds * 1e3 / max(1e-3, e0.elapsed_time(e1))))
    u0, t0 = energy_uj(), time.perf_counter()
    e0.record()
    for _ in range(n): fn()
    e1.record()
    torch.cuda.synchronize()
    t1, u1 = time.perf_counter(), energy_uj()
    us = e0.elapsed_time(e1) * 1e3 / n
    if u0 is None or u1 is None:
        return us, None
    busy = n * us * 1e-6
    mj = ((u1 - u0) * 1e-6 - IDLE_W * max(0.0, (t1 - t0) - busy)) / n * 1e3
    return us, mj


torch.cuda.init()
_ = torch.zeros(1, device="cuda")
IDLE_W = 0.0
if not ONCE:
    u0, t0 = energy_uj(), time.perf_counter()
    time.sleep(2.0)
    u1, t1 = energy_uj(), time.perf_counter()
    IDLE_W = ((u1 - u0) * 1e-6 / (t1 - t0)) if (u0 is not None and u1 is not None) else 0.0
    print(f"idle (context alive): {IDLE_W:.0f} W", flush=True)

BF, F16 = torch.bfloat16, torch.float16
SHAPES = [("decoder q|k|v", 13872, 12288, 4096, BF), ("decoder o_proj", 13872, 4096, 4096, BF), ("decoder gate|up", 13872, 22016, 4096, BF),
          ("decoder down", 13872, 4096, 11008, BF), ("clip q|k|v", 411200, 3072, 1024, F16), ("clip out_proj", 411200, 1024, 1024, F16),
          ("clip fc1", 411200, 4096, 1024, F16), ("clip fc2", 411200, 1024, 4096, F16)]
g = torch.Generator(device="cuda").manual_seed(0)
rows = []
print(f"{'shape':18s} {'M x N x K':>22s} {'dtype':5s} | {'hvlm us':>9s} {'W':>6s} {'mJ':>9s} {'TF/s':>6s} {'TF/J':>6s} | {'vendor us':>9s} {'W':>6s} {'mJ':>9s} {'TF/s':>6s} {'TF/J':>6s} | "
      f"{'time v/h':>8s} {'mJ v/h':>7s}", flush=True)
for name, M, N, K, dt in SHAPES:
    a = (torch.randn(M, K, device="cuda", generator=g) * 0.5).to(dt)
    b = (torch.randn(N, K, device="cuda", generator=g) * 0.05).to(dt)
    c = torch.empty(M, N, device="cuda", dtype=dt)
    ours = (lambda: ops.gemm_nt(a, b, out=c)) if dt == BF else (lambda: ops.gemm_f16_nt(a, b, out=c))
    vend = lambda: torch.matmul(a, b.t(), out=c)  # noqa: E731
    if ONCE:
        ours(); vend(); torch.cuda.synchronize()
        continue
    fl = 2.0 * M * N * K
    res = {}
    for rep in range(2):                       # h v h v: the second pair is reported (the first settles the chip on this shape)
        for who, fn in (("hvlm", ours), ("vendor", vend)):
            res[who] = measure(fn)
    (hu, hm), (vu, vm) = res["hvlm"], res["vendor"]

    def cols(us, mj):
        if mj is None:
            return f"{us:9.1f} {'-':>6s} {'-':>9s} {fl / us / 1e6:6.0f} {'-':>6s}"
        return f"{us:9.1f} {mj / (us * 1e-3):6.0f} {mj:9.2f} {fl / us / 1e6:6.0f} {fl / (mj * 1e-3) / 1e12:6.3f}"
    print(f"{name:18s} {f'{M} x {N} x {K}':>22s} {'bf16' if dt == BF else 'f16':5s} | {cols(hu, hm)} | {cols(vu, vm)} | {vu / hu:8.3f} "
          f"{(vm / hm) if (hm and vm) else float('nan'):7.3f}", flush=True)
    rows.append(dict(shape=name, M=M, N=N, K=K, dtype="bf16" if dt == BF else "f16", hvlm_us=hu, hvlm_mj=hm, vendor_us=vu, vendor_mj=vm))
    del a, b, c
if rows:
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    with open(os.path.join(ROOT, "gpurun_out", "vendor_energy.json"), "w") as f:
        json.dump(dict(idle_w=IDLE_W, rows=rows), f, indent=1)
