#!/usr/bin/env python3
"""HBM-bound helper kernels on the bench shapes (GPU box): GB/s = (bytes read + bytes written) / time."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "handsonvlm-release_amd"))
import torch
from hvlm import ops
dev = torch.device("cuda:0"); BF = torch.bfloat16

def timeit(fn, iters=20, warmup=3):
    for _ in range(warmup): fn()
    torch.cuda.synchronize()
    s = torch.cuda.Event(enable_timing=True); e = torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / iters

for (R, C) in [(13872, 4096), (13872, 12288), (13872, 22016), (13872, 11008)]:
    x = torch.randn(R, C, device=dev).to(BF)
    ms = timeit(lambda: ops.transpose(x))
    y = ops.transpose(x)
    ok = torch.equal(y[:, :R], x.t())
    print(f"transpose {R}x{C}: {ms*1e3:.0f} us  {2*R*C*2/ms/1e6:.0f} GB/s  ok={ok}", flush=True)

R, D = 13872, 4096
x = torch.randn(R, D, device=dev).to(BF); w = torch.randn(D, device=dev).to(BF); dy = torch.randn(R, D, device=dev).to(BF)
add = torch.randn(R, D, device=dev).to(BF)
y, rstd = ops.rmsnorm_fwd(x, w, 1e-6)
ms = timeit(lambda: ops.rmsnorm_fwd(x, w, 1e-6))
print(f"rmsnorm_fwd {R}x{D}: {ms*1e3:.0f} us  {2*R*D*2/ms/1e6:.0f} GB/s")
ms = timeit(lambda: ops.rmsnorm_bwd(dy, x, w, rstd, add=add))
print(f"rmsnorm_bwd(+add, +reduce) {R}x{D}: {ms*1e3:.0f} us  {4*R*D*2/ms/1e6:.0f} GB/s")

B, S, H, dh = 400, 257, 16, 64
qkv = (torch.randn(B * S, 3 * H * dh, device=dev) * 0.5).to(BF)
for name, env, ks in (("resident", None, "2"), ("streaming", "1", "1")):
    from hvlm import _lib as _hl
    _hl.configure("attn_no_resident", 1 if env else 0)
    os.environ["HVLM_ATTN_KSTEP"] = ks
    ms = timeit(lambda: ops.attn_fwd(qkv, B, S, H, dh, False, dh ** -0.5))
    print(f"CLIP attention fwd {name} B={B} S={S} H={H}: {ms*1e3:.0f} us  {4*S*S*dh*H*B/ms/1e9:.0f} TF/s (algorithmic)")

M, F = 13872, 11008
gu = torch.randn(M, 2 * F, device=dev).to(BF); dact = torch.randn(M, F, device=dev).to(BF)
rp = (M + 63) // 64 * 64
ms = timeit(lambda: ops.swiglu_fwd(gu))
print(f"swiglu_fwd {M}x{F}: {ms*1e3:.0f} us  {3*M*F*2/ms/1e6:.0f} GB/s")
ms = timeit(lambda: ops.swiglu_fwd_t(gu, rp))
print(f"swiglu_fwd_t {M}x{F}: {ms*1e3:.0f} us  {4*M*F*2/ms/1e6:.0f} GB/s")
ms = timeit(lambda: ops.swiglu_bwd(gu, dact))
print(f"swiglu_bwd {M}x{F}: {ms*1e3:.0f} us  {5*M*F*2/ms/1e6:.0f} GB/s")
ms = timeit(lambda: ops.swiglu_bwd_t(gu, dact, rp))
print(f"swiglu_bwd_t {M}x{F}: {ms*1e3:.0f} us  {7*M*F*2/ms/1e6:.0f} GB/s (algorithmic: 3 reads + 4 writes of M*F)")
