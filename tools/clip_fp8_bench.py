#!/usr/bin/env python3
"""CLIP ViT-L/14 tower forward on 1600 frames (one evaluation chunk): fp16 (the reference's dtype) against the e4m3 forward."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "handsonvlm-release_amd"))
import torch
from hvlm.clip import CLIPVisionTower
from hvlm.config import CLIPVisionConfig
vc = CLIPVisionConfig()
tw = CLIPVisionTower(vc, "cuda", dtype=torch.float16).init_random(0)
tw.fp8_prepare()
N = int(os.environ.get("FRAMES", "1600"))
img = torch.randn(N, 3, 224, 224, device="cuda").to(torch.bfloat16)
def T(fn, n=3):
    fn(); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
fl = N * tw.flops_per_frame()
t16 = T(lambda: tw.forward_tokens(img)); t8 = T(lambda: tw.forward_tokens(img, fp8=True))
print(f"CLIP tower {N} frames: fp16 {t16:.1f} ms ({fl / t16 / 1e9:.0f} TFLOP/s)   e4m3 {t8:.1f} ms ({fl / t8 / 1e9:.0f} TFLOP/s algorithmic)")

# the four e4m3 GEMMs of a layer on their own (M = frames * 257), with the epilogue each one has in the tower
from hvlm import ops
M = N * 257
d, m = 1024, 4096
g = torch.Generator(device="cuda").manual_seed(1)
def q(r, k): return ops.quantize_fp8_rowwise((torch.randn(r, k, device="cuda", generator=g) * 0.05).to(torch.bfloat16))
x8, ix = q(M, d); h8, ih = q(M, m)
wq, iq = q(3 * d, d); wo, io = q(d, d); w1, i1 = q(m, d); w2, i2 = q(d, m)
bq = torch.zeros(3 * d, device="cuda", dtype=torch.bfloat16); bo = torch.zeros(d, device="cuda", dtype=torch.bfloat16)
b1 = torch.zeros(m, device="cuda", dtype=torch.bfloat16)
res = torch.randn(M, d, device="cuda", generator=g).to(torch.bfloat16)
oinv = torch.full((M,), 0.01, device="cuda")
cases = [("qkv  N=3072 K=1024 bias", lambda: ops.gemm_fp8_nt_scaled(x8, wq, ix, iq, bias=bq), 3 * d, d),
         ("o    N=1024 K=1024 bias+residual", lambda: ops.gemm_fp8_nt_scaled(x8, wo, ix, io, bias=bo, residual=res), d, d),
         ("fc1  N=4096 K=1024 bias+gelu -> bf16", lambda: ops.gemm_fp8_nt_scaled(x8, w1, ix, i1, bias=b1, act=ops.ACT_QUICK_GELU), m, d),
         ("fc1  N=4096 K=1024 bias+gelu -> e4m3", lambda: ops.gemm_fp8_nt_scaled_q8(x8, w1, ix, i1, oinv, bias=b1, act=ops.ACT_QUICK_GELU), m, d),
         ("fc2  N=1024 K=4096 bias+residual", lambda: ops.gemm_fp8_nt_scaled(h8, w2, ih, i2, bias=bo, residual=res), d, m)]
for name, fn, n_, k_ in cases:
    ms = T(fn, 5)
    print(f"  {name:40s} {ms * 1e3:7.0f} us  {2.0 * M * n_ * k_ / ms / 1e9:6.0f} TFLOP/s")
