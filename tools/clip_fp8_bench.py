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
