#!/usr/bin/env python3
"""`it` calls of the e4m3 GEMM (row / column scales, bf16 out) on M x N x K: the PMC target of tools/fp8_traffic.sh"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "handsonvlm-release_amd"))
import torch
from hvlm import ops
M, N, K = [int(x) for x in sys.argv[1:4]]
it = int(sys.argv[4]) if len(sys.argv) > 4 else 3
a = torch.randint(0, 120, (M, K), device="cuda", dtype=torch.uint8); b = torch.randint(0, 120, (N, K), device="cuda", dtype=torch.uint8)
rs = torch.rand(M, device="cuda") + 0.5; cs = torch.rand(N, device="cuda") + 0.5
out = torch.empty(M, N, dtype=torch.bfloat16, device="cuda")
for _ in range(it): ops.gemm_fp8_nt_scaled(a, b, rs, cs, out=out)
torch.cuda.synchronize()
