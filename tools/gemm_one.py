#!/usr/bin/env python3
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "handsonvlm-release_amd"))
import torch
from hvlm import ops
M, N, K = [int(x) for x in sys.argv[1:4]]
it = int(sys.argv[4]) if len(sys.argv) > 4 else 3
a = torch.rand(M, K, device="cuda").sub_(0.5).to(torch.bfloat16); b = torch.rand(N, K, device="cuda").sub_(0.5).to(torch.bfloat16)
out = torch.empty(M, N, dtype=torch.bfloat16, device="cuda")
for _ in range(it): ops.gemm_nt(a, b, out=out)
torch.cuda.synchronize()
