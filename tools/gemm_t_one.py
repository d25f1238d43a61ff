import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "handsonvlm-release_amd"))
import torch
from hvlm import ops
M, N, K = 4096, 4096, 13872
a = (torch.rand(K, M, device="cuda") - 0.5).to(torch.bfloat16); b = (torch.rand(K, N, device="cuda") - 0.5).to(torch.bfloat16)
out = torch.empty(M, N, dtype=torch.bfloat16, device="cuda")
for _ in range(2): ops.gemm(a, b, trans_a=True, trans_b=True, out=out)
torch.cuda.synchronize()
