#!/usr/bin/env python3
"""What the vendor library does on the same problem (GPU box): torch.matmul (hipBLASLt) on 13872x12288x4096 bf16, TFLOP/s;
run under `rocprofv3 --kernel-trace` to see the kernel name (macro-tile, MFMA shape), or with an argument N to loop N times
for `rocm-smi` sampling."""
import sys, time
import torch
M, N, K = 13872, 12288, 4096
a = torch.rand(M, K, device="cuda").sub_(0.5).to(torch.bfloat16); b = torch.rand(N, K, device="cuda").sub_(0.5).to(torch.bfloat16)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 20
for _ in range(3): c = a @ b.t()
torch.cuda.synchronize()
s = torch.cuda.Event(enable_timing=True); e = torch.cuda.Event(enable_timing=True)
s.record()
for _ in range(n): c = a @ b.t()
e.record(); torch.cuda.synchronize()
print("hipBLASLt %.0f TFLOP/s" % (2.0 * M * N * K * n / s.elapsed_time(e) / 1e9))
