import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "handsonvlm-release_amd"))
import torch
from hvlm import ops
for (N, K) in [(12288, 4096), (4096, 4096), (22016, 4096), (4096, 11008), (32128, 4096)]:
    w = torch.randn(N, K, device="cuda").to(torch.bfloat16); x = torch.randn(K, device="cuda").to(torch.bfloat16)
    y = ops.gemv(w, x); ref = (w.float() @ x.float())
    err = float((y.float() - ref).norm() / ref.norm())
    for _ in range(3): ops.gemv(w, x, out=y)
    torch.cuda.synchronize()
    s = torch.cuda.Event(enable_timing=True); e = torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(20): ops.gemv(w, x, out=y)
    e.record(); torch.cuda.synchronize()
    ms = s.elapsed_time(e) / 20
    print(f"gemv {N}x{K}: {ms*1e3:.1f} us  {N*K*2/ms/1e9:.2f} TB/s  err {err:.2e}", flush=True)
# cold (HBM) rate: cycle through 24 distinct matrices (2.4 GB > 256 MB Infinity Cache)
N, K = 12288, 4096
ws = [torch.randn(N, K, device="cuda").to(torch.bfloat16) for _ in range(24)]
x = torch.randn(K, device="cuda").to(torch.bfloat16); y = torch.empty(N, device="cuda", dtype=torch.bfloat16)
for w in ws: ops.gemv(w, x, out=y)
torch.cuda.synchronize()
s = torch.cuda.Event(enable_timing=True); e = torch.cuda.Event(enable_timing=True)
s.record()
for _ in range(3):
    for w in ws: ops.gemv(w, x, out=y)
e.record(); torch.cuda.synchronize()
ms = s.elapsed_time(e) / 72
print(f"cold gemv {N}x{K}: {ms*1e3:.1f} us  {N*K*2/ms/1e9:.2f} TB/s")
s.record()
for _ in range(3):
    for w in ws: torch.mv(w, x)
e.record(); torch.cuda.synchronize()
ms = s.elapsed_time(e) / 72
print(f"cold torch.mv {N}x{K}: {ms*1e3:.1f} us  {N*K*2/ms/1e9:.2f} TB/s")
