import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "handsonvlm-release_amd"))
import torch
from hvlm import ops
BF = torch.bfloat16
def rel(a, b): return float((a.float() - b.float()).norm() / b.float().norm())
def timeit(fn, iters=8):
    for _ in range(2): fn()
    torch.cuda.synchronize(); s = torch.cuda.Event(enable_timing=True); e = torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters): fn()
    e.record(); torch.cuda.synchronize(); return s.elapsed_time(e) / iters
torch.manual_seed(0)
for (M, N, K) in [(300, 520, 128), (257, 264, 200), (1000, 1000, 520), (512, 768, 1000)]:
    a = torch.randn(M, K, device="cuda").to(BF); b = torch.randn(N, K, device="cuda").to(BF)
    ref = a.float() @ b.float().t()
    for ta, tb in [(False, True), (True, True), (True, False)]:
        if not (ta and tb) and K % 64: continue
        aa = a.t().contiguous() if ta else a; bb = b.t().contiguous() if tb else b
        if (aa.stride(0) % 8) or (bb.stride(0) % 8): continue
        c = ops.gemm(aa, bb, trans_a=ta, trans_b=tb)
        print(f"check {M}x{N}x{K} ta={ta} tb={tb} rel={rel(c, ref):.2e}", flush=True)
for (M, N, K, ta, tb) in [(13872, 4096, 12288, False, True), (13872, 11008, 4096, False, True), (4096, 4096, 13872, True, True),
                          (22016, 4096, 13872, True, True), (12288, 4096, 13872, True, True)]:
    a = (torch.rand(K, M, device="cuda") - 0.5).to(BF) if ta else (torch.rand(M, K, device="cuda") - 0.5).to(BF)
    b = (torch.rand(K, N, device="cuda") - 0.5).to(BF) if tb else (torch.rand(N, K, device="cuda") - 0.5).to(BF)
    out = torch.empty(M, N, dtype=BF, device="cuda")
    ms = timeit(lambda: ops.gemm(a, b, trans_a=ta, trans_b=tb, out=out))
    # baseline: explicit transposes + NT
    def base():
        aa = ops.transpose(a, r_pad=(K + 63) // 64 * 64) if ta else a
        bb = ops.transpose(b, r_pad=(K + 63) // 64 * 64) if tb else b
        ops.gemm_nt(aa, bb, out=out)
    ms0 = timeit(base)
    print(f"bench {M}x{N}x{K} ta={ta} tb={tb}: {2.0*M*N*K/ms/1e9:.0f} TF/s ({ms*1e3:.0f} us) vs transposes+NT {2.0*M*N*K/ms0/1e9:.0f} TF/s ({ms0*1e3:.0f} us)", flush=True)
