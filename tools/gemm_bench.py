#!/usr/bin/env python3
"""GEMM micro-benchmark + correctness (GPU box).  The variant list on the command line is kept for the round-1 logs only:
HVLM_GEMM_VARIANT is no longer read by the library (the first-generation kernels were removed in round 2); use
tools/gemm_knobs.py to A/B the switches that exist (HVLM_GEMM_M16, HVLM_GEMM_SEG4, HVLM_GEMM_GROUP_M, HVLM_GEMM_STATIC)."""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "handsonvlm-release_amd"))
import torch
from hvlm import ops
dev = torch.device("cuda:0"); BF = torch.bfloat16

def timeit(fn, iters=10, warmup=3):
    for _ in range(warmup): fn()
    torch.cuda.synchronize()
    s = torch.cuda.Event(enable_timing=True); e = torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / iters

def main():
    variants = sys.argv[1].split(",") if len(sys.argv) > 1 else ["0"]
    shapes = [(13872, 12288, 4096), (13872, 4096, 4096), (13872, 22016, 4096), (13872, 4096, 11008), (4096, 4096, 13888),
              (11008, 4096, 13888), (8192, 8192, 8192), (102800, 4096, 1024), (102800, 1024, 4096), (102800, 3072, 1024)]
    torch.manual_seed(0)
    res = {}
    # correctness
    for v in variants:
        os.environ["HVLM_GEMM_VARIANT"] = v
        errs = []
        for (M, N, K) in [(300, 520, 128), (1000, 1000, 512), (868, 4096, 4096), (257, 64, 64)] if int(v) < 16 else []:
            a = torch.randn(M, K, device=dev).to(BF); b = torch.randn(N, K, device=dev).to(BF)
            bias = torch.randn(N, device=dev).to(BF); r = torch.randn(M, N, device=dev).to(BF)
            ref = (a.float() @ b.float().t() + bias.float()); ref = ref * torch.sigmoid(1.702 * ref) + r.float()
            c = ops.gemm_nt(a, b, bias=bias, residual=r, act=1)
            errs.append(float((c.float() - ref).norm() / ref.norm()))
        res[f"v{v}_err"] = max(errs) if errs else None
    for (M, N, K) in shapes:
        a = torch.rand(M, K, device=dev).sub_(0.5).to(BF); b = torch.rand(N, K, device=dev).sub_(0.5).to(BF)
        out = torch.empty(M, N, dtype=BF, device=dev)
        row = {}
        for rep in range(2):
            for v in variants:
                os.environ["HVLM_GEMM_VARIANT"] = v
                ms = timeit(lambda: ops.gemm_nt(a, b, out=out), iters=8, warmup=2)
                row.setdefault(v, []).append(2.0 * M * N * K / ms / 1e9)
        res[f"{M}x{N}x{K}"] = {v: round(max(t), 1) for v, t in row.items()}
        print(f"{M}x{N}x{K}", res[f"{M}x{N}x{K}"], flush=True)
        del a, b, out
    print(json.dumps(res))

if len(sys.argv) <= 2 or sys.argv[2] == "main":
    main()

def epilogue_bench():
    os.environ.pop("HVLM_GEMM_VARIANT", None)
    for (M, N, K, kind) in [(102800, 1024, 1024, "bias+res"), (102800, 4096, 1024, "bias+gelu"), (102800, 3072, 1024, "bias"),
                            (13872, 4096, 4096, "res"), (13872, 4096, 11008, "res")]:
        a = torch.rand(M, K, device=dev).sub_(0.5).to(BF); b = torch.rand(N, K, device=dev).sub_(0.5).to(BF)
        bias = torch.rand(N, device=dev).to(BF); res = torch.rand(M, N, device=dev).to(BF)
        out = torch.empty(M, N, dtype=BF, device=dev)
        kw = dict(bias=bias if "bias" in kind else None, residual=res if "res" in kind else None, act=1 if "gelu" in kind else 0)
        ms = timeit(lambda: ops.gemm_nt(a, b, out=out, **kw), iters=8, warmup=2)
        print(f"epi {M}x{N}x{K} {kind}: {2.0 * M * N * K / ms / 1e9:.1f} TF/s  {ms * 1e3:.0f} us", flush=True)

if len(sys.argv) > 2 and sys.argv[2] == "epi":
    epilogue_bench()


def group_sweep():
    os.environ.pop("HVLM_GEMM_VARIANT", None)
    for (M, N, K) in [(13872, 12288, 4096), (13872, 4096, 4096), (13872, 4096, 11008), (4096, 4096, 13888), (102800, 4096, 1024), (102800, 1024, 4096)]:
        a = torch.rand(M, K, device=dev).sub_(0.5).to(BF); b = torch.rand(N, K, device=dev).sub_(0.5).to(BF)
        out = torch.empty(M, N, dtype=BF, device=dev)
        row = {}
        for gm in (1, 2, 4, 8, 16, 64):
            from hvlm import _lib as _hl
            _hl.configure("gemm_group_m", gm)
            ms = min(timeit(lambda: ops.gemm_nt(a, b, out=out), iters=8, warmup=2) for _ in range(2))
            row[gm] = round(2.0 * M * N * K / ms / 1e9, 1)
        print(f"group {M}x{N}x{K}", row, flush=True)

if len(sys.argv) > 2 and sys.argv[2] == "group":
    group_sweep()
