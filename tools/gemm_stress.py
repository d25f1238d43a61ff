#!/usr/bin/env python3
"""Race / edge screen for the GEMM (GPU box): random shapes and epilogues, each run several times, compared with an
fp32 torch matmul on the same device (element-wise: |err| <= 2 bf16 ulp of the reference + fp32 accumulation slack)
and with itself (bit-identical run to run)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "handsonvlm-release_amd"))
import torch
from hvlm import ops

dev = torch.device("cuda:0"); BF = torch.bfloat16
g = torch.Generator(device="cpu").manual_seed(int(sys.argv[1]) if len(sys.argv) > 1 else 0)
n_cases = int(sys.argv[2]) if len(sys.argv) > 2 else 120
bad = 0
for case in range(n_cases):
    M = int(torch.randint(1, 3000, (1,), generator=g)) if case % 3 else int(torch.randint(256, 9000, (1,), generator=g))
    if case % 7 == 0:
        M = int(torch.randint(9000, 30000, (1,), generator=g))      # >= 3 tiles per workgroup: dynamic tile scheduling
    N = int(torch.randint(1, 700, (1,), generator=g)) * 8
    K = int(torch.randint(1, 40, (1,), generator=g)) * 64
    use_bias, use_res, use_act = [bool(torch.randint(0, 2, (1,), generator=g)) for _ in range(3)]
    a = torch.randn(M, K, device=dev).to(BF); b = torch.randn(N, K, device=dev).to(BF)
    bias = torch.randn(N, device=dev).to(BF) if use_bias else None
    res = torch.randn(M, N, device=dev).to(BF) if use_res else None
    ref = a.float() @ b.float().t()
    if use_bias: ref = ref + bias.float()
    if use_act: ref = ref * torch.sigmoid(1.702 * ref)
    if use_res: ref = ref + res.float()
    outs = [ops.gemm_nt(a, b, bias=bias, residual=res, act=1 if use_act else 0) for _ in range(3)]
    same = all(torch.equal(outs[0], o) for o in outs[1:])
    err = (outs[0].float() - ref).abs()
    tol = ref.abs() * 2.0 ** -7 + (K ** 0.5) * 2e-3 + 1e-2
    nbad = int((err > tol).sum()) + int((~torch.isfinite(outs[0].float())).sum())
    if nbad or not same:
        bad += 1
        print(f"FAIL M={M} N={N} K={K} bias={use_bias} res={use_res} act={use_act}: {nbad} elements off, deterministic={same}", flush=True)
print(f"{n_cases - bad}/{n_cases} cases ok")
sys.exit(1 if bad else 0)
