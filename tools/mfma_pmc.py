#!/usr/bin/env python3
"""Driver for the MFMA-utilisation counters: launches every MFMA-bound kernel of the hot path a few times at the
shapes the SFT step runs them, so that `rocprofv3 --pmc ... -- python3 tools/mfma_pmc.py` (tools/mfma_pmc.sh) attributes
SQ_VALU_MFMA_BUSY_CYCLES / SQ_BUSY_CYCLES / SQ_INSTS_* to them.  Kernels: gemm_pp4m_kernel (decoder bf16 GEMMs), its fp16 twin
(CLIP tower GEMMs at M = 1600 frames x 257), decoder attention forward (C++ and the hand-scheduled one), backward (dQ, fused dK/dV),
and the CLIP S = 257 resident attention kernel."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "handsonvlm-release_amd"))
import torch
from hvlm import ops, _lib
IT = 3
BF, F16 = torch.bfloat16, torch.float16
dev = "cuda"
def r(*s, dt=BF): return (torch.rand(*s, device=dev) - 0.5).to(dt)
_mark = torch.empty(7, device=dev)
def mark(label):
    """a marker dispatch (FillFunctor) in front of every group: tools/mfma_pmc_table.py cuts the dispatch sequence at it"""
    torch.cuda.synchronize(); LABELS.append(label); _mark.fill_(3.0)
LABELS = []
# decoder GEMMs (7B, M = 16 x 867)
for (M, N, K) in [(13872, 12288, 4096), (13872, 22016, 4096), (13872, 4096, 11008), (13872, 4096, 4096)]:
    a, b = r(M, K), r(N, K); out = torch.empty(M, N, dtype=BF, device=dev)
    mark(f"gemm bf16 {M}x{N}x{K}")
    for _ in range(IT): ops.gemm_nt(a, b, out=out)
    del a, b, out
# CLIP tower GEMMs (fp16, M = 1600 x 257 = 411200)
M = 411200
for (N, K) in [(3072, 1024), (1024, 1024), (4096, 1024), (1024, 4096)]:
    a, b = r(M, K, dt=F16), r(N, K, dt=F16); out = torch.empty(M, N, dtype=F16, device=dev)
    mark(f"gemm f16 {M}x{N}x{K}")
    for _ in range(IT): ops.gemm_f16_nt(a, b, out=out)
    del a, b, out
torch.cuda.synchronize()
# decoder attention, forward (both kernels) and backward
for (B, S, H) in [(16, 867, 32), (4, 2048, 40)]:
    qkv = (torch.randn(B * S, 3 * H * 128, device=dev) * 0.5).to(BF)
    for asm in (0, 1):
        _lib.configure("attn_fwd_asm", asm)
        mark(f"attn fwd {'asm' if asm else 'c++'} B{B} S{S} H{H}")
        for _ in range(IT): o, lse = ops.attn_fwd(qkv, B, S, H, 128, True, 128 ** -0.5)
    _lib.configure("attn_fwd_asm", 0)
    do = torch.randn_like(o)
    mark(f"attn bwd B{B} S{S} H{H}")
    for _ in range(IT): ops.attn_bwd(qkv, o, do, lse, B, S, H, 128, True, 128 ** -0.5)
    del qkv, o, do, lse
# CLIP attention (fp16, S = 257, 1600 frames x 16 heads)
qkv = (torch.randn(1600 * 257, 3 * 1024, device=dev) * 0.5).to(F16)
for asm in (0, 1):
    _lib.configure("attn_clip_asm", asm)
    mark(f"clip attn {'asm' if asm else 'c++'} 1600x16 S257")
    for _ in range(IT): ops.attn_fwd_f16(qkv, 1600, 257, 16, 64, 64 ** -0.5)
_lib.configure("attn_clip_asm", 0)
torch.cuda.synchronize()
import json
json.dump(LABELS, open(os.path.join(ROOT, "gpurun_out", "mfma_pmc_labels.json"), "w"))
