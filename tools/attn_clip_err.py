#!/usr/bin/env python3
"""Error anatomy of the hand-scheduled CLIP attention against an fp64 reference, beside the C++ resident kernel's: signed mean, mean |e|, max |e|,
share of outputs that differ from the correctly rounded fp16 value, split by query-row group (regular rows / the class-token row) and by wave."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "handsonvlm-release_amd"))
import torch
from hvlm import ops, _lib
F16 = torch.float16
B, H = 64, 16
g = torch.Generator(device="cuda").manual_seed(0)
for name, scale in (("randn", 1.0), ("randn x 2 (peaked softmax)", 2.0)):
    qkv = (torch.randn(B * 257, 3 * H * 64, device="cuda", generator=g) * scale).to(F16)
    x = qkv.double().view(B, 257, 3, H, 64)
    q, k, v = x[:, :, 0].transpose(1, 2), x[:, :, 1].transpose(1, 2), x[:, :, 2].transpose(1, 2)
    ref = (torch.softmax((q @ k.transpose(-1, -2)) * 0.125, -1) @ v).transpose(1, 2).reshape(B, 257, H, 64)
    best = ref.to(F16)                                    # the correctly rounded answer
    print(f"== {name}: |O| max {ref.abs().max():.2f}, mean {ref.abs().mean():.3f}")
    lib = _lib.load()
    dev_variants = [("dotsum", 11)] if lib.hvlm_is_dev_build() else []          # index in attention_asm.hip kClipVariants
    for asm in [0, 1] + [v for _, v in dev_variants]:
        if asm in (0, 1):
            _lib.configure("attn_clip_asm", asm)
            o = ops.attn_fwd_f16(qkv, B, 257, H, 64, 64 ** -0.5).view(B, 257, H, 64)
        else:
            o = torch.empty(B * 257, H * 64, device="cuda", dtype=F16)
            rc = lib.hvlm_attn_clip_asm_dev(qkv.data_ptr(), o.data_ptr(), None, B, H, qkv.stride(0), 64 ** -0.5, asm, torch.cuda.current_stream().cuda_stream)
            assert rc == 0, _lib.last_error()
            torch.cuda.synchronize()
            o = o.view(B, 257, H, 64)
        e = o.double() - ref
        rel = e / ref.abs().clamp_min(1e-3)
        print(f"  {'c++' if asm == 0 else 'asm' if asm == 1 else 'asm variant %d' % asm}: mean e {e.mean():+.2e}  mean |e| {e.abs().mean():.2e}  max |e| {e.abs().max():.2e}  mean rel e {rel.mean():+.2e}  "
              f"!= correctly rounded: {100.0 * (o != best).double().mean():.1f} %")
        for lab, sl in (("rows 0-127", slice(0, 128)), ("rows 128-255", slice(128, 256)), ("class row 256", slice(256, 257))):
            ee = e[:, sl]
            print(f"       {lab:14s} mean e {ee.mean():+.2e}  mean |e| {ee.abs().mean():.2e}  max |e| {ee.abs().max():.2e}  != rounded {100.0 * (o[:, sl] != best[:, sl]).double().mean():.1f} %")
    _lib.configure("attn_clip_asm", 1)
