#!/usr/bin/env python3
"""BASELINE config 5: HandsOnVLM-7B forward with fp8 (e4m3) QKV/O/FFN GEMMs, batch 32, S = 867 (512-token prompt + 356
visual tokens; 8 distinct frames tiled to the 100 the reference asserts).  Decoder-stack prefill throughput, fp8 vs
bf16, random weights (timing is value independent)."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "handsonvlm-release_amd"))
import torch
from hvlm.config import HandsOnVLMConfig
from hvlm.llama import LlamaStack

B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
S = 867
cfg = HandsOnVLMConfig()
st = LlamaStack(cfg, device="cuda:0", training=False)
g = torch.Generator(device="cuda").manual_seed(0)
for lp in st.layers:
    for k in ("qkv", "o", "gu", "down"):
        lp.w[k].copy_((torch.randn(lp.w[k].shape, device="cuda", generator=g) * 0.02).to(torch.bfloat16))
    lp.w["ln1"].fill_(1.0); lp.w["ln2"].fill_(1.0)
st.norm_w.fill_(1.0)
x = torch.randn(B * S, cfg.hidden_size, device="cuda", generator=g).to(torch.bfloat16)

def timeit(fn, iters=3):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(iters): fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / iters

with torch.no_grad():
    t16 = timeit(lambda: st.forward(x, B, S, save=False))
    ref = st.forward(x, B, S, save=False).float()
    st.fp8_prepare(x, B, S)
    t8 = timeit(lambda: st.forward_fp8(x, B, S))
    got = st.forward_fp8(x, B, S).float()
flops = st.flops_forward(B, S, 0)
print(json.dumps({"config": f"7B decoder stack forward, B={B}, S={S}", "bf16_ms": t16 * 1e3, "fp8_ms": t8 * 1e3,
                  "bf16_samples_per_s": B / t16, "fp8_samples_per_s": B / t8, "bf16_tflops": flops / t16 / 1e12,
                  "fp8_tflops": flops / t8 / 1e12, "fp8_vs_bf16_rel_l2": float((got - ref).norm() / ref.norm())}))
