#!/usr/bin/env python3
"""Decode-step micro-benchmark (GPU box): ms per generated token of the 7B decoder stack at cache length ~880, eager native
launch loop (hvlm_llama_decode_step) vs replay of the captured hipGraph (hvlm_llama_decode_step_dev)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "handsonvlm-release_amd"))
import torch
from hvlm.config import HandsOnVLMConfig
from hvlm.llama import LlamaStack

cfg = HandsOnVLMConfig()
st = LlamaStack(cfg, device="cuda:0", training=False)
for lp in st.layers:
    for k in ("qkv", "o", "gu", "down"):
        lp.w[k].normal_(0, 0.02)
    lp.w["ln1"].fill_(1.0); lp.w["ln2"].fill_(1.0)
st.norm_w.fill_(1.0)
cache = st.new_cache(1100)
for c in cache:
    c.normal_(0, 1.0)
x = torch.randn(1, cfg.hidden_size, device="cuda").to(torch.bfloat16)
for mode in ("0", "1", "0", "1"):
    os.environ["HVLM_DECODE_GRAPH"] = mode
    for i in range(3):
        st.decode_step(x, 867 + i, cache)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    n = 48
    for i in range(n):
        st.decode_step(x, 867 + i, cache)
    torch.cuda.synchronize()
    print(f"HVLM_DECODE_GRAPH={mode}: {(time.perf_counter() - t0) / n * 1e3:.3f} ms/token", flush=True)
