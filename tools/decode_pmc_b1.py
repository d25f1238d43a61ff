#!/usr/bin/env python3
"""The batch-1 decode step (7B stack, cache length ~880) for a PMC pass: 1 warm-up + `n` eager steps.  `tools/decode_pmc_b1.sh` wraps it in
`rocprofv3 --pmc FETCH_SIZE` (alone in its pass) and sums the fetched bytes per token: eval.roofline.decode.traffic."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "handsonvlm-release_amd"))
import torch
from hvlm.config import HandsOnVLMConfig
from hvlm.llama import LlamaStack
n = int(sys.argv[1]) if len(sys.argv) > 1 else 2
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
os.environ["HVLM_DECODE_GRAPH"] = "0"
for i in range(1 + n):
    st.decode_step(x, 867 + i, cache)
torch.cuda.synchronize()
