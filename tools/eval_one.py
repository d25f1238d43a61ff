#!/usr/bin/env python3
"""BASELINE config 2 under a profiler: 1 warm-up + 3 batch-1 evaluation samples (CLIP on 100 frames, 512-token prompt, 12 forced tokens of
which four are <hand_traj>, KV cache).  Prints ms per sample; `tools/eval_trace.sh` wraps it in rocprofv3 --kernel-trace."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "handsonvlm-release_amd"))
import torch
from hvlm.config import HandsOnVLMConfig
from hvlm.model import HandsOnVLMForCausalLM
from hvlm.synthetic import make_text_batch_np
cfg = HandsOnVLMConfig.vicuna_7b()
m = HandsOnVLMForCausalLM(cfg, device="cuda:0", training=False).init_random(0)
ids, _ = make_text_batch_np(1, 512, 0, n_hand=0); ids = torch.from_numpy(ids)
g = torch.Generator(device="cuda").manual_seed(0)
base = torch.randn(1, 10, 3, 224, 224, device="cuda", generator=g).to(torch.bfloat16)
img = base[:, torch.arange(100, device="cuda") % 10].contiguous()
forced = [11, 12, 32100, 13, 32100, 14, 32100, 15, 32100, 16, 17, 2]
n0 = ids.shape[1]
def force(input_ids, scores):
    s = torch.full_like(scores, -1e9); s[:, forced[input_ids.shape[1] - n0]] = 0.0; return s
run = lambda: m.sample(ids, logits_processor=[force], max_length=n0 + len(forced), eos_token_id=2, image=img)  # noqa: E731
run(); torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(3): run()
torch.cuda.synchronize()
print(f"eval batch 1: {(time.perf_counter() - t0) / 3 * 1e3:.2f} ms per sample", flush=True)
