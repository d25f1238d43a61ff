#!/usr/bin/env python3
"""Five fp8 evaluation batches of 32 samples (bench.py --fp8-eval's first leg on its own), for tools/eval_batch_prof.sh."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "handsonvlm-release_amd"))
import torch
import bench
from hvlm.model import HandsOnVLMForCausalLM
from hvlm.synthetic import make_text_batch_np
sys.argv = ["bench.py", "--fp8-eval"]
args = bench.parse(); cfg = bench.make_config(args)
device = torch.device("cuda", 0)
B, T = 32, 512
model = HandsOnVLMForCausalLM(cfg, device=device, training=False).init_random(seed=0)
ids, _ = make_text_batch_np(B, T, 7, n_hand=0); ids = torch.from_numpy(ids)
g = torch.Generator(device=device).manual_seed(7)
base = torch.randn(B, 8, 3, 224, 224, device=device, generator=g).to(torch.bfloat16)
img = base[:, torch.arange(cfg.num_frames, device=device) % 8].contiguous()
forced = [11, 12, 32100, 13, 32100, 14, 32100, 15, 32100, 16, 17, 2]
def force(cur, scores):
    s_ = torch.full_like(scores, -1e9); s_[:, forced[cur.shape[1] - T]] = 0.0; return s_
for i in range(5):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    model.sample_batch(ids, logits_processor=[force], max_length=T + len(forced), eos_token_id=2, image=img, fp8=True, fp8_decode=True, fp8_clip=True)
    torch.cuda.synchronize(); print(f"batch {i}: {(time.perf_counter() - t0) * 1e3:.0f} ms", flush=True)
