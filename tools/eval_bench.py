#!/usr/bin/env python3
"""Eval-throughput probe (BASELINE configs[1]): HandsOnVLM-7B inference, batch 1, 100x224px frames + 512-token prompt,
forced generation of 12 tokens containing four <hand_traj> (the reference's recipe: no KV cache across steps)."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "handsonvlm-release_amd"))
import torch
from hvlm.config import HandsOnVLMConfig
from hvlm.model import HandsOnVLMForCausalLM
from hvlm.synthetic import make_text_batch_np

cfg = HandsOnVLMConfig.vicuna_7b()
cfg.dedup_frames = "--dedup" in sys.argv
m = HandsOnVLMForCausalLM(cfg, device="cuda:0", training=False).init_random(0)
ids, _ = make_text_batch_np(1, 512, 0, n_hand=0)
ids = torch.from_numpy(ids)
g = torch.Generator(device="cuda").manual_seed(0)
base = torch.randn(1, 10, 3, 224, 224, device="cuda", generator=g).to(torch.bfloat16)
img = base[:, torch.arange(100, device="cuda") % 10].contiguous()
forced = [11, 12, 32100, 13, 32100, 14, 32100, 15, 32100, 16, 17, 2]
n0 = ids.shape[1]
def force(input_ids, scores):
    s = torch.full_like(scores, -1e9); s[:, forced[input_ids.shape[1] - n0]] = 0.0; return s
for kv in (True, False):
    def run():
        return m.sample(ids, logits_processor=[force], max_length=n0 + len(forced), eos_token_id=2, image=img, use_kv_cache=kv)
    out = run(); torch.cuda.synchronize()
    t0 = time.perf_counter(); n = 3
    for _ in range(n): out = run()
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / n
    t0 = time.perf_counter(); v = m.visual_tokens(img); torch.cuda.synchronize(); tv = time.perf_counter() - t0
    print(json.dumps(dict(kv_cache=kv, eval_samples_per_s=1 / dt, ms_per_sample=dt * 1e3, new_tokens=len(forced), clip_ms=tv * 1e3,
                          pred_hands_shape=list(out.pred_hands.shape), dedup=cfg.dedup_frames)))
