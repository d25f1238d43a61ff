#!/usr/bin/env python3
"""Where do the torch-side device copies / fills of one SFT step come from?  (rocprofv3: ~200 __amd_rocclr_copyBuffer launches and
~90 float fills per step, ~7 ms of kernel time.)  One bench-shaped step under torch.profiler with stacks; prints the copy_ / fill_ /
zero_ / clone / contiguous calls grouped by the hvlm source line that issued them, with their device time."""
import os, sys, collections
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "handsonvlm-release_amd"))
import torch
import bench
from torch.profiler import profile, ProfilerActivity

args = bench.parse()
cfg = bench.make_config(args)
from hvlm.model import HandsOnVLMForCausalLM
from hvlm.optim import AdamW
from hvlm import dp as hdp
dev = torch.device("cuda", 0)
model = HandsOnVLMForCausalLM(cfg, device=dev, training=True).init_random(seed=0)
opt = AdamW(model, lr=2e-5, overlap=True)
buckets = hdp.attach(model, hdp.GradBuckets(), opt)
batch = bench.make_batch(cfg, args.batch, args.text_len, 10, seed=100, device=dev)


def step():
    buckets.begin_step(); out = model(**batch); opt.begin_step(grad_scale=1.0); model.backward(); buckets.wait(); opt.step(); return out


for _ in range(2):
    step()
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True, record_shapes=True) as prof:
    step()
    torch.cuda.synchronize()
agg = collections.defaultdict(lambda: [0, 0.0, ""])
for ev in prof.events():
    if ev.name in ("aten::copy_", "aten::fill_", "aten::zero_", "aten::clone", "aten::contiguous", "aten::index_select", "aten::to", "aten::_to_copy", "aten::cat", "aten::zeros", "aten::index_copy_"):
        dt = getattr(ev, "device_time_total", 0) or getattr(ev, "cuda_time_total", 0)
        if dt <= 0:
            continue
        where = next((f for f in (ev.stack or []) if "hvlm" in f or "bench.py" in f or "copy_audit" in f), (ev.stack or ["?"])[0] if ev.stack else "?")
        k = (ev.name, where.strip()[:110], str(ev.input_shapes)[:60])
        agg[k][0] += 1; agg[k][1] += dt
tot = sum(v[1] for v in agg.values())
print(f"total device time of torch copy/fill ops in one step: {tot/1e3:.2f} ms")
for k, v in sorted(agg.items(), key=lambda kv: -kv[1][1])[:40]:
    print(f"{v[1]/1e3:8.3f} ms  x{v[0]:4d}  {k[0]:18s} {k[2]:60s} {k[1]}")
