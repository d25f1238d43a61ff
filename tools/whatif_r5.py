#!/usr/bin/env python3
"""What a removed millisecond is worth on the round-5 build (third bullet; the round-3 table is
docs/kernels/energy_whatif.md).  The bench's 7B SFT step is run as it is, then again with ONE kernel family replaced by the cached REAL
result of its calls in the last warm-up step (operand VALUES matter: with uninitialised outputs the following GEMMs draw less power
and the step gets faster for the wrong reason), same process, same box, interleaved A B A B so that drift shows.

    python3 tools/whatif_r5.py [--steps 6] [--what swiglu_bwd attn_bwd attn_fwd clip_attn adamw]

Prints one line per family: kernel family, step ms with / without, difference.  (Kernel time per step of each family: the kernel stats
of the same build, profiles/r05*_bench7b_kernel_stats.txt.)"""
import argparse
import os
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "handsonvlm-release_amd"))

ap = argparse.ArgumentParser()
ap.add_argument("--steps", type=int, default=6)
ap.add_argument("--what", nargs="*", default=["swiglu_bwd", "attn_bwd", "attn_fwd", "clip_attn", "adamw"])
wargs = ap.parse_args()
sys.argv = [sys.argv[0]]

import numpy as np  # noqa: E402
import torch  # noqa: E402
import bench  # noqa: E402
from hvlm import ops  # noqa: E402
from hvlm import dp as hdp  # noqa: E402
from hvlm.model import HandsOnVLMForCausalLM  # noqa: E402
from hvlm.optim import AdamW  # noqa: E402

args = bench.parse()
device = torch.device("cuda", 0)
torch.cuda.set_device(device)
torch.manual_seed(20240)
cfg = bench.make_config(args)
model = HandsOnVLMForCausalLM(cfg, device=device, training=True).init_random(seed=0)
opt = AdamW(model, lr=2e-5, weight_decay=0.0, overlap=True)
buckets = hdp.attach(model, hdp.GradBuckets(), opt)
batch = bench.make_batch(cfg, args.batch, args.text_len, args.unique_frames, seed=100, device=device)


class Cached:
    """Wraps one ops function: mode 'record' runs it and keeps (clones of) what it returned, in call order within a step; mode 'replay'
    returns those without launching anything; mode 'off' is the function itself."""

    def __init__(self, name, clone=True):
        self.name, self.fn, self.clone = name, getattr(ops, name), clone
        self.mode, self.i, self.kept = "off", 0, []
        setattr(ops, name, self)

    def __call__(self, *a, **k):
        if self.mode == "off":
            return self.fn(*a, **k)
        if self.mode == "record":
            r = self.fn(*a, **k)
            if self.clone:
                r = tuple(t.clone() if torch.is_tensor(t) else t for t in r) if isinstance(r, tuple) else (r.clone() if torch.is_tensor(r) else r)
            self.kept.append(r)
            return r
        r = self.kept[self.i]
        self.i += 1
        return r


FAMILIES = {
    "swiglu_bwd": ["swiglu_bwd_t", "swiglu_bwd"],
    "attn_bwd": ["attn_bwd"],
    "attn_fwd": ["attn_fwd"],
    "clip_attn": ["attn_fwd_f16"],
    "adamw": ["adamw_step_"],          # returns nothing: the weights simply stay (same values next step)
}
wraps = {n: Cached(n) for fam in FAMILIES.values() for n in fam}


def step():
    for w in wraps.values():
        w.i = 0
    buckets.begin_step()
    out = model(**batch)
    opt.begin_step(grad_scale=1.0)
    model.backward()
    buckets.wait()
    opt.step()
    return out


def timed(n):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        step()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


for _ in range(3):
    step()
timed(3)                                       # (the allocator still settles after the first recording step: one discarded leg)
print(f"7B B={args.batch} T={args.text_len}: what-if runs, {wargs.steps} steps per leg, A = the step as it is, B = one family replayed from cache", flush=True)
base = []
for fam in wargs.what:
    names = FAMILIES[fam]
    if fam == "attn_fwd":
        # the decoder's forward attention shares ops.attn_fwd with nothing else in the fp16-tower configuration
        pass
    for n in names:
        wraps[n].kept, wraps[n].mode = [], "record"
    step()                                     # the recording step (real kernels + clones)
    for n in names:
        wraps[n].mode = "off"
    a1 = timed(wargs.steps)
    for n in names:
        wraps[n].mode = "replay"
    b1 = timed(wargs.steps)
    for n in names:
        wraps[n].mode = "off"
    a2 = timed(wargs.steps)
    for n in names:
        wraps[n].mode = "replay"
    b2 = timed(wargs.steps)
    for n in names:
        wraps[n].mode, wraps[n].kept = "off", []
    torch.cuda.empty_cache()
    calls = sum(1 for _ in names)
    base += [a1, a2]
    print(f"  {fam:12s} A {a1:7.1f} {a2:7.1f}   B {b1:7.1f} {b2:7.1f}   B - A = {0.5 * (b1 + b2) - 0.5 * (a1 + a2):+6.1f} ms", flush=True)
print(f"  step as it is: mean {np.mean(base):.1f} ms  (min {np.min(base):.1f}, max {np.max(base):.1f})")
