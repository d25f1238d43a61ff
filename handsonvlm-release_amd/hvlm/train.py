"""Data-parallel SFT loop: the build's replacement for `HandsOnVLMTrainer(...).train()` driven by
scripts/finetune.sh — AdamW lr 2e-5, cosine schedule with 3 % warm-up, weight decay 0, per-device batch 16, epochs over the
dataset, checkpoint per epoch with auto-resume — on one process per GPU with RCCL gradient all-reduce (hvlm.dp), no
DeepSpeed / ZeRO / HF Trainer.

    python -m torch.distributed.run --nproc-per-node 8 -m hvlm.train --model 7b --output_dir ckpt --steps 100
"""
from __future__ import annotations

import argparse
import json
import os
import time

import torch

from . import checkpoint, dp
from .config import CLIPVisionConfig, HandsOnVLMConfig
from .data import DataCollatorForSupervisedDataset, SyntheticEpicDataset
from .optim import AdamW, cosine_lr


def train(model, dataset, output_dir=None, per_device_batch=16, epochs=1, max_steps=None, lr=2e-5, warmup_ratio=0.03,
          weight_decay=0.0, save_every_epoch=True, log_every=10, seed=0, num_workers=0):
    rank, local, world = dp.init_from_env()
    opt = AdamW(model, lr=lr, weight_decay=weight_decay, overlap=True)
    buckets = dp.attach(model, dp.GradBuckets(), opt)
    sampler = torch.utils.data.distributed.DistributedSampler(dataset, num_replicas=world, rank=rank, shuffle=True, seed=seed) \
        if world > 1 else None
    loader = torch.utils.data.DataLoader(dataset, batch_size=per_device_batch, sampler=sampler, shuffle=sampler is None,
                                         collate_fn=DataCollatorForSupervisedDataset(num_frames=model.config.num_frames),
                                         num_workers=num_workers, drop_last=True)
    total = max_steps or epochs * len(loader)
    step, t0, log = 0, time.perf_counter(), []
    start_epoch = 0
    if output_dir and os.path.isdir(output_dir):                      # auto-resume (handsonvlm/train/train.py:399-402)
        done = sorted(int(d.split("-")[1]) for d in os.listdir(output_dir) if d.startswith("checkpoint-"))
        if done:
            model.load_state_dict(checkpoint.load_state(os.path.join(output_dir, f"checkpoint-{done[-1]}")))
            step = done[-1]
            start_epoch = step // max(1, len(loader))
    for epoch in range(start_epoch, epochs if max_steps is None else 10 ** 9):
        if sampler is not None:
            sampler.set_epoch(epoch)
        for batch in loader:
            if step >= total:
                break
            out = model(**batch)
            step += 1
            opt.begin_step(lr=cosine_lr(step, total, lr, warmup_ratio), grad_scale=1.0 / world)
            model.backward()
            buckets.wait()
            opt.step()
            if step % log_every == 0 or step == total:
                rec = dict(step=step, loss=float(out.loss.detach()), **{k: float(v) for k, v in out.loss_components.items()},
                           samples_per_s=world * per_device_batch * log_every / (time.perf_counter() - t0))
                t0 = time.perf_counter()
                log.append(rec)
                if rank == 0:
                    print(json.dumps(rec), flush=True)
        if output_dir and save_every_epoch and rank == 0:
            checkpoint.save_pretrained(model, os.path.join(output_dir, f"checkpoint-{step}"))
        if step >= total:
            break
    return log


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--model", default="7b", choices=["7b", "13b", "tiny"])
    ap.add_argument("--output_dir", default=None)
    ap.add_argument("--per_device_train_batch_size", type=int, default=16)
    ap.add_argument("--num_train_epochs", type=int, default=1)
    ap.add_argument("--steps", type=int, default=None)
    ap.add_argument("--learning_rate", type=float, default=2e-5)
    ap.add_argument("--samples", type=int, default=48296, help="samples per epoch (scripts/finetune.sh:19)")
    a = ap.parse_args()
    rank, local, world = dp.init_from_env()
    from .model import HandsOnVLMForCausalLM
    if a.model == "tiny":
        cfg = HandsOnVLMConfig(hidden_size=256, intermediate_size=512, num_hidden_layers=2, num_attention_heads=2, mm_hidden_size=64,
                               vision=CLIPVisionConfig(hidden_size=64, intermediate_size=128, num_hidden_layers=3, num_attention_heads=1))
    else:
        cfg = HandsOnVLMConfig.vicuna_7b() if a.model == "7b" else HandsOnVLMConfig.vicuna_13b()
    cfg.dedup_frames = True
    model = HandsOnVLMForCausalLM(cfg, device=torch.device("cuda", local), training=True).init_random(0)
    train(model, SyntheticEpicDataset(a.samples), a.output_dir, a.per_device_train_batch_size, a.num_train_epochs, a.steps,
          a.learning_rate)


if __name__ == "__main__":
    main()
