"""Data-parallel SFT loop: the build's replacement for `HandsOnVLMTrainer(...).train()` driven by
scripts/finetune.sh — AdamW lr 2e-5, cosine schedule with 3 % warm-up, weight decay 0, per-device batch 16, epochs over the
dataset, checkpoint per epoch (`--save_strategy epoch`) with auto-resume of weights + optimizer + schedule + data order —
on one process per GPU with RCCL gradient all-reduce (hvlm.dp), no
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


class _SkipBatches:
    """Batch sampler that drops the first `skip` batches of the wrapped one WITHOUT loading them (resume mid-epoch)."""

    def __init__(self, inner, skip: int):
        self.inner, self.skip = inner, skip

    def __iter__(self):
        for i, b in enumerate(self.inner):
            if i >= self.skip:
                yield b

    def __len__(self):
        return max(0, len(self.inner) - self.skip)


def latest_checkpoint(output_dir):
    if not output_dir or not os.path.isdir(output_dir):
        return None
    done = sorted(int(d.split("-")[1]) for d in os.listdir(output_dir)
                  if d.startswith("checkpoint-") and os.path.exists(os.path.join(output_dir, d, checkpoint.TRAINER_STATE)))
    return os.path.join(output_dir, f"checkpoint-{done[-1]}") if done else None


def train(model, dataset, output_dir=None, per_device_batch=16, epochs=1, max_steps=None, lr=2e-5, warmup_ratio=0.03,
          weight_decay=0.0, save_every_epoch=True, save_steps=None, log_every=10, seed=0, num_workers=0, stop_after=None):
    """One process per GPU.  Auto-resume (handsonvlm/train/train.py:399-402: `trainer.train(resume_from_checkpoint=True)`
    when output_dir holds a checkpoint-*): weights, the optimizer's fp32 master weights / moments / step count, the CVAE
    head's optimizer, the position in the cosine schedule, the data order (epoch + batches already consumed) and each
    rank's RNG state are restored, so the continued run is bit-identical to an uninterrupted one
    (tests/test_train_resume_cpu.py, tests/test_gpu_next_rows.py).  `stop_after` ends the run early after that many
    global steps (a simulated interruption); `save_steps` adds step-based checkpoints to the per-epoch ones."""
    rank, local, world = dp.init_from_env()
    resume = latest_checkpoint(output_dir)
    if resume is not None:
        model.load_state_dict(checkpoint.load_state(resume))          # BEFORE the optimizer clones its fp32 masters
    opt = AdamW(model, lr=lr, weight_decay=weight_decay, overlap=True)
    buckets = dp.attach(model, dp.GradBuckets(), opt)
    # a seeded sampler also at world size 1: the data order is a function of (seed, epoch), never of global RNG state
    sampler = torch.utils.data.distributed.DistributedSampler(dataset, num_replicas=world, rank=rank, shuffle=True, seed=seed)
    collate = DataCollatorForSupervisedDataset(num_frames=model.config.num_frames)
    batches = torch.utils.data.BatchSampler(sampler, per_device_batch, drop_last=True)
    per_epoch = len(batches)
    assert per_epoch > 0, "dataset smaller than one per-device batch"
    total = max_steps or epochs * per_epoch
    step, t0, log = 0, time.perf_counter(), []
    if resume is not None:
        step = int(checkpoint.load_training_state(opt, resume, rank)["global_step"])
    last_saved = step if resume is not None else -1

    def save():
        nonlocal last_saved
        if not output_dir or last_saved == step:
            return
        path = os.path.join(output_dir, f"checkpoint-{step}")
        if rank == 0:
            checkpoint.save_pretrained(model, path)
        dist = torch.distributed.is_initialized()
        if dist:
            torch.distributed.barrier()                               # the directory exists before the other ranks write
        checkpoint.save_training_state(opt, path, step, step // per_epoch, rank, write_marker=False)
        if dist:
            torch.distributed.barrier()                               # every rank's RNG file / every optimizer shard is on disk ...
        if rank == 0:
            checkpoint.mark_complete(path, step, step // per_epoch)   # ... before the marker latest_checkpoint() trusts appears
        if dist:
            torch.distributed.barrier()
        last_saved = step

    epoch = step // per_epoch
    while step < total and (stop_after is None or step < stop_after):
        sampler.set_epoch(epoch)
        loader = torch.utils.data.DataLoader(dataset, batch_sampler=_SkipBatches(batches, step - epoch * per_epoch),
                                             collate_fn=collate, num_workers=num_workers,
                                             generator=torch.Generator().manual_seed(seed + 7919 * epoch))   # its base seed
                                             # must not be drawn from the global RNG (the CVAE noise's stream)
        for batch in loader:
            if step >= total or (stop_after is not None and step >= stop_after):
                break
            out = model(**batch)
            opt.begin_step(lr=cosine_lr(step, total, lr, warmup_ratio), grad_scale=1.0 / world)   # HF: lr_lambda(steps taken)
            model.backward()
            buckets.wait()
            opt.step()
            step += 1
            if step % log_every == 0 or step == total:
                rec = dict(step=step, loss=float(out.loss.detach()), **{k: float(v) for k, v in out.loss_components.items()},
                           samples_per_s=world * per_device_batch * log_every / (time.perf_counter() - t0))
                t0 = time.perf_counter()
                log.append(rec)
                if rank == 0:
                    print(json.dumps(rec), flush=True)
            if save_steps and step % save_steps == 0:
                save()
        if step % per_epoch == 0 and save_every_epoch:
            save()
        epoch = step // per_epoch
    if step >= total:
        save()
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
    cfg.lm_head_labelled_rows = True      # the trainer reads the loss only: lm_head and the last layer's MLP on labelled rows (exact)
    model = HandsOnVLMForCausalLM(cfg, device=torch.device("cuda", local), training=True).init_random(0)
    train(model, SyntheticEpicDataset(a.samples), a.output_dir, a.per_device_train_batch_size, a.num_train_epochs, a.steps,
          a.learning_rate)


if __name__ == "__main__":
    main()
