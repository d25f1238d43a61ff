"""Data-parallel SFT loop: the build's replacement for `HandsOnVLMTrainer(...).train()` driven by
scripts/finetune.sh — AdamW lr 2e-5, cosine schedule with 3 % warm-up, weight decay 0, per-device batch 16, epochs over the
dataset, checkpoint per epoch (`--save_strategy epoch`) with auto-resume of weights + optimizer + schedule + data order —
on one process per GPU with RCCL gradient all-reduce (hvlm.dp), no
DeepSpeed / ZeRO / HF Trainer.

    python -m torch.distributed.run --nproc-per-node 8 -m hvlm.train --model 7b --output_dir ckpt --steps 100
    python -m torch.distributed.run --nproc-per-node 8 -m hvlm.train --model_name_or_path ./checkpoints/vicuna-v1-3-7b \
        --pretrain_mm_mlp_adapter ./checkpoints/llava-vicuna-v1-3-7b-pretrain/mm_projector.bin --vision_tower ./checkpoints/clip-vit-large-patch14 \
        --bf16 True --output_dir ./checkpoints/HandsOnVLM-CVAE-7B --num_train_epochs 40 --per_device_train_batch_size 16 --learning_rate 2e-5 \
        --weight_decay 0. --warmup_ratio 0.03 --lr_scheduler_type cosine --model_max_length 2048 ...      (scripts/finetune.sh:6-60)
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
from .prefetch import DevicePrefetcher


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
          weight_decay=0.0, save_every_epoch=True, save_steps=None, log_every=10, seed=0, num_workers=0, stop_after=None,
          model_max_length=2048):
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
    if buckets.active:
        dp.check_topology()                  # rank 0 prints the device map; two ranks on one GPU abort here, not after an epoch
    # a seeded sampler also at world size 1: the data order is a function of (seed, epoch), never of global RNG state
    sampler = torch.utils.data.distributed.DistributedSampler(dataset, num_replicas=world, rank=rank, shuffle=True, seed=seed)
    # --model_max_length is the TOKENIZER's truncation length in the reference (handsonvlm/train/train.py:243-249,
    # hybrid_dataset.py:127-129): it bounds the collated ids, never the checkpoint's max_position_embeddings / RoPE table
    collate = DataCollatorForSupervisedDataset(num_frames=model.config.num_frames, model_max_length=model_max_length,
                                               image_size=model.config.vision.image_size)
    batches = torch.utils.data.BatchSampler(sampler, per_device_batch, drop_last=True)
    per_epoch = len(batches)
    assert per_epoch > 0, "dataset smaller than one per-device batch"
    total = max_steps or epochs * per_epoch
    step, t0, log = 0, time.perf_counter(), []
    if resume is not None:
        step = int(checkpoint.load_training_state(opt, resume, rank)["global_step"])
    logged_at = step                       # (samples_per_s of a log record = the steps since the previous record)
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
        # the host -> device edge: batch i + 1 is staged in pinned memory and uploaded on a copy stream while step i computes;
        # tiled EK100 clips cross the bus as their 10 distinct frames + an index (hvlm/prefetch.py).  CPU runs pass through.
        feed = DevicePrefetcher(loader, model.device, dedup_period=10 if model.config.dedup_frames else None)
        for batch in feed:
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
                           samples_per_s=world * per_device_batch * (step - logged_at) / (time.perf_counter() - t0))
                t0, logged_at = time.perf_counter(), step
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


# scripts/finetune.sh flags that steer subsystems outside the hot path (DeepSpeed, the HF Trainer's logging / eval plumbing, the
# multi-task dataset mixer): accepted so that the script's command line runs unchanged, listed once, otherwise ignored
_IGNORED_NOTE = "accepted for command-line compatibility with scripts/finetune.sh and ignored (outside the hot path)"


def build_parser():
    ap = argparse.ArgumentParser(description="HandsOnVLM SFT on MI355X (the hot path of scripts/finetune.sh)", allow_abbrev=False)
    b = lambda v: str(v).lower() in ("1", "true", "yes")  # noqa: E731   (HfArgumentParser-style booleans: `--bf16 True`)
    ap.add_argument("--model", default="7b", choices=["7b", "13b", "tiny"], help="random-init geometry when no --model_name_or_path is given")
    ap.add_argument("--model_name_or_path", default=None, help="HandsOnVLM checkpoint or base Vicuna directory (scripts/finetune.sh:8)")
    ap.add_argument("--pretrain_mm_mlp_adapter", default=None, help="LLaVA mm_projector.bin (scripts/finetune.sh:12)")
    ap.add_argument("--vision_tower", default=None, help="LOCAL CLIPVisionModel directory (the reference passes a hub name, :11)")
    ap.add_argument("--output_dir", default=None)
    ap.add_argument("--per_device_train_batch_size", type=int, default=16)
    ap.add_argument("--num_train_epochs", type=int, default=1)
    ap.add_argument("--steps", "--max_steps", dest="steps", type=int, default=None)
    ap.add_argument("--learning_rate", type=float, default=2e-5)
    ap.add_argument("--weight_decay", type=float, default=0.0)
    ap.add_argument("--warmup_ratio", type=float, default=0.03)
    ap.add_argument("--lr_scheduler_type", default="cosine")
    ap.add_argument("--bf16", type=b, default=True)
    ap.add_argument("--model_max_length", type=int, default=2048)
    ap.add_argument("--gradient_checkpointing", type=b, default=False,
                    help="activation recompute (config.activation_recompute).  The reference needs it on 80-GB parts (:36); 288 GB do "
                         "not at 7B / per-device 16, so the default here is off — pass True to mirror the script")
    ap.add_argument("--lambda_traj", type=float, default=None)
    ap.add_argument("--lambda_traj_kl", type=float, default=None)
    ap.add_argument("--hoi_lambda", type=float, default=None)
    # architecture flags: None = "not given" -> a loaded checkpoint's config.json decides (random-init models take the
    # reference's script values: CVAE, origin, temporal_spatial_pool, -2)
    ap.add_argument("--traj_decoder", default=None)
    ap.add_argument("--fuse_input_mode", default=None)
    ap.add_argument("--video_arch", default=None)
    ap.add_argument("--mm_vision_select_layer", type=int, default=None)
    ap.add_argument("--samples_per_epoch", "--samples", dest="samples", type=int, default=48296, help="scripts/finetune.sh:56")
    ap.add_argument("--save_strategy", default="epoch", help='"epoch" (scripts/finetune.sh:27) | "steps" (then --save_steps applies, as in HF)')
    ap.add_argument("--save_steps", type=int, default=None)
    ap.add_argument("--dataloader_num_workers", type=int, default=0)
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--labelled_rows", type=b, default=False,
                    help="lm_head / CE / the last layer's MLP on labelled rows only (exact; `out.logits` is then unavailable)")
    ap.add_argument("--dedup_frames", type=b, default=True, help="CLIP once per distinct frame (EK100 tiles 10 frames x10; exact)")
    ap.add_argument("--device", default="cuda", help="cuda (default) | cpu (host-logic tests with the ops stand-in)")
    ap.add_argument("--stop_after", type=int, default=None, help=argparse.SUPPRESS)
    return ap


def build_model(a, device):
    from .model import HandsOnVLMForCausalLM
    # only flags that were GIVEN override the configuration (a checkpoint's config.json carries its own traj_decoder_name,
    # fuse_input_mode, ...); --model_max_length is not a model property at all (see train(): it truncates the collated ids)
    over = {}
    for flag, field in (("lambda_traj", "lambda_traj"), ("lambda_traj_kl", "lambda_traj_kl"), ("hoi_lambda", "hoi_lambda"),
                        ("traj_decoder", "traj_decoder_name"), ("fuse_input_mode", "fuse_input_mode"),
                        ("video_arch", "video_compress_mode"), ("mm_vision_select_layer", "mm_vision_select_layer")):
        if getattr(a, flag) is not None:
            over[field] = getattr(a, flag)
    over.update(dedup_frames=a.dedup_frames, lm_head_labelled_rows=a.labelled_rows, activation_recompute=a.gradient_checkpointing)
    if a.model_name_or_path:
        return checkpoint.from_pretrained(a.model_name_or_path, device=device, training=True, mm_projector_bin=a.pretrain_mm_mlp_adapter,
                                          vision_tower_path=a.vision_tower, **over)
    if a.model == "tiny":
        cfg = HandsOnVLMConfig(hidden_size=256, intermediate_size=512, num_hidden_layers=2, num_attention_heads=2, mm_hidden_size=64,
                               vision=CLIPVisionConfig(hidden_size=64, intermediate_size=128, num_hidden_layers=3, num_attention_heads=1))
    else:
        cfg = HandsOnVLMConfig.vicuna_7b() if a.model == "7b" else HandsOnVLMConfig.vicuna_13b()
    for k, v in over.items():
        setattr(cfg, k, v)
    return HandsOnVLMForCausalLM(cfg, device=device, training=True).init_random(0)


def main(argv=None, dataset=None):
    """`python -m hvlm.train <the flags of scripts/finetune.sh>`: loads `--model_name_or_path` (+ `--pretrain_mm_mlp_adapter`,
    `--vision_tower`) through hvlm.checkpoint like handsonvlm/train/train.py:235-240, or a random-init model of `--model`
    geometry when no path is given, and runs the DP loop above.  Dataset readers are out of scope: `dataset`
    (any torch Dataset yielding the collator's per-sample dicts, handsonvlm/dataset/hybrid_dataset.py:108-169) is passed by the
    caller; the command line alone trains on SyntheticEpicDataset."""
    a, unknown = build_parser().parse_known_args(argv)
    rank, local, world = dp.init_from_env()
    if rank == 0 and unknown:
        names = sorted({u for u in unknown if u.startswith("--")})
        print(f"hvlm.train: {len(names)} flags {_IGNORED_NOTE}: {' '.join(names)}", flush=True)
    assert a.bf16, "the hot path computes in bf16 (scripts/finetune.sh:15); fp32 / fp16 training is not built"
    assert a.lr_scheduler_type == "cosine", "only the cosine schedule of scripts/finetune.sh:33 is built"
    device = torch.device("cuda", local) if a.device == "cuda" else torch.device(a.device)
    model = build_model(a, device)
    ds = dataset if dataset is not None else SyntheticEpicDataset(a.samples, seed=a.seed, image_size=model.config.vision.image_size)
    return train(model, ds, a.output_dir, a.per_device_train_batch_size, a.num_train_epochs, a.steps, a.learning_rate,
                 warmup_ratio=a.warmup_ratio, weight_decay=a.weight_decay, save_steps=a.save_steps if a.save_strategy == "steps" else None,
                 save_every_epoch=a.save_strategy != "no", seed=a.seed,
                 num_workers=a.dataloader_num_workers, stop_after=a.stop_after, log_every=1 if (a.steps or 0) and a.steps <= 10 else 10,
                 model_max_length=a.model_max_length)


if __name__ == "__main__":
    main()
