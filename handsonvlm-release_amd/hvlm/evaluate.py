"""Evaluation harness: counterpart of `HandsOnVLMInference.evaluate_epic_kitchen_traj`
(handsonvlm/evaluation/handsonvlm_inference.py:127-174) with its two defects fixed — `inference()` returning three values
into a two-name unpack (:116,124 vs :159) and the metric recomputed over all samples after every sample (O(N^2), :173) —
and the reference's decoding recipe kept: batch 1, sampling T=0.5 / top-p 0.9, <= 100 new tokens (:83-85,99-109).
Shards samples across ranks (one process per GPU) and gathers predictions for the metric."""
from __future__ import annotations

from typing import Dict, Iterable, Optional

import numpy as np
import torch
import torch.distributed as dist

from .traj_eval import evaluate_traj_stochastic


@torch.no_grad()
def inference(model, sample: Dict, temperature=0.5, top_p=0.9, max_new_tokens=100, do_sample=True, hand_noise=None):
    """One sample -> (sequences, pred_hands [1,2,n,2] or None).  `sample`: input_ids [1,T] (prompt), image [1,100,3,H,W]."""
    out = model.generate(sample["input_ids"], image=sample["image"], do_sample=do_sample, temperature=temperature, top_p=top_p,
                         num_beams=1, max_new_tokens=max_new_tokens, use_cache=False, output_hidden_states=True,
                         return_dict_in_generate=True, hand_noise=hand_noise)
    ph = out.pred_hands
    return out.sequences, (ph if ph.shape[2] > 0 else None)


@torch.no_grad()
def inference_batch(model, batch, temperature=0.5, top_p=0.9, max_new_tokens=100, do_sample=True, fp8=False, pad_token_id=0,
                    fp8_decode=False, fp8_clip=False):
    """A list of samples -> list of pred_hands ([1,2,n,2] or None), generated in lock step (model.generate_batch): prompts
    are right-padded to the longest one, exactly like the reference's collator would.  fp8 / fp8_decode / fp8_clip: e4m3 prefill
    GEMMs / e4m3-weight decode steps / e4m3 CLIP tower (BASELINE config 5; all off = the reference's bf16 + fp16-tower numerics)."""
    T = max(s["input_ids"].shape[1] for s in batch)
    ids = torch.full((len(batch), T), pad_token_id, dtype=torch.long)
    mask = torch.zeros((len(batch), T), dtype=torch.bool)
    for b, s in enumerate(batch):
        n = s["input_ids"].shape[1]
        ids[b, :n] = s["input_ids"][0]
        mask[b, :n] = True
    image = torch.cat([s["image"] for s in batch], 0)
    out = model.generate_batch(ids, image=image, attention_mask=mask, do_sample=do_sample, temperature=temperature, top_p=top_p,
                               max_new_tokens=max_new_tokens, fp8=fp8, fp8_decode=fp8_decode, fp8_clip=fp8_clip)
    return [out.pred_hands[b:b + 1, :, : out.num_hands[b]] if out.num_hands[b] > 0 else None for b in range(len(batch))]


@torch.no_grad()
def evaluate_traj(model, samples: Iterable[Dict], reference_labels: bool = True, batch_size: int = 1, **gen_kw):
    """samples yield dicts with input_ids, image, future_hands [1,2,5,2] (GT; first point dropped), future_valid [1,2].
    Samples whose generation does not contain exactly 4 <hand_traj> tokens are scored with zeros and valid=0, the
    reference's handling of failed generations.  Returns {'ade','fde','wde','n'} with the reference's label convention.
    batch_size > 1 generates that many samples in lock step (model.sample_batch) instead of the reference's one by one."""
    rank = dist.get_rank() if dist.is_initialized() else 0
    world = dist.get_world_size() if dist.is_initialized() else 1
    preds, gts, valids = [], [], []

    def score(s, ph):
        gt = s["future_hands"][:, :, 1:, :].float().cpu().numpy()[0]
        fv = s["future_valid"].float().cpu().numpy()[0]
        if ph is None or ph.shape[2] != 4:
            preds.append(np.zeros((1, 2, 4, 2), np.float32)); valids.append(np.zeros(2, np.float32))
        else:
            preds.append(ph.float().cpu().numpy()); valids.append(fv)
        gts.append(gt)

    pending = []
    for i, s in enumerate(samples):
        if i % world != rank:
            continue
        if batch_size <= 1:
            score(s, inference(model, s, **gen_kw)[1])
            continue
        pending.append(s)
        if len(pending) == batch_size:
            for ss, ph in zip(pending, inference_batch(model, pending, **gen_kw)):
                score(ss, ph)
            pending = []
    if pending:
        for ss, ph in zip(pending, inference_batch(model, pending, **gen_kw)):
            score(ss, ph)
    preds = np.stack(preds) if preds else np.zeros((0, 1, 2, 4, 2), np.float32)
    gts = np.stack(gts) if gts else np.zeros((0, 2, 4, 2), np.float32)
    valids = np.stack(valids) if valids else np.zeros((0, 2), np.float32)
    if world > 1:                                                  # handsonvlm_trainer.py:241-251 accelerator.gather
        objs = [None] * world
        dist.all_gather_object(objs, (preds, gts, valids))
        preds, gts, valids = (np.concatenate([o[j] for o in objs]) for j in range(3))
    if valids.sum() == 0:
        return dict(ade=float("nan"), fde=float("nan"), wde=float("nan"), n=int(len(gts)))
    a, f, w = evaluate_traj_stochastic(preds, gts, valids, reference_labels=reference_labels)
    return dict(ade=float(a), fde=float(f), wde=float(w), n=int(len(gts)))
