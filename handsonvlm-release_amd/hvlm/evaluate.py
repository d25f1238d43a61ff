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


def main(argv=None, samples=None):
    """`python -m hvlm.evaluate --model-path <ckpt> [--batch-size N]` — the entry point of handsonvlm/evaluation/evaluate.py:18-26
    (`--model-path`, `--conv-mode`, `--load-8bit/4bit`, `--use_reason`, `--mode` accepted).  Loads the checkpoint through
    hvlm.checkpoint.from_pretrained (handsonvlm/model/builder.py:18) and prints the metric dict the reference's
    evaluate_epic_kitchen_traj reports.  Dataset readers are out of scope: the evaluation samples come from
    `--samples-pt` (a torch-saved list of dicts with input_ids [1,T], image [1,100,3,H,W], future_hands [1,2,5,2],
    future_valid [1,2] — what the reference's EpicConversationDataset + collator hand to `inference()`), from `samples`, or
    are synthetic (`--synthetic N`)."""
    import argparse
    import json
    from . import checkpoint, dp
    ap = argparse.ArgumentParser(description="HandsOnVLM trajectory evaluation on MI355X", allow_abbrev=False)
    ap.add_argument("--model-path", required=True)
    ap.add_argument("--conv-mode", default=None)
    ap.add_argument("--load-8bit", action="store_true")
    ap.add_argument("--load-4bit", action="store_true")
    ap.add_argument("--use_reason", action="store_true")
    ap.add_argument("--mode", default="general")
    ap.add_argument("--batch-size", type=int, default=1, help="1 = the reference's recipe; > 1 generates that many samples in lock step")
    ap.add_argument("--samples-pt", default=None)
    ap.add_argument("--synthetic", type=int, default=None, metavar="N")
    ap.add_argument("--temperature", type=float, default=0.5)
    ap.add_argument("--top-p", type=float, default=0.9)
    ap.add_argument("--max-new-tokens", type=int, default=100)
    ap.add_argument("--fp8", action="store_true", help="e4m3 prefill + decode + CLIP GEMMs (BASELINE config 5; needs --batch-size > 1)")
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--device", default="cuda")
    a = ap.parse_args(argv)
    assert not (a.load_8bit or a.load_4bit), "bitsandbytes quantised loading is not part of the hot path"
    rank, local, world = dp.init_from_env()
    device = torch.device("cuda", local) if a.device == "cuda" else torch.device(a.device)
    model = checkpoint.from_pretrained(a.model_path, device=device, training=False)
    model.eval()
    if samples is None and a.samples_pt:
        samples = torch.load(a.samples_pt, map_location="cpu", weights_only=False)
    if samples is None:
        from .data import SyntheticEpicDataset
        ds = SyntheticEpicDataset(a.synthetic or 8, seed=a.seed, image_size=model.config.vision.image_size)
        samples = []
        for i in range(len(ds)):
            s = ds[i]
            n_prompt = int((s["labels"] == -100).sum())               # the prompt = the masked prefix of the conversation
            samples.append(dict(input_ids=s["input_ids"][None, :max(n_prompt, 2)], image=s["image"][None],
                                future_hands=s["future_hands"][None], future_valid=s["future_valid"][None]))
    torch.manual_seed(a.seed + rank)
    kw = dict(temperature=a.temperature, top_p=a.top_p, max_new_tokens=a.max_new_tokens)
    if a.fp8:
        assert a.batch_size > 1, "--fp8 runs through the batched generation path"
        v = model.config.vision                      # (the e4m3 GEMM needs K % 128 == 0: every tower GEMM of ViT-L/14 qualifies)
        kw.update(fp8=True, fp8_decode=True, fp8_clip=(v.hidden_size % 128 == 0 and v.intermediate_size % 128 == 0))
    # under torchrun every rank scores samples[rank::world] (evaluate_traj) and the predictions are all-gathered for the metric:
    # N ranks = 1/N of the work each, one metric (handsonvlm_trainer.py:241-251)
    res = evaluate_traj(model, samples, batch_size=a.batch_size, **kw)
    if rank == 0:
        print(json.dumps(res), flush=True)
    return res


if __name__ == "__main__":
    main()
