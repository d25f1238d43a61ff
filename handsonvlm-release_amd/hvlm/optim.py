"""AdamW for the kernel-side parameters (fp32 master weights and moments, bf16 gradients) plus the
PyTorch-side CVAE head.  Recipe of scripts/finetune.sh:30-33: lr 2e-5, cosine decay, 3 % warm-up,
weight decay 0, betas (0.9, 0.999), eps 1e-8 (HF Trainer defaults).

One fused HIP kernel per tensor (hvlm_adamw_step) updates master/m/v, writes the bf16 working copy
and — for GEMM weights — its transposed copy used by the dX GEMMs, so no separate transpose pass
runs after the step.

Overlapped mode (`overlap=True`, GPU only): the update of a gradient bucket (one decoder layer; lm_head; the
embed / norm / projector tail) is enqueued on the optimizer's own HIP stream the moment that bucket is final —
after its all-reduce under DP, after its backward kernels otherwise — so the HBM-bound update (≈190 GB of state
traffic per 7B step) runs beside the MFMA-bound backward of the earlier layers instead of after it.  The reference
does not clip gradients (scripts/zero3.json has no `gradient_clipping`, so DeepSpeed's default 0 applies): no
global-norm dependency forces the update to wait for the whole backward, and the result is bit-identical to the
end-of-step update."""
from __future__ import annotations

import math
from typing import List, Optional

import torch

from . import ops


def cosine_lr(step: int, total_steps: int, base_lr: float, warmup_ratio: float = 0.03) -> float:
    """HF `cosine` schedule with linear warm-up: lr_lambda(step) of transformers.get_cosine_schedule_with_warmup with
    num_warmup_steps = ceil(total * warmup_ratio) (HF Trainer).  `step` is the number of optimizer steps ALREADY taken
    (0 for the first update): the HF scheduler is stepped after optimizer.step(), so the first update runs at lr_lambda(0)
    (= 0 when there is a warm-up) and the last at lr_lambda(total - 1) > 0."""
    warm = math.ceil(total_steps * warmup_ratio)
    if step < warm:
        return base_lr * step / max(1, warm)
    prog = (step - warm) / max(1, total_steps - warm)
    return base_lr * max(0.0, 0.5 * (1.0 + math.cos(math.pi * prog)))


class AdamW:
    def __init__(self, model, lr: float = 2e-5, betas=(0.9, 0.999), eps: float = 1e-8, weight_decay: float = 0.0,
                 overlap: bool = False):
        self.model = model
        self.lr, self.betas, self.eps, self.wd = lr, betas, eps, weight_decay
        self.step_count = 0
        self.state = []
        for name, w, g, wt in model.trainable_tensors():
            master = w.to(torch.float32).clone()
            self.state.append(dict(name=name, w=w, g=g, wt=wt, master=master, m=torch.zeros_like(master),
                                   v=torch.zeros_like(master)))
        self.head = torch.optim.AdamW(model.hand_traj_decoder.parameters(), lr=lr, betas=betas, eps=eps,
                                      weight_decay=weight_decay)
        # gradient-bucket key -> its tensors (the keys dp.attach's hooks report: "layers.<i>", "head", "tail")
        self.groups = {}
        for st in self.state:
            n = st["name"]
            key = n.rsplit(".", 1)[0] if n.startswith("layers.") else ("head" if n == "lm_head" else "tail")
            self.groups.setdefault(key, []).append(st)
        self.overlap = bool(overlap) and torch.cuda.is_available()
        self.stream = torch.cuda.Stream() if self.overlap else None
        self._open = None                  # (lr, grad_scale) between begin_step() and step()
        self._done = set()

    def _weights_changed(self):
        """The bf16 weights are about to be / were rewritten in place: the e4m3 copies made from them (fp8 evaluation, and the
        decode graphs captured on those copies) are stale — drop them; they are rebuilt lazily at the next fp8 call."""
        stack = getattr(self.model, "stack", None)
        if stack is not None and hasattr(stack, "invalidate_fp8"):
            stack.invalidate_fp8()

    def _update(self, sts, lr, grad_scale):
        b1, b2 = self.betas
        for st in sts:
            ops.adamw_step_(st["master"], st["m"], st["v"], st["g"], st["w"], st["wt"], lr, b1, b2, self.eps, self.wd,
                            grad_scale, self.step_count)

    def begin_step(self, lr: Optional[float] = None, grad_scale: float = 1.0):
        """Overlapped mode: call before backward; fixes this step's lr / grad_scale for the per-bucket updates."""
        self.step_count += 1
        self._open = (self.lr if lr is None else lr, grad_scale)
        self._done = set()
        self._weights_changed()

    def bucket_ready(self, key: str, after=(), post=None):
        """The gradients of bucket `key` are final once the work handles in `after` (its all-reduces) complete, or —
        without handles — once the kernels enqueued so far on the current stream have run: update it on the optimizer
        stream.  No-op outside overlapped mode / outside begin_step()..step()."""
        if not self.overlap or self._open is None or key not in self.groups or key in self._done:
            return
        lr, grad_scale = self._open
        cur = torch.cuda.current_stream()
        with torch.cuda.stream(self.stream):
            if after:
                for h in after:
                    h.wait()               # NCCL/RCCL work: orders THIS stream after the collective, the host does not block
            else:
                self.stream.wait_stream(cur)
            if post is not None:
                post()                     # e.g. dp.EmbedRows.finish: scatter the reduced rows back, behind their all-reduce
            self._update(self.groups[key], lr, grad_scale)
        self._done.add(key)

    def step(self, lr: Optional[float] = None, grad_scale: float = 1.0):
        """grad_scale multiplies every gradient (1/world_size after a SUM all-reduce).  After begin_step(): finishes
        the step (buckets not yet updated + the CVAE head) and joins the optimizer stream."""
        self._weights_changed()
        if self._open is None:
            self.step_count += 1
            lr = self.lr if lr is None else lr
            rest = self.state
        else:
            lr, grad_scale = self._open
            rest = [st for k, sts in self.groups.items() if k not in self._done for st in sts]
            self._open = None
        self._update(rest, lr, grad_scale)
        if self.overlap:
            torch.cuda.current_stream().wait_stream(self.stream)   # next forward reads the weights written there
        for grp in self.head.param_groups:
            grp["lr"] = lr
        if grad_scale != 1.0:
            for p in self.model.hand_traj_decoder.parameters():
                if p.grad is not None:
                    p.grad.mul_(grad_scale)
        self.head.step()

    def memory_bytes(self) -> int:
        return sum(st["master"].numel() * 12 for st in self.state)

    # -- checkpoint / resume (the reference resumes the HF Trainer's optimizer + scheduler state,
    #    handsonvlm/train/train.py:399-402) ------------------------------------------------------
    def reload_from_model(self):
        """Re-derive the fp32 master weights from the model's bf16 tensors (after loading WEIGHTS ONLY into a model this
        optimizer was already built for; moments and the step count are reset)."""
        for st in self.state:
            st["master"].copy_(st["w"].to(torch.float32))
            st["m"].zero_(); st["v"].zero_()
        self.step_count = 0
        self._weights_changed()                     # (the weights were loaded behind the optimizer's back)

    def state_groups(self):
        """Yields (bucket key, {tensor name: {"master","m","v"}}) — one checkpoint shard per gradient bucket."""
        for key, sts in self.groups.items():
            yield key, {st["name"]: {"master": st["master"], "m": st["m"], "v": st["v"]} for st in sts}

    def meta_state(self) -> dict:
        return {"step_count": self.step_count, "head": self.head.state_dict()}

    def load_group(self, key: str, tensors: dict):
        self._weights_changed()
        for st in self.groups[key]:
            src = tensors[st["name"]]
            for k in ("master", "m", "v"):
                st[k].copy_(src[k].to(st[k].device))
            # the bf16 working copy (and its transposed copy) always follow the fp32 master
            st["w"].copy_(st["master"].to(st["w"].dtype).view_as(st["w"]))
            if st["wt"] is not None:
                ops.transpose(st["w"], out=st["wt"])

    def load_meta_state(self, meta: dict):
        self.step_count = int(meta["step_count"])
        self.head.load_state_dict(meta["head"])
        self._open, self._done = None, set()
