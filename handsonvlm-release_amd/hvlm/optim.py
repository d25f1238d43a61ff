"""AdamW for the kernel-side parameters (fp32 master weights and moments, bf16 gradients) plus the
PyTorch-side CVAE head.  Recipe of scripts/finetune.sh:30-33: lr 2e-5, cosine decay, 3 % warm-up,
weight decay 0, betas (0.9, 0.999), eps 1e-8 (HF Trainer defaults).

One fused HIP kernel per tensor (hvlm_adamw_step) updates master/m/v, writes the bf16 working copy
and — for GEMM weights — its transposed copy used by the dX GEMMs, so no separate transpose pass
runs after the step."""
from __future__ import annotations

import math
from typing import List, Optional

import torch

from . import ops


def cosine_lr(step: int, total_steps: int, base_lr: float, warmup_ratio: float = 0.03) -> float:
    """HF `cosine` schedule with linear warm-up (get_cosine_schedule_with_warmup)."""
    warm = math.ceil(total_steps * warmup_ratio)
    if step < warm:
        return base_lr * step / max(1, warm)
    prog = (step - warm) / max(1, total_steps - warm)
    return base_lr * max(0.0, 0.5 * (1.0 + math.cos(math.pi * prog)))


class AdamW:
    def __init__(self, model, lr: float = 2e-5, betas=(0.9, 0.999), eps: float = 1e-8, weight_decay: float = 0.0):
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

    def step(self, lr: Optional[float] = None, grad_scale: float = 1.0):
        """grad_scale multiplies every gradient (1/world_size after a SUM all-reduce)."""
        self.step_count += 1
        lr = self.lr if lr is None else lr
        b1, b2 = self.betas
        for st in self.state:
            ops.adamw_step_(st["master"], st["m"], st["v"], st["g"], st["w"], st["wt"], lr, b1, b2, self.eps, self.wd,
                            grad_scale, self.step_count)
        for grp in self.head.param_groups:
            grp["lr"] = lr
        if grad_scale != 1.0:
            for p in self.model.hand_traj_decoder.parameters():
                if p.grad is not None:
                    p.grad.mul_(grad_scale)
        self.head.step()

    def memory_bytes(self) -> int:
        return sum(st["master"].numel() * 12 for st in self.state)
