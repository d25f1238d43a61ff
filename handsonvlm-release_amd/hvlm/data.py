"""Batch contract of the reference's data layer — the parts that touch the hot path.

* `DataCollatorForSupervisedDataset`: mirror of handsonvlm/dataset/hybrid_dataset.py:108-169 (right-pad ids with
  pad_token_id and labels with -100, truncate to model_max_length, zero-fill the missing HOI keys, tile a single image
  to 100 frames, stack what can be stacked).
* `tile_epic_frames`: the EK100 10 -> 100 frame tiling of handsonvlm/dataset/epic_dataset.py:90-95 ([f0..f9] x 10),
  which is what makes `HandsOnVLMConfig.dedup_frames` exact.
* `SyntheticEpicDataset`: EK100-shaped random samples (no dataset / tokenizer is available offline).
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import Dict, List, Sequence

import numpy as np
import torch

from .config import HAND_TOKEN_ID, IGNORE_INDEX, IMAGE_TOKEN_INDEX


def tile_epic_frames(frames10: torch.Tensor) -> torch.Tensor:
    """[10,3,H,W] -> [100,3,H,W] as the reference does: the 10-frame sequence repeated 10 times."""
    assert frames10.shape[0] == 10, frames10.shape
    return frames10.unsqueeze(0).repeat(10, 1, 1, 1, 1).reshape(100, *frames10.shape[1:])


@dataclass
class DataCollatorForSupervisedDataset:
    pad_token_id: int = 0
    model_max_length: int = 2048
    num_frames: int = 100
    image_size: int = 224

    def __call__(self, instances: Sequence[Dict]) -> Dict[str, object]:
        assert "input_ids" in instances[0] and "labels" in instances[0]
        assert instances[0]["input_ids"].shape == instances[0]["labels"].shape
        dt = torch.bfloat16
        defaults = {"feat": lambda: torch.zeros(5, 10, 1024, dtype=dt), "bbox_feat": lambda: torch.zeros(4, 10, 4, dtype=dt),
                    "valid_mask": lambda: torch.zeros(5, 10, dtype=torch.bool),
                    "future_hands": lambda: torch.zeros(2, 5, 2, dtype=dt), "contact_point": lambda: torch.zeros(2, dtype=dt),
                    "future_valid": lambda: torch.zeros(2, dtype=torch.bool),
                    "gt_label_valid": lambda: torch.tensor(False)}
        for inst in instances:
            for k, mk in defaults.items():
                if k not in inst:
                    inst[k] = mk()
            if isinstance(inst["gt_label_valid"], bool):
                inst["gt_label_valid"] = torch.tensor(inst["gt_label_valid"])
            inst.setdefault("prompt", "")
            if inst["image"].dim() == 3:                      # single image -> 100 identical frames
                inst["image"] = inst["image"].unsqueeze(0).repeat(self.num_frames, 1, 1, 1)
            assert inst["image"].shape == (self.num_frames, 3, self.image_size, self.image_size), inst["image"].shape
            assert inst["future_hands"].shape == (2, 5, 2) and inst["future_valid"].shape == (2,)
        ids = torch.nn.utils.rnn.pad_sequence([i["input_ids"] for i in instances], batch_first=True, padding_value=self.pad_token_id)
        labels = torch.nn.utils.rnn.pad_sequence([i["labels"] for i in instances], batch_first=True, padding_value=IGNORE_INDEX)
        ids, labels = ids[:, : self.model_max_length], labels[:, : self.model_max_length]
        batch = dict(input_ids=ids, labels=labels, attention_mask=ids.ne(self.pad_token_id),
                     prompt=[np.array(i["prompt"]) for i in instances])
        for k in ("feat", "bbox_feat", "valid_mask", "future_hands", "contact_point", "future_valid", "gt_label_valid", "image"):
            vals = [i[k] for i in instances]
            batch[k] = torch.stack(vals) if all(v.shape == vals[0].shape for v in vals) else vals
        return batch


class SyntheticEpicDataset(torch.utils.data.Dataset):
    """Random EK100-shaped samples: a prompt with one <image> marker, an answer ending in four <hand_traj> tokens."""

    def __init__(self, n: int, min_len: int = 64, max_len: int = 192, image_size: int = 224, seed: int = 0):
        self.n, self.min_len, self.max_len, self.image_size, self.seed = n, min_len, max_len, image_size, seed

    def __len__(self):
        return self.n

    def __getitem__(self, i):
        r = np.random.RandomState(self.seed * 100003 + i)
        T = int(r.randint(self.min_len, self.max_len + 1))
        ids = r.randint(3, 32000, size=T).astype(np.int64)
        ids[35 if T > 45 else 1] = IMAGE_TOKEN_INDEX
        for j in range(4):
            ids[T - 2 - 2 * (3 - j)] = HAND_TOKEN_ID
        labels = ids.copy()
        labels[: int(0.4 * T)] = IGNORE_INDEX
        labels[ids == IMAGE_TOKEN_INDEX] = IGNORE_INDEX
        frames = torch.from_numpy(r.standard_normal((10, 3, self.image_size, self.image_size)).astype(np.float32)).to(torch.bfloat16)
        return dict(input_ids=torch.from_numpy(ids), labels=torch.from_numpy(labels), image=tile_epic_frames(frames),
                    future_hands=torch.from_numpy(r.uniform(0.05, 0.95, (2, 5, 2)).astype(np.float32)),
                    contact_point=torch.zeros(2), future_valid=torch.ones(2), gt_label_valid=torch.tensor(True), prompt="synthetic")
