"""Synthetic EK100-shaped minibatches (no dataset / tokenizer / weights are available offline):
uniform token ids with one <image> marker (-200), labels -100 on the prompt part, four <hand_traj>
tokens (32100) at the tail of the answer — the batch contract of DataCollatorForSupervisedDataset
(handsonvlm/dataset/hybrid_dataset.py:108-169)."""
from __future__ import annotations

import numpy as np

from .config import HAND_TOKEN_ID, IGNORE_INDEX, IMAGE_TOKEN_INDEX


def make_text_batch_np(B: int, T: int, seed: int = 0, image_pos: int = 35, n_hand: int = 4, label_from: float = 0.4):
    r = np.random.RandomState(seed)
    ids = r.randint(3, 32000, size=(B, T)).astype(np.int64)
    ids[:, min(image_pos, T - 10)] = IMAGE_TOKEN_INDEX
    for j in range(n_hand):
        ids[:, T - 2 - 2 * (n_hand - 1 - j)] = HAND_TOKEN_ID
    labels = ids.copy()
    labels[:, : int(label_from * T)] = IGNORE_INDEX
    labels[ids == IMAGE_TOKEN_INDEX] = IGNORE_INDEX
    return ids, labels
