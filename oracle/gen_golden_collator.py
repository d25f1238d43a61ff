#!/usr/bin/env python3
"""Generates tests/golden/g9_collator.npz by running the REFERENCE's batch contract (build container only):
  * `DataCollatorForSupervisedDataset` (handsonvlm/dataset/hybrid_dataset.py:108-169) on seeded instances that exercise
    every branch: a single-image sample without HOI keys (zero-filled in bf16, image tiled to 100 frames), a full EK100
    sample (python-bool gt_label_valid, fp32 hands), a sequence longer than model_max_length (truncation), ragged lengths
    (right padding with pad_token_id / -100, attention_mask = ids != pad);
  * the EK100 frame tiling of handsonvlm/dataset/epic_dataset.py:90-95 is exercised through hvlm.data.tile_epic_frames in
    the test against the explicit index pattern recorded here.
Instances are regenerated from seeds by the test (oracle/synth.collator_instances); the fixture stores only the reference's
OUTPUT (ids / labels / mask in full, images as strided slices + checksums, dtypes as strings)."""
import os
import sys
import types

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import shim  # noqa: E402

shim.install()
for name, attrs in (("decord", dict(VideoReader=None, cpu=None)), ("lmdbdict", dict(lmdbdict=object)),
                    ("lmdbdict.methods", dict(DUMPS_FUNC={}, LOADS_FUNC={}))):
    m = types.ModuleType(name); m.__dict__.update(attrs); sys.modules.setdefault(name, m)

import numpy as np  # noqa: E402
import torch  # noqa: E402

from handsonvlm.dataset.hybrid_dataset import DataCollatorForSupervisedDataset  # noqa: E402
from oracle.synth import collator_instances  # noqa: E402


class Tok:
    pad_token_id = 0
    model_max_length = 40


def main():
    out = {}
    for case, inst in enumerate(collator_instances()):
        batch = DataCollatorForSupervisedDataset(tokenizer=Tok())(inst)
        for k, v in batch.items():
            if k == "prompt":
                out[f"c{case}.prompt"] = np.array([str(p) for p in v])
            elif k == "image":
                assert torch.is_tensor(v)
                out[f"c{case}.image_slice"] = v[:, :, :, ::56, ::56].float().numpy()
                out[f"c{case}.image_sum"] = v.float().sum(dim=(2, 3, 4)).numpy()
                out[f"c{case}.image_dtype"] = np.array(str(v.dtype))
            elif torch.is_tensor(v):
                out[f"c{case}.{k}"] = v.float().numpy() if v.is_floating_point() else v.numpy()
                out[f"c{case}.{k}_dtype"] = np.array(str(v.dtype))
            else:
                raise AssertionError(f"unstackable key {k} in case {case}")
        out[f"c{case}.keys"] = np.array(list(batch.keys()))
    p = os.path.join(ROOT, "tests", "golden", "g9_collator.npz")
    np.savez_compressed(p, **out)
    print(f"wrote {p}: {os.path.getsize(p) / 1024:.1f} KiB", sorted(k for k in out if k.startswith('c0.')))


if __name__ == "__main__":
    main()
