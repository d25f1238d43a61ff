"""CPU tests of the rows SURVEY.md §8f marks "next": batch contract / collator, EK100 frame tiling, checkpoint format."""
import json
import os

import numpy as np
import torch

from hvlm import checkpoint
from hvlm.config import HandsOnVLMConfig
from hvlm.data import DataCollatorForSupervisedDataset, SyntheticEpicDataset, tile_epic_frames


def test_tile_epic_frames_is_sequence_repeated():
    f = torch.arange(10).float().view(10, 1, 1, 1).expand(10, 3, 4, 4)
    t = tile_epic_frames(f)
    assert t.shape == (100, 3, 4, 4)
    assert t[:, 0, 0, 0].tolist() == [float(i % 10) for i in range(100)]      # [f0..f9] x 10 (epic_dataset.py:90-95)


def test_collator_contract():
    col = DataCollatorForSupervisedDataset(pad_token_id=0, model_max_length=12)
    a = dict(input_ids=torch.arange(1, 9), labels=torch.arange(1, 9), image=torch.zeros(3, 224, 224))          # single image
    b = dict(input_ids=torch.arange(1, 21), labels=torch.arange(1, 21), image=torch.ones(100, 3, 224, 224),
             future_hands=torch.rand(2, 5, 2), future_valid=torch.ones(2), gt_label_valid=True, prompt="p")
    batch = col([a, b])
    assert batch["input_ids"].shape == (2, 12) and batch["labels"].shape == (2, 12)                 # padded then truncated
    assert batch["input_ids"][0, 8:].eq(0).all() and batch["labels"][0, 8:].eq(-100).all()
    assert batch["attention_mask"][0].tolist() == [True] * 8 + [False] * 4
    assert batch["image"].shape == (2, 100, 3, 224, 224) and batch["image"][0].abs().sum() == 0     # tiled single image
    assert batch["future_hands"][0].abs().sum() == 0 and batch["future_valid"][0].sum() == 0         # zero-filled keys
    assert batch["feat"].shape == (2, 5, 10, 1024) and batch["gt_label_valid"].shape == (2,)
    assert len(batch["prompt"]) == 2


def test_synthetic_dataset_sample_shape():
    s = SyntheticEpicDataset(4, min_len=50, max_len=60)[2]
    assert s["image"].shape == (100, 3, 224, 224) and (s["input_ids"] == -200).sum() == 1 and (s["input_ids"] == 32100).sum() == 4
    assert torch.equal(s["image"][:10], s["image"][10:20])


def test_checkpoint_shards_roundtrip(tmp_path):
    sd = {f"model.layers.{i}.w": torch.randn(64, 64).to(torch.bfloat16) for i in range(6)}
    sd["lm_head.weight"] = torch.randn(10, 64)
    checkpoint.save_state(sd, str(tmp_path), max_shard_bytes=3 * 64 * 64 * 2)
    idx = json.load(open(tmp_path / checkpoint.INDEX))
    assert len(set(idx["weight_map"].values())) >= 3 and all(f.startswith("pytorch_model-0000") for f in idx["weight_map"].values())
    back = checkpoint.load_state(str(tmp_path))
    assert back.keys() == sd.keys() and all(torch.equal(back[k], sd[k]) for k in sd)
    one = tmp_path / "single"
    checkpoint.save_state(sd, str(one))
    assert os.path.exists(one / "pytorch_model.bin") and checkpoint.load_state(str(one)).keys() == sd.keys()


def test_config_json_roundtrip(tmp_path):
    cfg = HandsOnVLMConfig.vicuna_13b(lambda_traj_kl=5e-3)
    checkpoint.config_to_json(cfg, str(tmp_path))
    raw = json.load(open(tmp_path / "config.json"))
    assert raw["model_type"] == "handsonvlm" and raw["hidden_size"] == 5120 and raw["video_compress_mode"] == "temporal_spatial_pool"
    back = checkpoint.config_from_json(str(tmp_path))
    assert (back.hidden_size, back.num_hidden_layers, back.lambda_traj_kl, back.vision.hidden_size) == (5120, 40, 5e-3, 1024)


def test_prompt_plumbing_vs_reference_golden(golden):
    """tokenizer_image_token / preprocess_v1 (v1 template, label masking incl. the reference's off-by-two and the
    multi-round and padded cases) against vectors produced by the reference functions with the same stand-in
    tokenizer (oracle/gen_golden_prompt.py)."""
    import numpy as np
    from hvlm import prompt as P
    from oracle.synth import PROMPT_SOURCES, StandInTokenizer
    g = golden("g8_prompt.npz")
    tok = StandInTokenizer()
    assert P.tokenizer_image_token("A <image> b <image><image> c", tok) == g["tit"].tolist()
    for n, (sources, has_image) in enumerate(PROMPT_SOURCES):
        r = P.preprocess_v1(sources, tok, has_image=has_image)
        assert np.array_equal(r["input_ids"].numpy(), g[f"ids{n}"])
        assert np.array_equal(r["labels"].numpy(), g[f"labels{n}"])
        assert (r["labels"] != P.IGNORE_INDEX).any()          # the fixtures exercise the supervised branch, not "ignored"
    # image placeholder and hand tokens land where the splice / hand gather expect them
    ids = P.preprocess_v1(PROMPT_SOURCES[0][0], tok, has_image=True)["input_ids"][0]
    assert int((ids == P.IMAGE_TOKEN_INDEX).sum()) == 1 and int((ids == 32100).sum()) == 4


def test_collator_vs_reference_golden(golden):
    """hvlm.data.DataCollatorForSupervisedDataset against the OUTPUT OF THE REFERENCE CLASS
    (handsonvlm/dataset/hybrid_dataset.py:108-169, run by oracle/gen_golden_collator.py) on the same seeded instances:
    key set and order, values, dtypes (bf16 zero-fill, dtype promotion when present and zero-filled samples mix), padding,
    truncation at model_max_length, single-image tiling to 100 frames."""
    from oracle.synth import collator_instances
    g = golden("g9_collator.npz")
    col = DataCollatorForSupervisedDataset(pad_token_id=0, model_max_length=40)
    for case, inst in enumerate(collator_instances()):
        b = col(inst)
        assert list(b.keys()) == [str(k) for k in g[f"c{case}.keys"]]
        assert [str(p) for p in b["prompt"]] == [str(p) for p in g[f"c{case}.prompt"]]
        img = b["image"]
        assert str(img.dtype) == str(g[f"c{case}.image_dtype"])
        assert np.array_equal(img[:, :, :, ::56, ::56].float().numpy(), g[f"c{case}.image_slice"])
        assert np.array_equal(img.float().sum(dim=(2, 3, 4)).numpy(), g[f"c{case}.image_sum"])
        for k, v in b.items():
            if k in ("prompt", "image"):
                continue
            assert torch.is_tensor(v), k
            assert str(v.dtype) == str(g[f"c{case}.{k}_dtype"]), (k, v.dtype, g[f"c{case}.{k}_dtype"])
            want = g[f"c{case}.{k}"]
            got = v.float().numpy() if v.is_floating_point() else v.numpy()
            assert got.shape == want.shape and np.array_equal(got, want), (case, k)
    # EK100 tiling pattern (handsonvlm/dataset/epic_dataset.py:90-95): the 10-frame sequence repeated 10 times
    f = torch.arange(10).float().view(10, 1, 1, 1).expand(10, 3, 2, 2)
    assert tile_epic_frames(f)[:, 0, 0, 0].tolist() == [float(i % 10) for i in range(100)]
