"""SURVEY.md §8 f1 + f2 joined at the entry points (round-3 verdict item 7), on CPU through the ops stand-in (tests/cpu_ops.py):

  * `python -m hvlm.train` takes the command line of scripts/finetune.sh (:6-60): `--model_name_or_path` +
    `--pretrain_mm_mlp_adapter` load through hvlm.checkpoint like handsonvlm/train/train.py:235-240 + llava_arch.py:64-69
    instead of init_random; the recipe flags (`--bf16 --weight_decay --warmup_ratio --lr_scheduler_type --model_max_length
    --gradient_checkpointing --hoi_lambda ...`) reach the config / the loop; flags of out-of-scope subsystems are listed once
    and ignored;
  * a BASE LLM directory (32000-row embeddings, no projector / tower / head tensors — what `./checkpoints/vicuna-v1-3-7b` is)
    plus the adapter file plus a local CLIP directory loads, its vocabulary extended to 32101 rows like
    handsonvlm/model/handsonvlm_arch.py:17-31 does;
  * `python -m hvlm.evaluate --model-path ... --batch-size ...` (handsonvlm/evaluation/evaluate.py:18-26) loads the same
    fixture and returns the metric dict.
Fixture: tests/golden/ckpt_tiny (written from the reference model by oracle/gen_golden_ckpt.py)."""
import json
import os

import numpy as np
import pytest
import torch

from test_ckpt_reference_cpu import inflate

FINETUNE_SH_FLAGS = """--deepspeed ./scripts/zero3.json --version vicuna-v1-3-7b --data_path none --mm_vision_select_layer -2
 --mm_use_im_start_end False --mm_use_im_patch_token False --bf16 True --epic_kitchen_use_percentage 1
 --gradient_accumulation_steps 1 --per_device_eval_batch_size 8 --evaluation_strategy no --logging_strategy epoch
 --report_to tensorboard --logging_steps 5 --save_strategy epoch --save_steps 5 --save_total_limit 100 --learning_rate 2e-5
 --weight_decay 0. --warmup_ratio 0.03 --lr_scheduler_type cosine --tf32 True --model_max_length 2048
 --gradient_checkpointing True --dataloader_num_workers 0 --lazy_preprocess True --tasks dvc||event_loc||epic_kitchen
 --task_sample_rate 1 1 1 1 1 5 --hoi_lambda 1 --fuse_input_mode origin --video_arch temporal_spatial_pool
 --traj_decoder CVAE --backbone handsonvlm --ddp_backend gloo""".split()


def test_train_entry_point_takes_finetune_sh_flags_and_loads_the_checkpoint(tmp_path, monkeypatch, capsys):
    import cpu_ops
    cpu_ops.install(monkeypatch)
    from hvlm import checkpoint, train
    d = inflate(str(tmp_path / "ck"))
    out = str(tmp_path / "out")
    seen = {}
    real = checkpoint.from_pretrained

    def spy(path, **kw):
        m = real(path, **kw)
        seen.update(path=path, kw=kw, model=m, w0=m.stack.layers[0].w["qkv"].clone())
        return m
    monkeypatch.setattr(checkpoint, "from_pretrained", spy)
    torch.manual_seed(0)
    log = train.main(["--model_name_or_path", d, "--pretrain_mm_mlp_adapter", os.path.join(d, "mm_projector.bin"),
                      "--vision_tower", "openai/clip-vit-large-patch14", "--output_dir", out, "--num_train_epochs", "1",
                      "--per_device_train_batch_size", "2", "--samples_per_epoch", "4", "--device", "cpu"] + FINETUNE_SH_FLAGS)
    txt = capsys.readouterr().out
    assert "flags accepted for command-line compatibility" in txt and "--deepspeed" in txt and "--task_sample_rate" in txt
    assert txt.count("accepted for command-line compatibility") == 1                       # listed once
    m = seen["model"]
    # loaded, not init_random: the geometry and the tensors are the fixture's; the recipe flags reached the config
    assert (m.config.hidden_size, m.config.num_hidden_layers, m.config.vocab_size) == (128, 2, 32101)
    sd = checkpoint.load_state(d)
    assert torch.equal(seen["w0"][:128].float(), sd["model.layers.0.self_attn.q_proj.weight"].to(torch.bfloat16).float())
    assert m.config.hoi_lambda == 1.0 and m.config.max_position_embeddings == 2048 and m.config.activation_recompute is True
    assert m.stack.recompute == m.config.num_hidden_layers                                 # --gradient_checkpointing True
    assert m.config.lm_head_labelled_rows is False                                          # the reference's shapes unless asked
    # 4 samples / 2 per batch = 2 steps of one epoch; per-epoch checkpoint (save_strategy epoch: --save_steps 5 is inert, as in HF)
    assert [r["step"] for r in log] == [1, 2] or [r["step"] for r in log] == [2]
    assert all(np.isfinite(r["loss"]) for r in log)
    assert os.path.exists(os.path.join(out, "checkpoint-2", checkpoint.TRAINER_STATE))
    assert not torch.equal(m.stack.layers[0].w["qkv"], seen["w0"])                         # it trained
    # the written checkpoint loads back through the evaluation entry point
    from hvlm import evaluate
    res = evaluate.main(["--model-path", os.path.join(out, "checkpoint-2"), "--device", "cpu", "--synthetic", "2", "--batch-size", "2",
                         "--max-new-tokens", "3", "--conv-mode", "v1", "--mode", "general"])
    assert set(res) == {"ade", "fde", "wde", "n"} and res["n"] == 2
    assert json.loads(capsys.readouterr().out.strip().splitlines()[-1]) == json.loads(json.dumps(res))


def test_base_llm_plus_adapter_plus_clip_directory(tmp_path, monkeypatch):
    import cpu_ops
    cpu_ops.install(monkeypatch)
    from hvlm import checkpoint
    d = inflate(str(tmp_path / "ck"))
    full = checkpoint.load_state(d)
    vt = "model.vision_tower.vision_tower."
    # a base-LLM directory as `--model_name_or_path ./checkpoints/vicuna-v1-3-7b` is: LLM tensors only, 32000-row embeddings
    base = {k: v for k, v in full.items() if k.startswith(("model.layers.", "model.norm.")) or k in ("model.embed_tokens.weight", "lm_head.weight")}
    base["model.embed_tokens.weight"] = base["model.embed_tokens.weight"][:32000].clone()
    base["lm_head.weight"] = base["lm_head.weight"][:32000].clone()
    bdir, cdir = str(tmp_path / "vicuna"), str(tmp_path / "clip")
    os.makedirs(bdir); os.makedirs(cdir)
    torch.save(base, os.path.join(bdir, "pytorch_model.bin"))
    raw = json.load(open(os.path.join(d, "config.json")))
    raw["vocab_size"] = 32000
    for k in ("mm_hidden_size", "mm_vision_tower", "vision_config"):
        raw.pop(k, None)
    json.dump(raw, open(os.path.join(bdir, "config.json"), "w"))
    torch.save({k[len(vt):]: v for k, v in full.items() if k.startswith(vt)}, os.path.join(cdir, "pytorch_model.bin"))
    adapter = os.path.join(d, "mm_projector.bin")
    with pytest.raises(FileNotFoundError, match="mm_projector"):
        checkpoint.from_pretrained(bdir, device="cpu", vision_tower_path=cdir)
    with pytest.raises(FileNotFoundError, match="CLIP tower"):
        checkpoint.from_pretrained(bdir, device="cpu", mm_projector_bin=adapter)
    with pytest.raises(FileNotFoundError, match="not a local directory"):
        checkpoint.from_pretrained(bdir, device="cpu", mm_projector_bin=adapter, vision_tower_path="openai/clip-vit-large-patch14")
    m = checkpoint.from_pretrained(bdir, device="cpu", training=True, mm_projector_bin=adapter, vision_tower_path=cdir)
    assert m.config.vocab_size == 32101 and m.embed.shape[0] == 32101
    assert (m.config.vision.hidden_size, m.config.vision.num_hidden_layers, m.config.mm_hidden_size) == (64, 3, 64)
    assert torch.equal(m.embed[:32000].float(), full["model.embed_tokens.weight"][:32000].to(torch.bfloat16).float())
    new = m.embed[32000:].float()
    assert 0.015 < float(new.std()) < 0.025 and abs(float(new.mean())) < 2e-3             # HF initializer_range = 0.02
    assert float(m.lm_head[32101:].abs().max()) == 0.0                                     # the K-tile padding rows stay zero
    ad = torch.load(adapter, map_location="cpu", weights_only=True)
    wkey = [k for k in ad if k.endswith("mm_projector.weight")][0]
    assert torch.equal(m.proj_w.float(), ad[wkey].to(torch.bfloat16).float())
    assert m.tower.is_loaded
    # and it runs: a forward with labels (the <hand_traj> id 32100 is one of the NEW rows) gives a finite loss
    from test_ckpt_reference_cpu import batch
    ids, labels, mask, images, fh, fv = batch()
    out = m(input_ids=ids, attention_mask=mask, labels=labels, image=images.to(torch.bfloat16), future_hands=fh.clone(),
            contact_point=torch.zeros(2, 2), future_valid=fv.clone(), gt_label_valid=torch.ones(2))
    assert np.isfinite(float(out.loss.detach()))
