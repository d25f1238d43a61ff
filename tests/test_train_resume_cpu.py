"""Host logic of the training loop, driven on CPU through the ops stand-in of tests/cpu_ops.py (test infrastructure):
  * auto-resume (reference: handsonvlm/train/train.py:399-402, HF Trainer resume of optimizer + scheduler state):
    train 8 steps uninterrupted vs. train -> interruption after step 6 -> resume from checkpoint-4 -> steps 5..8;
    losses, weights and optimizer state must be BIT-identical;
  * the cosine schedule against transformers' get_cosine_schedule_with_warmup;
  * a text-only sample inside an image batch gives exactly zero projector gradient (handsonvlm.py:234-244)."""
import math
import os

import numpy as np
import pytest
import torch

from oracle import synth


def _tiny(training=True):
    from hvlm.config import CLIPVisionConfig, HandsOnVLMConfig
    from hvlm.model import HandsOnVLMForCausalLM
    c = synth.TINY
    cfg = HandsOnVLMConfig(hidden_size=c["D"], intermediate_size=c["FFN"], num_hidden_layers=c["L"], num_attention_heads=c["H"],
                           mm_hidden_size=c["clip_d"],
                           vision=CLIPVisionConfig(hidden_size=c["clip_d"], intermediate_size=c["clip_mlp"],
                                                   num_hidden_layers=c["clip_layers"], num_attention_heads=c["clip_heads"]))
    cfg.dedup_frames = True
    m = HandsOnVLMForCausalLM(cfg, device="cpu", training=training)
    sd = dict(synth.to_torch(synth.llama_params(c)))
    sd.update({"model.vision_tower.vision_tower." + k: v for k, v in synth.to_torch(synth.clip_params(c)).items()})
    m.load_state_dict(sd)
    return m, cfg


def _snapshot(model, opt_dir):
    w = {n: t.clone() for n, t, _, _ in model.trainable_tensors()}
    w.update({"head." + n: p.detach().clone() for n, p in model.hand_traj_decoder.named_parameters()})
    o = {}
    for f in sorted(os.listdir(opt_dir)):
        o[f] = torch.load(os.path.join(opt_dir, f), map_location="cpu", weights_only=True)
    return w, o


def _flat(d, pre=""):
    for k, v in d.items():
        if isinstance(v, dict):
            yield from _flat(v, f"{pre}{k}.")
        elif torch.is_tensor(v):
            yield pre + str(k), v


def test_resume_is_bit_identical_to_uninterrupted_run(tmp_path, monkeypatch):
    import cpu_ops
    cpu_ops.install(monkeypatch)
    from hvlm.data import SyntheticEpicDataset
    from hvlm.train import train
    ds = SyntheticEpicDataset(6, min_len=50, max_len=60)        # 3 batches per epoch -> the interruption is mid-epoch
    kw = dict(per_device_batch=2, epochs=1, max_steps=8, lr=3e-3, log_every=1, warmup_ratio=0.25, save_every_epoch=False,
              save_steps=4, seed=5)
    torch.manual_seed(0)
    m, _ = _tiny()
    full = train(m, ds, output_dir=str(tmp_path / "a"), **kw)
    wa, oa = _snapshot(m, str(tmp_path / "a" / "checkpoint-8" / "optimizer"))
    torch.manual_seed(0)
    m1, _ = _tiny()
    part = train(m1, ds, output_dir=str(tmp_path / "b"), stop_after=6, **kw)             # killed after step 6
    assert [r["step"] for r in part] == [1, 2, 3, 4, 5, 6] and os.path.isdir(tmp_path / "b" / "checkpoint-4")
    assert not os.path.isdir(tmp_path / "b" / "checkpoint-6")
    torch.manual_seed(12345)                                                              # a fresh process has other RNG state
    m2, _ = _tiny()                                                                       # fresh model: pre-training weights
    rest = train(m2, ds, output_dir=str(tmp_path / "b"), **kw)
    assert [r["step"] for r in rest] == [5, 6, 7, 8]
    for a, b in zip(full[4:], rest):
        assert a["loss"] == b["loss"] and a["traj_loss"] == b["traj_loss"], (a, b)
    assert [r["loss"] for r in part] == [r["loss"] for r in full[:6]]
    wb, ob = _snapshot(m2, str(tmp_path / "b" / "checkpoint-8" / "optimizer"))
    for k in wa:
        assert torch.equal(wa[k], wb[k]), k
    fa, fb = dict(_flat(oa)), dict(_flat(ob))
    assert fa.keys() == fb.keys() and len(fa) > 30
    for k in fa:
        assert torch.equal(fa[k], fb[k]), k
    # and the run did continue rather than jump back: the resumed step-5 loss is not the step-1 loss
    assert rest[0]["loss"] != full[0]["loss"]


def test_resume_refuses_weights_only_checkpoint(tmp_path, monkeypatch):
    import cpu_ops
    cpu_ops.install(monkeypatch)
    from hvlm import checkpoint
    from hvlm.optim import AdamW
    m, _ = _tiny()
    checkpoint.save_pretrained(m, str(tmp_path / "checkpoint-3"))
    with pytest.raises(FileNotFoundError):
        checkpoint.load_training_state(AdamW(m), str(tmp_path / "checkpoint-3"))


def test_cosine_schedule_matches_transformers():
    from transformers import get_cosine_schedule_with_warmup
    from hvlm.optim import cosine_lr
    for total, ratio in ((40, 0.03), (100, 0.03), (8, 0.25), (10, 0.0)):
        warm = math.ceil(total * ratio)                           # HF TrainingArguments.get_warmup_steps
        p = torch.nn.Parameter(torch.zeros(1))
        o = torch.optim.SGD([p], lr=2e-5)
        sch = get_cosine_schedule_with_warmup(o, warm, total)
        want = []
        for _ in range(total):
            want.append(o.param_groups[0]["lr"])                  # the lr optimizer.step() number k runs at
            o.step(); sch.step()
        got = [cosine_lr(k, total, 2e-5, ratio) for k in range(total)]
        assert np.allclose(got, want, rtol=1e-12, atol=0), (total, ratio)
        assert got[-1] > 0 and (warm == 0 or got[0] == 0.0)


def test_text_only_sample_gets_zero_projector_gradient(monkeypatch):
    import cpu_ops
    cpu_ops.install(monkeypatch)
    m, cfg = _tiny()
    c = synth.TINY
    ids, labels, mask = synth.make_text_batch(2, 24, seed=3, n_hand=(4, 4))
    images = synth.make_images(2, c["frames"], 2, c["image"], seed=11).to(torch.bfloat16)
    fh, fv = synth.make_hands(2, seed=5)
    eps = torch.randn(16, 256, generator=torch.Generator().manual_seed(0))

    def grads(ids_):
        out = m(input_ids=ids_, attention_mask=mask, labels=labels, image=images, future_hands=fh, contact_point=torch.zeros(2, 2),
                future_valid=fv.clone(), gt_label_valid=torch.ones(2), cvae_eps=eps)
        out.loss.backward()
        return m.proj_w_g.clone(), m.proj_b_g.clone()

    # sample 1 loses its <image> token (text-only): its NV rows of d(visual) must be zero, so the projector gradient is
    # finite and equals the one obtained when those rows are explicitly zeroed (oracle autograd on the same batch is the
    # GPU-side check, tests/test_gpu_model.py::test_mixed_text_only_batch_backward_vs_oracle)
    ids1 = ids.clone()
    ids1[1][ids1[1] == -200] = 7
    real_zeros = torch.zeros
    monkeypatch.setattr(torch, "empty", lambda *a, **k: real_zeros(*a, **k).fill_(float("nan")) if k.get("dtype", torch.float32).is_floating_point else real_zeros(*a, **k))
    gw, gb = grads(ids1)
    monkeypatch.undo()
    assert torch.isfinite(gw.float()).all() and torch.isfinite(gb.float()).all() and gw.float().abs().sum() > 0


def test_checkpoint_marker_is_written_last_and_fp8_copies_follow_the_weights(tmp_path, monkeypatch):
    """(i) a checkpoint directory becomes visible to latest_checkpoint() only through mark_complete(), which rank 0 calls after
    every rank has written its files (hvlm/train.py::save); a missing per-rank RNG file is reported on resume;
    (ii) any in-place rewrite of the bf16 weights by the optimizer (step / load_group / reload_from_model) drops the e4m3 copies
    an fp8 evaluation made from them, so the next fp8 call re-quantises the NEW weights."""
    import warnings
    import cpu_ops
    cpu_ops.install(monkeypatch)
    from hvlm import checkpoint
    from hvlm.optim import AdamW
    from hvlm.train import latest_checkpoint
    m, _ = _tiny()
    opt = AdamW(m, lr=1e-3)
    path = tmp_path / "out" / "checkpoint-3"
    checkpoint.save_pretrained(m, str(path))
    checkpoint.save_training_state(opt, str(path), 3, 0, rank=0, write_marker=False)
    assert latest_checkpoint(str(tmp_path / "out")) is None                 # optimizer shards on disk, marker not yet
    checkpoint.mark_complete(str(path), 3, 0)
    assert latest_checkpoint(str(tmp_path / "out")) == str(path)
    assert not os.path.exists(path / (checkpoint.TRAINER_STATE + ".tmp"))
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter("always")
        checkpoint.load_training_state(opt, str(path), rank=1)               # rank 1 never wrote an RNG file
    assert any("rng_state_1.pth" in str(x.message) for x in w)
    # (ii)
    m.stack._fp8_rw = ["stale"]
    opt.load_group("layers.0", torch.load(path / "optimizer" / "layers.0.pt", weights_only=True))
    assert m.stack._fp8_rw is None
    m.stack._fp8_rw = ["stale"]
    opt.reload_from_model()
    assert m.stack._fp8_rw is None
    m.stack._fp8_rw = ["stale"]
    opt.step()
    assert m.stack._fp8_rw is None
