"""a checkpoint written from the REFERENCE model (tests/golden/ckpt_tiny/, made by
oracle/gen_golden_ckpt.py: reference state-dict key names + its config.json + the transformers-4.31 .bin shard layout the
reference reads at handsonvlm/model/builder.py:93,158, and the safetensors file today's transformers writes for the same
model) is ingested by hvlm.checkpoint.from_pretrained, every reference tensor is either consumed by the hot path or carried
through untouched, and the loaded model reproduces the reference's logits / loss (expected.npz, fp32 CPU run of the
reference).  The LLaVA adapter file `mm_projector.bin` (llava/model/llava_arch.py:64-69) overrides the projector.
Here on CPU the kernels are the ops stand-in (tests/cpu_ops.py); tests/test_gpu_next_rows.py runs the same fixture on HIP."""
import lzma
import os
import shutil

import numpy as np
import pytest
import torch

from oracle import synth

FIX = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "ckpt_tiny")


def inflate(dst):
    """Fixture files are xz-compressed (size); inflate them into `dst` with their real names."""
    for d, _, files in os.walk(FIX):
        out = os.path.join(dst, os.path.relpath(d, FIX))
        os.makedirs(out, exist_ok=True)
        for f in files:
            if f.endswith(".xz"):
                with lzma.open(os.path.join(d, f)) as fi, open(os.path.join(out, f[:-3]), "wb") as fo:
                    shutil.copyfileobj(fi, fo)
            else:
                shutil.copy(os.path.join(d, f), out)
    return dst


def batch():
    c = synth.TINY
    ids, labels, mask = synth.make_text_batch(2, 24, seed=3, n_hand=(4, 0), pad_tail=(0, 5))
    images = synth.make_images(2, c["frames"], 10, c["image"], seed=11)
    fh, fv = synth.make_hands(2, seed=5)
    return ids, labels, mask, images, fh, fv


def rel(a, b):
    a = torch.as_tensor(a).float().cpu(); b = torch.as_tensor(b).float()
    return float((a - b).norm() / b.norm())


def check_loaded_model(m, exp, device):
    ids, labels, mask, images, fh, fv = batch()
    S = 24 + 355
    valid = torch.ones(2, S, dtype=torch.bool); valid[1, S - 5:] = False
    out = m(input_ids=ids, attention_mask=mask, labels=labels, image=images.to(torch.bfloat16), future_hands=fh.clone(),
            contact_point=torch.zeros(2, 2), future_valid=fv.clone(), gt_label_valid=torch.ones(2), output_hidden_states=True,
            cvae_eps=torch.from_numpy(exp["eps_noise"]).to(device))
    r_h = rel(out.hidden_states[-1].float().cpu()[valid], exp["hidden"][valid.numpy()])
    r_l = rel(out.logits.float().cpu()[:, ::7, ::251][valid[:, ::7]], exp["logits_slice"][valid[:, ::7].numpy()])
    r_loss = abs(float(out.loss.detach()) - float(exp["loss"])) / float(exp["loss"])
    assert r_h < 1.3e-2 and r_l < 1.5e-2 and r_loss < 5e-3, (r_h, r_l, r_loss)       # bf16 path vs fp32 reference
    return out


def test_reference_checkpoint_ingestion(tmp_path, monkeypatch):
    import cpu_ops
    cpu_ops.install(monkeypatch)
    from hvlm import checkpoint
    d = inflate(str(tmp_path / "ck"))
    exp = dict(np.load(os.path.join(d, "expected.npz")))
    m = checkpoint.from_pretrained(d, device="cpu", training=False)
    # geometry came from the reference's config.json + the tower tensors (the reference names the tower, it does not describe it)
    assert (m.config.hidden_size, m.config.num_hidden_layers, m.config.vocab_size) == (128, 2, 32101)
    assert (m.config.vision.hidden_size, m.config.vision.num_hidden_layers, m.config.vision.intermediate_size) == (64, 3, 128)
    assert m.tower.is_loaded
    # every tensor the reference wrote is consumed by the hot path or carried through untouched for the next save
    consumed = set(m.state_dict()) | {k for k in exp["key_names"] if k.startswith("model.vision_tower.")}
    carried = set(m._extra_state)
    names = set(str(k) for k in exp["key_names"])
    assert names <= consumed | carried, sorted(names - consumed - carried)[:5]
    assert {"feat_fusion.0.weight", "downproject.weight"} <= carried
    check_loaded_model(m, exp, "cpu")
    # save -> load round trip keeps the reference's key set (incl. carried tensors) and values
    checkpoint.save_pretrained(m, str(tmp_path / "resaved"))
    back = checkpoint.load_state(str(tmp_path / "resaved"))
    orig = checkpoint.load_state(d)
    for k in names:                                       # (the frozen tower is re-saved too, complete, in its fp16 dtype)
        assert k in back, k
        assert torch.equal(back[k].float(), orig[k].to(back[k].dtype).float()), k
    m2 = checkpoint.from_pretrained(str(tmp_path / "resaved"), device="cpu", training=False)     # self-contained: loads and reproduces
    check_loaded_model(m2, exp, "cpu")


def test_safetensors_variant_and_adapter_override(tmp_path, monkeypatch):
    import cpu_ops
    cpu_ops.install(monkeypatch)
    from hvlm import checkpoint
    d = inflate(str(tmp_path / "ck"))
    exp = dict(np.load(os.path.join(d, "expected.npz")))
    a, b = checkpoint.load_state(d), checkpoint.load_state(os.path.join(d, "st"))
    norm = lambda k: k.replace("vision_tower.vision_tower.vision_model.", "vision_tower.vision_tower.")  # noqa: E731  (4.31 vs 5.x CLIP keys)
    a = {norm(k): v for k, v in a.items()}
    assert a.keys() == b.keys() and all(torch.equal(a[k], b[k]) for k in a)
    m = checkpoint.from_pretrained(os.path.join(d, "st"), device="cpu", training=False,
                                   mm_projector_bin=os.path.join(d, "mm_projector.bin"))
    ad = torch.load(os.path.join(d, "mm_projector.bin"), map_location="cpu", weights_only=True)
    assert torch.equal(m.proj_w.float(), ad["model.mm_projector.weight"].to(torch.bfloat16).float())
    ids, labels, mask, images, fh, fv = batch()
    out = m(input_ids=ids, attention_mask=mask, image=images.to(torch.bfloat16), is_evaluate=True)
    S = 24 + 355
    valid = torch.ones(2, S, dtype=torch.bool); valid[1, S - 5:] = False
    got = out.logits.float()[:, ::7, ::251][valid[:, ::7]]
    assert rel(got, exp["logits_slice_adapter"][valid[:, ::7].numpy()]) < 1.5e-2
    assert rel(got, exp["logits_slice"][valid[:, ::7].numpy()]) > 5e-2            # the adapter did change the output


def test_load_state_probe_order_is_transformers(tmp_path):
    """A directory holding BOTH a stale pytorch_model.bin and a model.safetensors (re-saved by a newer transformers) loads
    the safetensors file — the order transformers' from_pretrained (called by handsonvlm/model/builder.py:93,158) probes a
    local directory in: model.safetensors, model.safetensors.index.json, pytorch_model.bin, pytorch_model.bin.index.json."""
    import torch
    from safetensors.torch import save_file
    from hvlm import checkpoint
    torch.save({"w": torch.zeros(3)}, tmp_path / "pytorch_model.bin")
    save_file({"w": torch.ones(3)}, str(tmp_path / "model.safetensors"))
    assert torch.equal(checkpoint.load_state(str(tmp_path))["w"], torch.ones(3))
    (tmp_path / "model.safetensors").unlink()
    assert torch.equal(checkpoint.load_state(str(tmp_path))["w"], torch.zeros(3))
    (tmp_path / "pytorch_model.bin").unlink()
    import pytest
    with pytest.raises(FileNotFoundError):
        checkpoint.load_state(str(tmp_path))
