#!/usr/bin/env python3
"""Generates tests/golden/ckpt_tiny/: a checkpoint WRITTEN FROM THE REFERENCE MODEL (run only in the build container,
through oracle/shim.py), for SURVEY.md §8f-1 "load HF checkpoints produced by the reference".

    PYTHONDONTWRITEBYTECODE=1 python oracle/gen_golden_ckpt.py

What comes from the reference: the module tree and therefore every state-dict KEY NAME and tensor shape
(HandsOnVLMForCausalLM + CLIPVisionTower registered as a sub-module, as in a real LLaVA-style checkpoint), the
`config.json` written by the reference's own config class (`model.config.save_pretrained`), the expected logits / loss of
that model on a fixed batch (fp32, CPU), and the `mm_projector.bin` adapter in the layout llava_arch.py:64-69 reads.

On-disk weight format: the reference pins transformers==4.31.0, whose `save_pretrained` writes `pytorch_model-0000i-of-
0000N.bin` shards (torch.save of a state-dict slice) + `pytorch_model.bin.index.json` (handsonvlm/model/builder.py:93,158
loads them).  The container has transformers 5.15, which can only write safetensors, so the .bin shards are written here
with torch.save in exactly that 4.31 layout from the reference model's own state_dict(), and the 5.15 `save_pretrained`
output (model.safetensors) of the same model is kept too — hvlm.checkpoint reads both.

Size: embed_tokens / lm_head have 32101 rows (token id 32100 is hard-coded by the reference); to keep the fixture small
their rows repeat a 251- / 241-row codebook and every weight file is xz-compressed (the test inflates them into a tmp dir).
Fixtures are data: tensors, json, expected outputs."""
import lzma
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import shim  # noqa: E402

shim.install()

import numpy as np  # noqa: E402
import torch  # noqa: E402

from oracle import synth  # noqa: E402
from oracle.gen_golden import build_reference_model  # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden", "ckpt_tiny")


def main():
    cfg = synth.TINY
    model = build_reference_model(cfg)
    model.eval()
    with torch.no_grad():
        e = model.model.embed_tokens.weight
        e.copy_(e[:251].repeat(128, 1)[: e.shape[0]].clone())
        h = model.lm_head.weight
        h.copy_(h[:241].repeat(134, 1)[: h.shape[0]].clone())
        for name, prm in model.named_parameters():          # reference modules off the hot path (HOI feature fusion):
            if name.startswith(("feat_fusion", "downproject")) and prm.dim() == 2:   # carried through untouched; low entropy
                prm.copy_(prm[:8].repeat(prm.shape[0] // 8 + 1, 1)[: prm.shape[0]].clone())
    B, T = 2, 24
    ids, labels, mask = synth.make_text_batch(B, T, seed=3, n_hand=(4, 0), pad_tail=(0, 5))
    images = synth.make_images(B, cfg["frames"], 10, cfg["image"], seed=11)
    fh, fv = synth.make_hands(B, seed=5)
    torch.manual_seed(123)
    eps_noise = torch.randn(B * 8, 256)
    torch.manual_seed(123)
    with torch.no_grad():
        out = model(input_ids=ids, attention_mask=mask, labels=labels, image=images, future_hands=fh.clone(),
                    contact_point=torch.zeros(B, 2), future_valid=fv.clone(), gt_label_valid=torch.ones(B),
                    output_hidden_states=True, return_dict=True)
    # register the tower as a real sub-module (it was held in a list for the forward: see gen_golden.build_reference_model)
    tower = model.model.vision_tower[0]
    model.model.vision_tower = tower
    sd = {k: v.detach().clone() for k, v in model.state_dict().items()}
    assert any(k.startswith("model.vision_tower.vision_tower.") for k in sd)
    # transformers 5.x dropped the `vision_model.` level from CLIPVisionModel's keys; a checkpoint written under the
    # reference's pin (4.31) has it (model.vision_tower.vision_tower.vision_model.embeddings...).  The .bin shards below
    # use the 4.31 names, the safetensors file keeps what 5.15 writes; hvlm.checkpoint accepts both.
    VT = "model.vision_tower.vision_tower."
    sd = {(VT + "vision_model." + k[len(VT):] if k.startswith(VT) and not k.startswith(VT + "vision_model.") else k): v
          for k, v in sd.items()}
    if os.path.isdir(OUT):
        shutil.rmtree(OUT)
    os.makedirs(OUT)
    model.config.save_pretrained(OUT)                                     # the reference config class writes config.json
    # transformers-4.31 layout: 3 shards + index
    keys = list(sd)
    n = 3
    per = (len(keys) + n - 1) // n
    wmap = {}
    for i in range(n):
        name = f"pytorch_model-{i + 1:05d}-of-{n:05d}.bin"
        part = {k: sd[k] for k in keys[i * per:(i + 1) * per]}
        torch.save(part, os.path.join(OUT, name))
        wmap.update({k: name for k in part})
    json.dump({"metadata": {"total_size": int(sum(v.numel() * v.element_size() for v in sd.values()))}, "weight_map": wmap},
              open(os.path.join(OUT, "pytorch_model.bin.index.json"), "w"), indent=1)
    # what THIS container's transformers writes for the same model (safetensors)
    st_dir = os.path.join(OUT, "st")
    hidden_stubs = {k: sys.modules.pop(k) for k in list(sys.modules) if k.split(".")[0] == "deepspeed"}   # 5.15 probes its spec
    try:
        model.save_pretrained(st_dir)
    except Exception as ex:  # noqa
        print("transformers", __import__("transformers").__version__, "save_pretrained failed:", repr(ex)[:300])
        shutil.rmtree(st_dir, ignore_errors=True)
    finally:
        sys.modules.update(hidden_stubs)
    # LLaVA adapter file (llava_arch.py:64-69 keeps every key containing 'mm_projector' and strips up to 'mm_projector.');
    # DIFFERENT values from the checkpoint's projector so that the override is observable
    g = torch.Generator().manual_seed(77)
    adapter = {"model.mm_projector.weight": torch.randn(cfg["D"], cfg["clip_d"], generator=g) * cfg["clip_d"] ** -0.5,
               "model.mm_projector.bias": torch.randn(cfg["D"], generator=g) * 0.1}
    torch.save(adapter, os.path.join(OUT, "mm_projector.bin"))
    with torch.no_grad():
        model.model.mm_projector.load_state_dict({k.split("mm_projector.")[1]: v for k, v in adapter.items()})
        del model.model._modules["vision_tower"]
        model.model.vision_tower = [tower]
        out2 = model(input_ids=ids, attention_mask=mask, image=images, return_dict=True, is_evaluate=True)
    np.savez_compressed(os.path.join(OUT, "expected.npz"), logits_slice=out.logits[:, ::7, ::251].numpy(),
                        hidden=out.hidden_states[-1].numpy(), loss=out.loss.numpy(), eps_noise=eps_noise.numpy(),
                        logits_slice_adapter=out2.logits[:, ::7, ::251].numpy(),
                        key_names=np.array(sorted(sd)), key_shapes=np.array([str(tuple(sd[k].shape)) for k in sorted(sd)]))
    total = 0
    for d, _, files in os.walk(OUT):
        for f in files:
            p = os.path.join(d, f)
            if f.endswith((".bin", ".safetensors")):
                with open(p, "rb") as fi, lzma.open(p + ".xz", "wb", preset=9) as fo:
                    shutil.copyfileobj(fi, fo)
                os.remove(p)
                p += ".xz"
            total += os.path.getsize(p)
    print(f"wrote {OUT}: {total / 1024:.0f} KiB, {len(sd)} tensors")


if __name__ == "__main__":
    main()
