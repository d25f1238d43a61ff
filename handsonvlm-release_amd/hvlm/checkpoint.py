"""Checkpoint I/O compatible with the reference.

Reads the Hugging Face layout the reference produces and consumes (handsonvlm/model/builder.py:93,158 —
`config.json` + `pytorch_model.bin` or sharded `pytorch_model-0000x-of-0000N.bin` with
`pytorch_model.bin.index.json`; `safe_save_model_for_hf_trainer`, handsonvlm/train/train.py:101-136) and the LLaVA
adapter file `mm_projector.bin` (llava/model/llava_arch.py:64-69).  Key names are the reference's;
tensors the hot path does not use (time_embedding, bbox_to_feature, ...) are carried through untouched on save when
they were present on load."""
from __future__ import annotations

import json
import os
from typing import Dict, Optional

import torch

from .config import CLIPVisionConfig, HandsOnVLMConfig

INDEX = "pytorch_model.bin.index.json"
_CFG_MAP = {"hidden_size": "hidden_size", "intermediate_size": "intermediate_size", "num_hidden_layers": "num_hidden_layers",
            "num_attention_heads": "num_attention_heads", "vocab_size": "vocab_size", "rms_norm_eps": "rms_norm_eps",
            "max_position_embeddings": "max_position_embeddings", "mm_hidden_size": "mm_hidden_size",
            "mm_vision_select_layer": "mm_vision_select_layer", "mm_vision_select_feature": "mm_vision_select_feature",
            "fuse_input_mode": "fuse_input_mode", "video_compress_mode": "video_compress_mode",
            "lambda_traj": "lambda_traj", "lambda_traj_kl": "lambda_traj_kl", "hoi_lambda": "hoi_lambda",
            "pad_token_id": "pad_token_id", "bos_token_id": "bos_token_id", "eos_token_id": "eos_token_id"}


def config_from_json(path: str) -> HandsOnVLMConfig:
    raw = json.load(open(os.path.join(path, "config.json")))
    kw = {dst: raw[src] for src, dst in _CFG_MAP.items() if raw.get(src) is not None}
    if raw.get("rope_theta") is not None:                                  # transformers 4.31 (the reference's pin)
        kw["rope_theta"] = float(raw["rope_theta"])
    elif isinstance(raw.get("rope_parameters"), dict) and "rope_theta" in raw["rope_parameters"]:   # transformers 5.x
        kw["rope_theta"] = float(raw["rope_parameters"]["rope_theta"])
    cfg = HandsOnVLMConfig(**kw)
    if "vision_config" in raw:
        cfg.vision = CLIPVisionConfig(**{k: v for k, v in raw["vision_config"].items() if k in CLIPVisionConfig.__dataclass_fields__})
    return cfg


def config_to_json(cfg: HandsOnVLMConfig, path: str):
    raw = {src: getattr(cfg, dst) for src, dst in _CFG_MAP.items()}
    raw.update(model_type="handsonvlm", architectures=["HandsOnVLMForCausalLM"], mm_vision_tower=cfg.mm_vision_tower,
               vision_config={k: getattr(cfg.vision, k) for k in CLIPVisionConfig.__dataclass_fields__})
    json.dump(raw, open(os.path.join(path, "config.json"), "w"), indent=1)


def load_state(path: str) -> Dict[str, torch.Tensor]:
    """Loads a (possibly sharded) HF checkpoint directory into one CPU state dict: the transformers-4.31 layout the reference
    writes and reads (`pytorch_model.bin`, or `pytorch_model-0000i-of-0000N.bin` + `pytorch_model.bin.index.json`;
    handsonvlm/model/builder.py:93,158) and the safetensors layout newer transformers versions write for the same model
    (`model.safetensors`, or shards + `model.safetensors.index.json`)."""
    def shard_files(index_name, single):
        idx = os.path.join(path, index_name)
        if os.path.exists(idx):
            return sorted(set(json.load(open(idx))["weight_map"].values()))
        return [single] if os.path.exists(os.path.join(path, single)) else None

    # probe order of transformers' from_pretrained for a local directory (4.31 — the reference's pin — and later):
    # model.safetensors, model.safetensors.index.json, pytorch_model.bin, pytorch_model.bin.index.json.  A directory re-saved
    # by a newer transformers next to an old .bin must load the tensors the reference's builder.py would load.
    sd: Dict[str, torch.Tensor] = {}
    files = shard_files("model.safetensors.index.json", "model.safetensors")
    if files is not None and os.path.exists(os.path.join(path, "model.safetensors")):
        files = ["model.safetensors"]
    if files is not None:
        from safetensors.torch import load_file
        for f in files:
            sd.update(load_file(os.path.join(path, f), device="cpu"))
        return sd
    files = ["pytorch_model.bin"] if os.path.exists(os.path.join(path, "pytorch_model.bin")) else shard_files(INDEX, "pytorch_model.bin")
    if files is None:
        raise FileNotFoundError(f"{path}: neither model*.safetensors nor pytorch_model*.bin found")
    for f in files:
        sd.update(torch.load(os.path.join(path, f), map_location="cpu", weights_only=True))
    return sd


def save_state(sd: Dict[str, torch.Tensor], path: str, max_shard_bytes: int = 10 * 2 ** 30):
    """Writes HF-style shards + index (gather-16bit-on-save semantics: tensors are saved in their current dtype)."""
    os.makedirs(path, exist_ok=True)
    shards, cur, size = [], {}, 0
    for k, v in sd.items():
        nb = v.numel() * v.element_size()
        if cur and size + nb > max_shard_bytes:
            shards.append(cur); cur, size = {}, 0
        cur[k] = v.detach().cpu().contiguous(); size += nb
    shards.append(cur)
    if len(shards) == 1:
        torch.save(shards[0], os.path.join(path, "pytorch_model.bin"))
        return
    wmap = {}
    for i, sh in enumerate(shards):
        name = f"pytorch_model-{i + 1:05d}-of-{len(shards):05d}.bin"
        torch.save(sh, os.path.join(path, name))
        wmap.update({k: name for k in sh})
    json.dump({"metadata": {"total_size": sum(v.numel() * v.element_size() for v in sd.values())}, "weight_map": wmap},
              open(os.path.join(path, INDEX), "w"), indent=1)


def infer_vision_config(sd: Dict[str, torch.Tensor], prefix: str = "model.vision_tower.vision_tower.vision_model.") -> Optional[CLIPVisionConfig]:
    """The reference's config.json names the tower (`mm_vision_tower`: openai/clip-vit-large-patch14) instead of describing
    it; when the checkpoint carries the tower's tensors (LLaVA-style full checkpoints do) its geometry is read off them."""
    k = prefix + "embeddings.class_embedding"
    if k not in sd:
        prefix = prefix[: -len("vision_model.")]          # key naming of transformers >= 5
        k = prefix + "embeddings.class_embedding"
        if k not in sd:
            return None
    d = sd[k].numel()
    patch = sd[prefix + "embeddings.patch_embedding.weight"].shape[-1]
    npos = sd[prefix + "embeddings.position_embedding.weight"].shape[0]
    layers = 1 + max(int(key[len(prefix + "encoder.layers."):].split(".")[0]) for key in sd if key.startswith(prefix + "encoder.layers."))
    return CLIPVisionConfig(hidden_size=d, intermediate_size=sd[prefix + "encoder.layers.0.mlp.fc1.weight"].shape[0],
                            num_hidden_layers=layers, num_attention_heads=max(1, d // 64), patch_size=patch,
                            image_size=int(round((npos - 1) ** 0.5)) * patch)


def from_pretrained(path: str, device="cuda:0", training: bool = False, mm_projector_bin: Optional[str] = None,
                    vision_tower_path: Optional[str] = None, **cfg_overrides):
    """load_pretrained_model (handsonvlm/model/builder.py:18) for the HIP model — and the training entry's
    `from_pretrained(model_name_or_path)` + `initialize_vision_modules` + `initialize_constants`
    (handsonvlm/train/train.py:235-240, llava/model/llava_arch.py:41-69, handsonvlm/model/handsonvlm_arch.py:17-45):
      path               a HandsOnVLM checkpoint, or a base LLM (Vicuna: 32000-row embeddings, no projector / tower / head
                         tensors) — the vocabulary is then extended to 32101 rows (100 time tokens + <hand_traj>)
      mm_projector_bin   the LLaVA adapter file (`--pretrain_mm_mlp_adapter`), overrides / provides the projector
      vision_tower_path  a local CLIPVisionModel checkpoint directory (`--vision_tower`; the reference names a hub model,
                         there is no network here) for checkpoints that do not carry the tower's tensors"""
    from .config import HAND_TOKEN_ID
    from .model import HandsOnVLMForCausalLM
    cfg = config_from_json(path)
    for k, v in cfg_overrides.items():
        setattr(cfg, k, v)
    cfg.vocab_size = max(cfg.vocab_size, HAND_TOKEN_ID + 1)
    sd = load_state(path)
    vt = "model.vision_tower.vision_tower."
    if vision_tower_path is not None and not any(k.startswith(vt) for k in sd):
        if not os.path.isdir(vision_tower_path):
            raise FileNotFoundError(f"--vision_tower {vision_tower_path!r} is not a local directory (hub names cannot be fetched here): "
                                    "point it at a CLIPVisionModel checkpoint directory")
        tower_sd = load_state(vision_tower_path)
        sd.update({vt + k: v for k, v in tower_sd.items() if "vision_model" in k or k.startswith(("embeddings.", "encoder.", "pre_layrnorm", "post_layernorm"))})
    if "vision" not in cfg_overrides and "vision_config" not in json.load(open(os.path.join(path, "config.json"))):
        vis = infer_vision_config(sd)
        if vis is not None:
            cfg.vision = vis
            cfg.mm_hidden_size = vis.hidden_size
    if mm_projector_bin is not None:                           # initialize_vision_modules(pretrain_mm_mlp_adapter=...)
        adapter = torch.load(mm_projector_bin, map_location="cpu", weights_only=True)
        for k, v in adapter.items():
            if "mm_projector" in k:
                sd["model.mm_projector." + k.split("mm_projector.")[1]] = v
    # every missing-file error BEFORE the model (13 - 26 GB of device memory at 7B / 13B) is constructed
    if "model.mm_projector.weight" not in sd:
        raise FileNotFoundError(f"{path} holds no mm_projector tensors: pass the LLaVA adapter file (mm_projector_bin / "
                                "--pretrain_mm_mlp_adapter, scripts/finetune.sh:12)")
    if not any(k.startswith(vt) for k in sd):
        raise FileNotFoundError(f"{path} holds no CLIP tower tensors: pass a local CLIPVisionModel directory (vision_tower_path / "
                                "--vision_tower, scripts/finetune.sh:11)")
    model = HandsOnVLMForCausalLM(cfg, device=device, training=training)
    model.load_state_dict(sd)
    model._extra_state = {k: v for k, v in sd.items() if k not in model.state_dict()
                          and not k.startswith("model.vision_tower.")}
    return model


def save_pretrained(model, path: str, max_shard_bytes: int = 10 * 2 ** 30):
    os.makedirs(path, exist_ok=True)
    sd = dict(model.state_dict())
    sd.update(getattr(model, "_extra_state", {}))
    if model.tower.is_loaded:
        # the reference saves the full model state dict, frozen tower included (handsonvlm/train/train.py:101-136): a checkpoint
        # written here is as self-contained as one of its own
        sd.update({"model.vision_tower.vision_tower." + k: v for k, v in model.tower.export_state_dict().items()})
    config_to_json(model.config, path)
    save_state(sd, path, max_shard_bytes)


# -- training state (optimizer, schedule position, RNG): what `trainer.train(resume_from_checkpoint=True)` restores in the
#    reference (handsonvlm/train/train.py:399-402; HF Trainer: optimizer.pt, scheduler.pt, trainer_state.json,
#    rng_state_{rank}.pth).  The cosine schedule is a pure function of the step, so the step count is its state. ---------
TRAINER_STATE = "trainer_state.json"


def save_training_state(opt, path: str, global_step: int, epoch: int, rank: int = 0, extra: Optional[dict] = None,
                        write_marker: bool = True):
    """Rank 0 writes the optimizer (fp32 master weights, both moments, step count, the CVAE head's torch optimizer) as
    one file per gradient bucket under `path`/optimizer/ plus trainer_state.json; EVERY rank writes its RNG state.
    trainer_state.json is the marker `latest_checkpoint` trusts: with several ranks pass write_marker=False, barrier once every
    rank has returned, and let rank 0 call mark_complete() — so a crash while another rank is still writing leaves a directory
    that is NOT taken for a checkpoint (hvlm/train.py::save)."""
    os.makedirs(os.path.join(path, "optimizer"), exist_ok=True)
    rng = {"cpu": torch.get_rng_state()}
    if torch.cuda.is_available():
        rng["cuda"] = torch.cuda.get_rng_state()
    torch.save(rng, os.path.join(path, f"rng_state_{rank}.pth"))
    if rank != 0:
        return
    for key, tensors in opt.state_groups():
        torch.save({n: {k: v.detach().cpu() for k, v in d.items()} for n, d in tensors.items()},
                   os.path.join(path, "optimizer", key + ".pt"))
    torch.save(opt.meta_state(), os.path.join(path, "optimizer", "meta.pt"))
    if write_marker:
        mark_complete(path, global_step, epoch, extra)


def mark_complete(path: str, global_step: int, epoch: int, extra: Optional[dict] = None):
    """Writes trainer_state.json atomically (temporary file + rename): its presence means every file of the checkpoint is
    complete."""
    st = {"global_step": int(global_step), "epoch": int(epoch)}
    st.update(extra or {})
    tmp = os.path.join(path, TRAINER_STATE + ".tmp")
    with open(tmp, "w") as f:
        json.dump(st, f, indent=1)
        f.flush()
        os.fsync(f.fileno())
    os.replace(tmp, os.path.join(path, TRAINER_STATE))


def load_training_state(opt, path: str, rank: int = 0) -> dict:
    """Restores what save_training_state wrote; returns trainer_state.json's dict.  Raises when the checkpoint has
    weights but no optimizer state (resuming from it would silently restart Adam) — use opt.reload_from_model() then."""
    odir = os.path.join(path, "optimizer")
    if not os.path.isdir(odir) or not os.path.exists(os.path.join(path, TRAINER_STATE)):
        raise FileNotFoundError(f"{path} holds no optimizer state (weights-only checkpoint): cannot resume training from it")
    for key in opt.groups:
        opt.load_group(key, torch.load(os.path.join(odir, key + ".pt"), map_location="cpu", weights_only=True))
    opt.load_meta_state(torch.load(os.path.join(odir, "meta.pt"), map_location="cpu", weights_only=True))
    rp = os.path.join(path, f"rng_state_{rank}.pth")
    if not os.path.exists(rp):
        import warnings
        warnings.warn(f"{rp} is missing (checkpoint written by fewer ranks?): this rank's RNG stream is NOT restored, the "
                      "continued run will not be bit-identical to an uninterrupted one")
    else:
        rng = torch.load(rp, map_location="cpu", weights_only=True)
        torch.set_rng_state(rng["cpu"])
        if "cuda" in rng and torch.cuda.is_available():
            torch.cuda.set_rng_state(rng["cuda"])
    return json.load(open(os.path.join(path, TRAINER_STATE)))
