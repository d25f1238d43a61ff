"""Deterministic synthetic weights and batches shared by oracle/gen_golden.py, tests/ and bench.py.

TEST INFRASTRUCTURE (see oracle/ref_cpu.py header).  Weights come from numpy's legacy RandomState
(bit-stable across numpy versions), so the reference run that produced tests/golden/ and every later
oracle / HIP run see identical parameters without shipping multi-megabyte tensors.
"""
from __future__ import annotations

import numpy as np
import torch

IMAGE_TOKEN_INDEX = -200
HAND_TOKEN_ID = 32100
VOCAB = 32101

TINY = dict(D=128, H=2, L=2, FFN=256, clip_d=64, clip_heads=1, clip_layers=3, clip_mlp=128,
            image=224, patch=14, frames=100, vocab=VOCAB)
# the same family at the PRODUCTION head size (dh = D / H = 128, the causal dh-128 attention path of Vicuna-7B/13B): the
# end-to-end fixtures g3_forward_dh128 / g3_bf16_dh128 / g6_sample_dh128 are written from the reference on this config
MID = dict(TINY, D=256, FFN=512)


def _rs(seed):
    return np.random.RandomState(seed)


def llama_params(cfg: dict, seed: int = 0, std: float = 0.05) -> dict:
    """HF-named LLM + heads parameters (state-dict keys of HandsOnVLMForCausalLM)."""
    r = _rs(seed)
    D, FFN, L, V = cfg["D"], cfg["FFN"], cfg["L"], cfg["vocab"]
    p = {}

    def w(*shape, s=std):
        return (r.standard_normal(shape) * s).astype(np.float32)

    p["model.embed_tokens.weight"] = w(V, D, s=0.5)
    for i in range(L):
        pre = f"model.layers.{i}."
        for n in ("q_proj", "k_proj", "v_proj", "o_proj"):
            p[pre + f"self_attn.{n}.weight"] = w(D, D, s=D ** -0.5)
        p[pre + "mlp.gate_proj.weight"] = w(FFN, D, s=D ** -0.5)
        p[pre + "mlp.up_proj.weight"] = w(FFN, D, s=D ** -0.5)
        p[pre + "mlp.down_proj.weight"] = w(D, FFN, s=FFN ** -0.5)
        p[pre + "input_layernorm.weight"] = (1.0 + 0.1 * r.standard_normal(D)).astype(np.float32)
        p[pre + "post_attention_layernorm.weight"] = (1.0 + 0.1 * r.standard_normal(D)).astype(np.float32)
    p["model.norm.weight"] = (1.0 + 0.1 * r.standard_normal(D)).astype(np.float32)
    p["lm_head.weight"] = w(V, D, s=D ** -0.5)
    p["model.mm_projector.weight"] = w(D, cfg["clip_d"], s=cfg["clip_d"] ** -0.5)
    p["model.mm_projector.bias"] = w(D, s=0.1)
    p.update(cvae_params(D // 2, seed + 1))
    return p


def cvae_params(cond_dim: int, seed: int = 1, hidden: int = 512, latent: int = 256) -> dict:
    r = _rs(seed)
    pre = "hand_traj_decoder.hand_traj_decoder.cvae."

    def lin(o, i):
        return ((r.standard_normal((o, i)) * i ** -0.5).astype(np.float32), (r.standard_normal(o) * 0.05).astype(np.float32))

    p = {}
    for name, (o, i) in {"enc_MLP.0": (hidden, 2 + cond_dim), "linear_means": (latent, hidden),
                         "linear_log_var": (latent, hidden), "dec_MLP.0": (hidden, latent + cond_dim),
                         "dec_MLP.2": (2, hidden)}.items():
        wgt, b = lin(o, i)
        if name == "linear_log_var":
            wgt *= 0.2
        p[pre + name + ".weight"], p[pre + name + ".bias"] = wgt, b
    return p


def clip_params(cfg: dict, seed: int = 7, prefix: str = "vision_model.") -> dict:
    """HF-named CLIPVisionModel parameters (keys relative to the CLIPVisionModel module)."""
    r = _rs(seed)
    d, mlp, L, P = cfg["clip_d"], cfg["clip_mlp"], cfg["clip_layers"], cfg["patch"]
    npos = (cfg["image"] // P) ** 2 + 1
    p = {}

    def w(*shape, s):
        return (r.standard_normal(shape) * s).astype(np.float32)

    p[prefix + "embeddings.class_embedding"] = w(d, s=0.5)
    p[prefix + "embeddings.patch_embedding.weight"] = w(d, 3, P, P, s=(3 * P * P) ** -0.5)
    p[prefix + "embeddings.position_embedding.weight"] = w(npos, d, s=0.3)
    for n in ("pre_layrnorm", "post_layernorm"):
        p[prefix + n + ".weight"] = (1.0 + 0.1 * r.standard_normal(d)).astype(np.float32)
        p[prefix + n + ".bias"] = w(d, s=0.05)
    for i in range(L):
        pre = f"{prefix}encoder.layers.{i}."
        for n in ("q_proj", "k_proj", "v_proj", "out_proj"):
            p[pre + f"self_attn.{n}.weight"] = w(d, d, s=d ** -0.5)
            p[pre + f"self_attn.{n}.bias"] = w(d, s=0.05)
        for n in ("layer_norm1", "layer_norm2"):
            p[pre + n + ".weight"] = (1.0 + 0.1 * r.standard_normal(d)).astype(np.float32)
            p[pre + n + ".bias"] = w(d, s=0.05)
        p[pre + "mlp.fc1.weight"] = w(mlp, d, s=d ** -0.5)
        p[pre + "mlp.fc1.bias"] = w(mlp, s=0.05)
        p[pre + "mlp.fc2.weight"] = w(d, mlp, s=mlp ** -0.5)
        p[pre + "mlp.fc2.bias"] = w(d, s=0.05)
    return p


def to_torch(p: dict, dtype=torch.float32) -> dict:
    return {k: torch.from_numpy(v).to(dtype) for k, v in p.items()}


def make_images(B: int, frames: int, unique: int, size: int, seed: int = 11) -> torch.Tensor:
    """[B, frames, 3, size, size]: `unique` random frames, each repeated frames/unique times consecutively.
    (The golden fixtures were generated with this pattern; EK100's own pattern is the sequence repeated,
    `arange(frames) % unique`, which bench.py uses.)"""
    r = _rs(seed)
    base = r.standard_normal((B, unique, 3, size, size)).astype(np.float32)
    idx = (np.arange(frames) * unique) // frames
    return torch.from_numpy(base[:, idx])


def make_text_batch(B: int, T: int, seed: int = 3, n_hand=(4,), image_pos: int = 5, pad_tail=(0,),
                    label_from: float = 0.4):
    """input_ids/labels/attention_mask [B,T] with one <image> (-200) at `image_pos` and n_hand[b]
    <hand_traj> tokens (id 32100) spread over the answer part; pad_tail[b] trailing pad tokens (id 0)."""
    r = _rs(seed)
    ids = r.randint(3, 32000, size=(B, T)).astype(np.int64)
    labels = ids.copy()
    mask = np.ones((B, T), dtype=bool)
    for b in range(B):
        nh = n_hand[b % len(n_hand)]
        pt = pad_tail[b % len(pad_tail)]
        end = T - pt
        ids[b, image_pos] = IMAGE_TOKEN_INDEX
        for j in range(nh):
            ids[b, end - 2 - 2 * (nh - 1 - j)] = HAND_TOKEN_ID
        labels[b] = ids[b]
        labels[b, : int(label_from * T)] = -100
        labels[b, image_pos] = -100
        if pt:
            ids[b, end:] = 0
            labels[b, end:] = -100
            mask[b, end:] = False
    return torch.from_numpy(ids), torch.from_numpy(labels), torch.from_numpy(mask)


def make_hands(B: int, seed: int = 5):
    r = _rs(seed)
    fh = r.uniform(0.05, 0.95, size=(B, 2, 5, 2)).astype(np.float32)
    fv = np.ones((B, 2), dtype=np.float32)
    return torch.from_numpy(fh), torch.from_numpy(fv)


# ---- stand-in tokenizer + conversations for the prompt-plumbing fixtures (g8) --------------------------------------
class StandInTokenizer:
    """Deterministic character-level tokenizer with the three properties the reference's masking arithmetic relies on:
    a BOS in front of every call, "</s>" and "<hand_traj>" as single tokens, right padding with id 0."""
    bos_token_id, eos_token_id, pad_token_id, model_max_length = 1, 2, 0, 2048

    def _ids(self, text):
        ids, i = [self.bos_token_id], 0
        while i < len(text):
            if text.startswith("</s>", i):
                ids.append(self.eos_token_id); i += 4
            elif text.startswith("<hand_traj>", i):
                ids.append(32100); i += 11
            else:
                ids.append(3 + ord(text[i]) % 997); i += 1
        return ids

    def __call__(self, text, return_tensors=None, padding=None, max_length=None, truncation=None, add_special_tokens=True):
        import types as _t
        import torch as _torch
        if isinstance(text, str):
            return _t.SimpleNamespace(input_ids=self._ids(text))
        rows = [self._ids(t)[:max_length] for t in text]
        n = max(len(r) for r in rows)
        return _t.SimpleNamespace(input_ids=_torch.tensor([r + [self.pad_token_id] * (n - len(r)) for r in rows], dtype=_torch.long))


PROMPT_SOURCES = [
    ([[{"from": "human", "value": "<image>\nWhere should my hand move to if I want to open the drawer?"},
       {"from": "gpt", "value": "Sure! Here is the hand trajectory <hand_traj><hand_traj><hand_traj><hand_traj>."}]], True),
    ([[{"from": "human", "value": "<image>\nWhat is happening?"}, {"from": "gpt", "value": "A person cuts an onion."},
       {"from": "human", "value": "And then?"}, {"from": "gpt", "value": "They put it in the pan <hand_traj>."}]], True),
    ([[{"from": "human", "value": "Describe a kitchen."}, {"from": "gpt", "value": "It has a sink."}],
      [{"from": "human", "value": "Hi"}, {"from": "gpt", "value": "Hello there, how can I help you today?"}]], False),
]


def collator_instances():
    """Seeded inputs for the batch-contract fixture (oracle/gen_golden_collator.py <-> tests/test_next_rows_cpu.py): a list
    of cases, each a list of per-sample dicts as the reference's datasets produce them."""
    r = _rs(901)

    def sample(T, full, single_image=False, bool_flag=False, dtype_hands=torch.float32):
        ids = torch.from_numpy(r.randint(3, 32000, size=T).astype(np.int64))
        labels = ids.clone()
        labels[: T // 3] = -100
        img = torch.from_numpy(r.standard_normal((3, 224, 224) if single_image else (100, 3, 224, 224)).astype(np.float32)).to(torch.bfloat16)
        d = dict(input_ids=ids, labels=labels, image=img)
        if full:
            d.update(feat=torch.from_numpy(r.standard_normal((5, 10, 1024)).astype(np.float32)).to(torch.bfloat16),
                     bbox_feat=torch.from_numpy(r.standard_normal((4, 10, 4)).astype(np.float32)).to(torch.bfloat16),
                     valid_mask=torch.from_numpy(r.rand(5, 10) > 0.5),
                     future_hands=torch.from_numpy(r.rand(2, 5, 2).astype(np.float32)).to(dtype_hands),
                     contact_point=torch.from_numpy(r.rand(2).astype(np.float32)).to(dtype_hands),
                     future_valid=torch.from_numpy(r.rand(2) > 0.3),
                     gt_label_valid=True if bool_flag else torch.tensor(True), prompt="what will the hands do?")
        return d

    return [
        [sample(12, full=False, single_image=True), sample(31, full=True, bool_flag=True)],          # zero-fill + tiling + ragged
        [sample(55, full=True, dtype_hands=torch.bfloat16), sample(20, full=True, dtype_hands=torch.bfloat16),
         sample(47, full=False)],                                                                     # truncation at 40
    ]
