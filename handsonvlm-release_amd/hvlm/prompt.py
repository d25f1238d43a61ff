"""Prompt / token plumbing in front of the hot path: the Vicuna-v1 conversation template, the
`<image>` placeholder tokenisation and the label masking of supervised samples.

Host-side Python only; works with any tokenizer object that offers `tokenizer(text).input_ids`, `bos_token_id`,
`pad_token_id` and `model_max_length` (the Vicuna SentencePiece model is not available offline: the tests pin the
behaviour against the reference functions with a deterministic stand-in tokenizer, tests/golden/g8_prompt.npz).

Reference behaviour reproduced (not copied):
  * `tokenizer_image_token`  llava/mm_utils.py:20-75       — every chunk between `<image>` markers is tokenised on its
    own (each gets a BOS), the chunks are joined by the image index (-200), one BOS is kept;
  * conversation template     llava/conversation.py:51-60,253-263 (`conv_vicuna_v1`, SeparatorStyle.TWO);
  * `preprocess_v1`           lita/dataset/base_dataset.py:183-255 — labels = ids with the system prompt, every
    instruction ("... ASSISTANT: ") and the padding set to -100; the instruction length is "tokens of the
    instruction minus 2" (the reference's SentencePiece off-by-two, kept bit-for-bit), and a sample whose round
    lengths do not add up to its token count is ignored entirely (all -100).
"""
from __future__ import annotations

from typing import Dict, List, Sequence

import torch

IGNORE_INDEX = -100
IMAGE_TOKEN_INDEX = -200
IMAGE_TOKEN = "<image>"
HAND_TOKEN = "<hand_traj>"

SYSTEM_V1 = ("A chat between a curious user and an artificial intelligence assistant. "
             "The assistant gives helpful, detailed, and polite answers to the user's questions.")
ROLES_V1 = ("USER", "ASSISTANT")
SEP_V1, SEP2_V1 = " ", "</s>"


def conversation_v1(messages: Sequence[Sequence[str]]) -> str:
    """[(role, text or None)] -> prompt string of the `v1` template (two separators: " " after USER turns and
    "</s>" after ASSISTANT turns; an empty last message leaves "ROLE:" open for generation)."""
    out = SYSTEM_V1 + SEP_V1
    for i, (role, text) in enumerate(messages):
        out += f"{role}: {text}{(SEP_V1, SEP2_V1)[i % 2]}" if text else f"{role}:"
    return out


def tokenizer_image_token(prompt: str, tokenizer, image_token_index: int = IMAGE_TOKEN_INDEX, return_tensors=None):
    chunks = [tokenizer(part).input_ids for part in prompt.split(IMAGE_TOKEN)]
    has_bos = bool(chunks) and bool(chunks[0]) and chunks[0][0] == tokenizer.bos_token_id
    skip = 1 if has_bos else 0
    ids: List[int] = [chunks[0][0]] if has_bos else []
    for n, chunk in enumerate(chunks):
        if n:
            ids.append(image_token_index)
        ids.extend(chunk[skip:])
    if return_tensors is None:
        return ids
    if return_tensors != "pt":
        raise ValueError(f"Unsupported tensor type: {return_tensors}")
    return torch.tensor(ids, dtype=torch.long)


def preprocess_v1(sources: Sequence[Sequence[Dict[str, str]]], tokenizer, has_image: bool = False) -> Dict[str, torch.Tensor]:
    """sources: per sample a list of {"from": "human" | "gpt", "value": text}.  Returns input_ids / labels [B, T]."""
    roles = {"human": ROLES_V1[0], "gpt": ROLES_V1[1]}
    prompts = []
    for turns in sources:
        if roles[turns[0]["from"]] != ROLES_V1[0]:
            turns = turns[1:]                                   # a sample must open with the human turn
        msgs = []
        for j, t in enumerate(turns):
            assert roles[t["from"]] == ROLES_V1[j % 2], "turns must alternate human / gpt"
            msgs.append((roles[t["from"]], t["value"]))
        prompts.append(conversation_v1(msgs))

    if has_image:
        input_ids = torch.stack([tokenizer_image_token(p, tokenizer, return_tensors="pt") for p in prompts], dim=0)
    else:
        input_ids = tokenizer(prompts, return_tensors="pt", padding="longest", max_length=tokenizer.model_max_length,
                              truncation=True).input_ids
    labels = input_ids.clone()

    def ntok(text: str) -> int:
        return len(tokenizer_image_token(text, tokenizer)) if has_image else len(tokenizer(text).input_ids)

    sep = SEP_V1 + ROLES_V1[1] + ": "
    for prompt, lab in zip(prompts, labels):
        total = int(lab.ne(tokenizer.pad_token_id).sum())
        cur = 1
        lab[:cur] = IGNORE_INDEX                                  # BOS
        for rnd in prompt.split(SEP2_V1):
            if rnd == "":
                break
            parts = rnd.split(sep)
            if len(parts) != 2:
                break
            n_instr = ntok(parts[0] + sep) - 2
            lab[cur:cur + n_instr] = IGNORE_INDEX
            cur += ntok(rnd)
        lab[cur:] = IGNORE_INDEX
        if cur < tokenizer.model_max_length and cur != total:
            lab[:] = IGNORE_INDEX                                 # tokenisation mismatch: the sample is ignored
    return dict(input_ids=input_ids, labels=labels)
