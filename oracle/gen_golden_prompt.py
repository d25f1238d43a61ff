#!/usr/bin/env python3
"""Generates tests/golden/g8_prompt.npz by running the REFERENCE's prompt plumbing (llava/mm_utils.py
tokenizer_image_token, lita/dataset/base_dataset.py preprocess_v1 with the `v1` conversation template) on a
deterministic stand-in tokenizer (the Vicuna SentencePiece model is not available offline).  Build container only."""
import os, sys, types
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import shim
from oracle.synth import StandInTokenizer, PROMPT_SOURCES

shim.install()
decord = types.ModuleType("decord"); decord.VideoReader = None; decord.cpu = None
sys.modules.setdefault("decord", decord)
from llava import conversation as conversation_lib          # noqa: E402
from llava.mm_utils import tokenizer_image_token             # noqa: E402
from lita.dataset import base_dataset                        # noqa: E402

conversation_lib.default_conversation = conversation_lib.conv_templates["v1"]
tok = StandInTokenizer()
out = {}
for n, (sources, has_image) in enumerate(PROMPT_SOURCES):
    r = base_dataset.preprocess_v1(sources, tok, has_image=has_image)
    out[f"ids{n}"] = r["input_ids"].numpy(); out[f"labels{n}"] = r["labels"].numpy()
prompt = "A <image> b <image><image> c"
out["tit"] = np.array(tokenizer_image_token(prompt, tok), dtype=np.int64)
np.savez_compressed(os.path.join(ROOT, "tests", "golden", "g8_prompt.npz"), **out)
print({k: v.shape for k, v in out.items()})
