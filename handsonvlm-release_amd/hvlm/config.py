"""Configuration of the HandsOnVLM hot path.  Field names follow the reference's config object
(HandsOnVLMConfig(LlamaConfig) + the fields initialize_constants / initialize_vision_modules copy
onto it: handsonvlm/model/handsonvlm_arch.py:34-45, llava/model/llava_arch.py:41-62) so that the
flags of scripts/finetune.sh map 1:1."""
from __future__ import annotations

from dataclasses import dataclass, field

IGNORE_INDEX = -100        # handsonvlm/constants.py
IMAGE_TOKEN_INDEX = -200   # handsonvlm/constants.py
HAND_TOKEN_ID = 32100      # hard-coded at handsonvlm.py:146,349,609


@dataclass
class CLIPVisionConfig:
    hidden_size: int = 1024
    intermediate_size: int = 4096
    num_hidden_layers: int = 24
    num_attention_heads: int = 16
    image_size: int = 224
    patch_size: int = 14
    layer_norm_eps: float = 1e-5

    @property
    def num_patches(self) -> int:
        return (self.image_size // self.patch_size) ** 2


@dataclass
class HandsOnVLMConfig:
    # LlamaConfig fields (Vicuna-7B v1.3 defaults)
    hidden_size: int = 4096
    intermediate_size: int = 11008
    num_hidden_layers: int = 32
    num_attention_heads: int = 32
    vocab_size: int = 32101            # 32000 + 100 time tokens + <hand_traj>
    rms_norm_eps: float = 1e-6
    rope_theta: float = 10000.0
    max_position_embeddings: int = 2048
    pad_token_id: int = 0
    bos_token_id: int = 1
    eos_token_id: int = 2
    # multimodal fields
    mm_vision_tower: str = "openai/clip-vit-large-patch14"
    mm_hidden_size: int = 1024
    mm_vision_select_layer: int = -2
    mm_vision_select_feature: str = "patch"
    fuse_input_mode: str = "origin"
    video_compress_mode: str = "temporal_spatial_pool"
    num_frames: int = 100
    # loss weights (handsonvlm/arguments.py:104-109)
    lambda_traj: float = 1.0
    lambda_traj_kl: float = 1e-3
    hoi_lambda: float = 1.0
    traj_decoder_name: str = "CVAE"
    # build-specific knobs (no reference counterpart)
    project_after_pool: bool = True    # mm_projector commutes with the mean/avg pooling: 356 rows instead of 25 600
    pool_before_last_fc2: bool = True  # ... and so do CLIPMLP.fc2 + the residual add of the LAST tower layer that is run (nothing but the
                                       # pooling consumes that layer's output): its QuickGELU output and its residual stream are pooled
                                       # first and fc2 runs on 356 rows per sample instead of 25 700 (1/23 of the tower's fc2 work).  The
                                       # rounding order changes like with project_after_pool (one fp16 rounding of the pooled operands
                                       # instead of 256 roundings that are then averaged); applies to the fp16 tower with
                                       # project_after_pool, without frame de-duplication; False = fc2 on every token, as the reference
    dedup_frames: bool = False         # run CLIP once per unique frame (EK100 tiles 10 frames x10)
    lm_head_labelled_rows: bool = False  # training passes: lm_head + cross-entropy only on rows whose shifted label is not IGNORE_INDEX
                                         # (visual-token, prompt and padding rows have zero loss and zero gradient: handsonvlm.py:132-145
                                         # computes their logits and CrossEntropyLoss(ignore_index=-100) drops them) — same loss and
                                         # gradients; reading `out.logits` in such a pass raises unless keep_logits=True (model.LabelledRowsOutput).  Without
                                         # output_hidden_states the LAST decoder layer's o_proj and MLP run on those rows only, too
                                         # (every row still feeds every layer's K / V; LlamaStack.forward(last_rows=...))
    clip_chunk_frames: int = 1600      # frames per CLIP pass (bounds activation memory: ~11 GB of fp16 activations at 224 px).  One
                                       # pass for the 16 x 100 frames of the SFT batch: 6428 / 19284 / 25712 tiles per GEMM = 25-100 rounds
                                       # of 256 CUs with a 3-4 % tail, against 6.28 rounds (10 % tail) for the N = 1024 GEMMs of a 400-frame
                                       # pass — measured 744-745 vs 753-754 ms per step
    vision_dtype: str = "fp16"         # the CLIP tower in the reference's dtype (train.py:316, builder.py:108); "bf16" for A/B
    activation_recompute: bool = False  # keep only each decoder layer's input, re-run its forward in backward
                                        # (scripts/finetune.sh:36 `--gradient_checkpointing True`); off: 288 GB hold them
    pack_sequences: bool = False       # training batches without padding rows (cu_seqlens through the whole decoder: the
                                        # reference's flash-attn patch unpads around attention only)
    vision: CLIPVisionConfig = field(default_factory=CLIPVisionConfig)

    @property
    def head_dim(self) -> int:
        return self.hidden_size // self.num_attention_heads

    @property
    def vocab_padded(self) -> int:
        return (self.vocab_size + 63) // 64 * 64

    @classmethod
    def vicuna_7b(cls, **kw):
        return cls(**kw)

    @classmethod
    def vicuna_13b(cls, **kw):
        return cls(hidden_size=5120, intermediate_size=13824, num_hidden_layers=40, num_attention_heads=40, **kw)
