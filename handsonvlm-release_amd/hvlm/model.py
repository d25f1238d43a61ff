"""HandsOnVLMForCausalLM on the MI355X kernels — host-side mirror of the reference interface.

Keeps the keyword signatures of handsonvlm/model/language_model/handsonvlm.py:
    forward(input_ids, attention_mask, past_key_values, inputs_embeds, labels, image, future_hands,
            contact_point, future_valid, gt_label_valid, use_cache, output_attentions,
            output_hidden_states, return_dict, **kwargs)  -> CausalLMOutputWithPast-like (:70-210)
    sample(input_ids, logits_processor, stopping_criteria, logits_warper, ..., **model_kwargs)
            -> SampleDecoderOnlyOutput(sequences, pred_hands)                             (:478-672)
    generate(input_ids, image=..., do_sample, temperature, top_p, max_new_tokens, ...)
so the reference's training recipe (`model(**batch).loss.backward()`) and evaluation recipe
(`model.generate(input_ids, image=...)` -> `.sequences`, `.pred_hands`) map onto it unchanged.
Unknown collator keys (prompt, feat, bbox_feat, valid_mask ...) are swallowed by **kwargs.

Everything heavy runs through the C ABI (hvlm.ops); the CVAE head and the loss assembly stay in
PyTorch.  There is no torch fallback for the kernels.
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import Callable, Dict, List, Optional, Tuple

import numpy as np
import torch

from . import ops
from .clip import CLIPVisionTower
from .config import HAND_TOKEN_ID, IGNORE_INDEX, IMAGE_TOKEN_INDEX, HandsOnVLMConfig
from .llama import LlamaStack, _round_up
from .traj_decoder import CVAETrajDecoder, MLPTrajDecoder

BF16 = torch.bfloat16


@dataclass
class CausalLMOutputWithPast:
    loss: Optional[torch.Tensor] = None
    logits: Optional[torch.Tensor] = None
    past_key_values: Optional[object] = None
    hidden_states: Optional[Tuple[torch.Tensor, ...]] = None
    attentions: Optional[object] = None
    loss_components: Optional[Dict[str, torch.Tensor]] = None   # text / traj / kl (the reference logs them to wandb)
    cu_seqlens: Optional[torch.Tensor] = None   # set when logits / hidden_states are PACKED [M, ...] (config.pack_sequences)

    def __getitem__(self, i):
        return tuple(v for v in (self.loss, self.logits, self.past_key_values, self.hidden_states, self.attentions)
                     if v is not None)[i]


class LabelledRowsOutput(CausalLMOutputWithPast):
    """Output of a training pass under config.lm_head_labelled_rows: the [B, S, V] logits of the reference's forward
    (handsonvlm.py:204-210) were never computed — lm_head ran on the rows that carry a target only.  Reading `.logits` raises
    with the names of the switches instead of handing back None; the compact logits are `.logits_rows` [n_v, V] (row i is
    `.rows[i]` of the flattened [B*S] (padded) or [M] (packed) row space), valid until backward() overwrites them in place."""

    def __init__(self, logits_rows=None, rows=None, **kw):
        super().__init__(**kw)
        self.logits_rows, self.rows = logits_rows, rows

    @property
    def logits(self):
        raise RuntimeError("out.logits is not available: this training pass ran lm_head on the labelled rows only "
                           "(config.lm_head_labelled_rows=True).  Use out.logits_rows / out.rows, pass keep_logits=True to "
                           "forward(), or set config.lm_head_labelled_rows=False to get the reference's [B, S, V] logits.")

    @logits.setter
    def logits(self, value):
        assert value is None

    def __getitem__(self, i):
        return tuple(v for v in (self.loss, self.past_key_values, self.hidden_states, self.attentions) if v is not None)[i]


@dataclass
class SampleDecoderOnlyOutput:
    sequences: torch.Tensor = None
    scores: Optional[tuple] = None
    attentions: Optional[tuple] = None
    hidden_states: Optional[tuple] = None
    pred_hands: Optional[torch.Tensor] = None


@dataclass
class SplicePlan:
    """Host-resolved layout of inputs_embeds (handsonvlm.py:212-451)."""
    B: int
    S: int
    src: torch.Tensor          # int32 [B*S]: >=0 token id, -1 pad, <=-2 visual index
    hand_slot: torch.Tensor    # int32 [B*S]: which hand encoding is added on this row, -1 none
    hand_scale: torch.Tensor   # f32 [B]
    labels: Optional[torch.Tensor]    # int64 [B,S] (device) or None
    targets: Optional[torch.Tensor]   # int32 [B*S] shifted labels
    hand_rows: Optional[torch.Tensor]  # int32 [B*4] rows preceding each <hand_traj> label, -1 if none
    has_hand: Optional[np.ndarray]    # bool [B]
    lengths: np.ndarray
    seg: Optional[tuple] = None       # (seg_rows, seg_start, seg_token) int32 device tensors: text rows sorted by token id
    # packed layout (config.pack_sequences, training): src / hand_slot / targets have M = sum(lengths) rows, no padding
    M: int = 0                        # rows of inputs_embeds (B*S when padded)
    cu: Optional[torch.Tensor] = None          # int32 [B+1] cu_seqlens
    pos: Optional[torch.Tensor] = None         # int32 [M] position of every row inside its sample
    row_sample: Optional[torch.Tensor] = None  # int32 [M]
    # rows with a target (shifted label != IGNORE_INDEX), for config.lm_head_labelled_rows
    rows_v: Optional[torch.Tensor] = None      # int64 [n_v] row indices into inputs_embeds
    targets_v: Optional[torch.Tensor] = None   # int32 [n_v] = targets[rows_v]
    hand_rows_v: Optional[torch.Tensor] = None # int32 [B*4]: hand_rows as indices into rows_v (every hand row carries a target)
    n_v: int = 0


class _EngineLoss(torch.autograd.Function):
    """Bridges `loss.backward()` to the explicit engine backward."""

    @staticmethod
    def forward(ctx, anchor, model, value):
        ctx.model = model
        return value.detach().clone()

    @staticmethod
    def backward(ctx, g):
        ctx.model.backward(g)
        return None, None, None


class HandsOnVLMForCausalLM:
    def __init__(self, config: HandsOnVLMConfig, device="cuda:0", training: bool = True, **kwargs):
        self.config = config
        self.device = torch.device(device)
        self.training = training
        self.token_dim = config.hidden_size
        D, V, Vp = config.hidden_size, config.vocab_size, config.vocab_padded
        dev = self.device
        self.traj_decoder_name = kwargs.get("traj_decoder_name", config.traj_decoder_name)
        self.tower = CLIPVisionTower(config.vision, dev, config.mm_vision_select_layer, config.mm_vision_select_feature,
                                     dtype=torch.float16 if config.vision_dtype == "fp16" else BF16)
        self.stack = LlamaStack(config, dev, training)
        self.embed = torch.zeros(V, D, dtype=BF16, device=dev)
        self.lm_head = torch.zeros(Vp, D, dtype=BF16, device=dev)          # rows >= V stay zero
        self.proj_w = torch.zeros(D, config.mm_hidden_size, dtype=BF16, device=dev)
        self.proj_b = torch.zeros(D, dtype=BF16, device=dev)
        if training:
            self.lm_head_t = torch.zeros(D, Vp, dtype=BF16, device=dev)
            self.embed_g = torch.zeros(V, D, dtype=BF16, device=dev)
            self.lm_head_g = torch.zeros(Vp, D, dtype=BF16, device=dev)
            self.proj_w_g = torch.zeros_like(self.proj_w)
            self.proj_b_g = torch.zeros_like(self.proj_b)
        dec = CVAETrajDecoder if self.traj_decoder_name == "CVAE" else MLPTrajDecoder
        self.hand_traj_decoder = dec(token_dim=D // 2).to(dev)              # fp32 host-side head
        self._anchor = torch.zeros(1, device=dev, requires_grad=True)
        self._ctx: Optional[dict] = None
        self.on_grads_ready: Optional[Callable[[str], None]] = None         # DP hook for the non-layer bucket
        self.on_token_ids: Optional[Callable[[torch.Tensor], None]] = None  # DP hook: this step's distinct text-token ids (device int32)

    # ------------------------------------------------------------------------------------------
    # parameters / checkpoints
    # ------------------------------------------------------------------------------------------
    def train(self, mode: bool = True):
        self.hand_traj_decoder.train(mode)
        return self

    def eval(self):
        return self.train(False)

    def get_vision_tower(self):
        return self.tower

    def init_random(self, seed: int = 0, std: float = 0.02):
        gen = torch.Generator(device=self.device).manual_seed(seed)
        rn = lambda *s: (torch.randn(*s, device=self.device, generator=gen) * std).to(BF16)  # noqa: E731
        self.embed.copy_(rn(*self.embed.shape))
        self.lm_head[: self.config.vocab_size].copy_(rn(self.config.vocab_size, self.token_dim))
        self.proj_w.copy_(rn(*self.proj_w.shape))
        self.proj_b.zero_()
        self.stack.init_random(seed + 1, std)
        self.tower.init_random(seed + 2, std)
        self._refresh_transposes()
        return self

    def _refresh_transposes(self):
        if self.training:
            ops.transpose(self.lm_head, out=self.lm_head_t)

    def load_state_dict(self, sd: Dict[str, torch.Tensor], strict: bool = False):
        """Reference-named state dict: model.embed_tokens.weight, model.layers.*,
        model.norm.weight, lm_head.weight, model.mm_projector.*, model.vision_tower.vision_tower.*,
        hand_traj_decoder.*.  Unused reference tensors (time_embedding, bbox_to_feature ...) are ignored."""
        dev = self.device
        V = self.config.vocab_size
        for name, dst in (("model.embed_tokens.weight", self.embed), ("lm_head.weight", self.lm_head)):
            src = sd[name]
            n = src.shape[0]
            assert n <= V, f"{name}: {n} rows, config.vocab_size is {V}"
            dst[:n].copy_(src.to(dev, BF16))
            if n < V:
                # a base LLM (Vicuna: 32000 rows) before the reference's initialize_constants added the 100 time tokens and
                # <hand_traj> (handsonvlm/model/handsonvlm_arch.py:17-31: tokenizer.add_tokens + resize_token_embeddings): the new
                # rows get HF's initializer, N(0, initializer_range = 0.02), here from a fixed seed
                g = torch.Generator().manual_seed(32100 + (0 if dst is self.embed else 1))
                dst[n:V].copy_((torch.randn(V - n, dst.shape[1], generator=g) * 0.02).to(dev, BF16))
        if "model.mm_projector.weight" in sd:                # (absent in a base LLM: the projector comes from the LLaVA adapter file)
            self.proj_w.copy_(sd["model.mm_projector.weight"].to(dev, BF16))
            self.proj_b.copy_(sd["model.mm_projector.bias"].to(dev, BF16))
        self.stack.load_hf(sd, "model.")
        head = {k[len("hand_traj_decoder."):]: v for k, v in sd.items() if k.startswith("hand_traj_decoder.")}
        if head:
            self.hand_traj_decoder.load_state_dict({k: v.to(torch.float32) for k, v in head.items()}, strict=True)
        vt = "model.vision_tower.vision_tower."
        if any(k.startswith(vt) for k in sd):
            self.tower.load_state_dict({k[len(vt):]: v for k, v in sd.items() if k.startswith(vt)})
        self._refresh_transposes()
        return self

    def state_dict(self) -> Dict[str, torch.Tensor]:
        V = self.config.vocab_size
        out = {"model.embed_tokens.weight": self.embed, "lm_head.weight": self.lm_head[:V],
               "model.mm_projector.weight": self.proj_w, "model.mm_projector.bias": self.proj_b}
        out.update(self.stack.export_hf("model."))
        out.update({"hand_traj_decoder." + k: v for k, v in self.hand_traj_decoder.state_dict().items()})
        return out

    def trainable_tensors(self) -> List[Tuple[str, torch.Tensor, torch.Tensor, Optional[torch.Tensor]]]:
        """(name, weight bf16, grad bf16, transposed copy or None) for every kernel-side parameter."""
        out = []
        for i, lp in enumerate(self.stack.layers):
            for k in lp.NAMES:
                out.append((f"layers.{i}.{k}", lp.w[k], lp.g[k], lp.wt.get(k)))
        out.append(("norm", self.stack.norm_w, self.stack.norm_g, None))
        out.append(("embed", self.embed, self.embed_g, None))
        out.append(("lm_head", self.lm_head, self.lm_head_g, self.lm_head_t))
        out.append(("proj_w", self.proj_w, self.proj_w_g, None))
        out.append(("proj_b", self.proj_b, self.proj_b_g, None))
        return out

    # ------------------------------------------------------------------------------------------
    # visual tokens (VisualToTokenHelper.pipeline, 'origin' + 'temporal_spatial_pool')
    # ------------------------------------------------------------------------------------------
    def visual_tokens(self, image: Optional[torch.Tensor], keep_for_backward: bool = False, fp8: bool = False,
                      image_frames=None) -> torch.Tensor:
        """image [B, T, 3, H, W] — or image_frames = (distinct frames [n, 3, H, W], frame index int32 [B * T], (B, T)), both on the
        device: what hvlm.prefetch.DevicePrefetcher uploads when the collator's 10 x 10 tiling is found on the host (the index
        replaces the device-side hash + verify of config.dedup_frames; frame (b, t) = distinct frame index[b * T + t])."""
        cfg = self.config
        assert cfg.fuse_input_mode == "origin" and cfg.video_compress_mode == "temporal_spatial_pool", \
            "only fuse_input_mode='origin' + video_arch='temporal_spatial_pool' are on the hot path"
        d = cfg.mm_hidden_size
        inv = None
        if image_frames is not None:
            frames, inv, (B, T) = image_frames
            assert frames.dim() == 4 and inv.numel() == B * T and frames.device == self.device and inv.device == self.device
            inv = inv.to(torch.int64)
        else:
            B, T = image.shape[:2]
            frames = image.reshape(B * T, *image.shape[2:])
        if cfg.dedup_frames and image_frames is None:
            # EK100 clips repeat a 10-frame sequence 10x (handsonvlm/dataset/epic_dataset.py:90-95: [f0..f9]*10):
            # encode each distinct frame once.  Frames are matched by a 64-bit hash and then VERIFIED bit-for-bit,
            # so the result is identical to encoding all frames (falls back to no de-duplication otherwise).
            flat = frames.reshape(B * T, -1).view(torch.int16)
            gen = torch.Generator(device=flat.device).manual_seed(0x5EED)
            wgt = torch.randint(1, 1 << 31, (flat.shape[1],), device=flat.device, generator=gen, dtype=torch.int64) | 1
            key = (flat.to(torch.int64) * wgt).sum(dim=1)
            uniq, inverse = torch.unique(key, return_inverse=True)
            first = torch.full((uniq.numel(),), B * T, dtype=torch.int64, device=flat.device)
            first.scatter_reduce_(0, inverse, torch.arange(B * T, device=flat.device), reduce="amin")
            if bool(torch.equal(flat, flat[first[inverse]])):
                inv = inverse
                frames = frames[first]
        n = frames.shape[0]
        NT = cfg.vision.num_patches + 1
        G = cfg.vision.image_size // cfg.vision.patch_size       # patches per side: 16 at the reference's 224 px
        NV = T + G * G
        # the last tower layer's fc2 + residual AFTER the pooling (config.pool_before_last_fc2): the tower hands back the pooled
        # [B, NV, d] features directly, chunked by whole samples
        if (cfg.pool_before_last_fc2 and cfg.project_after_pool and not fp8 and self.tower.wdtype == torch.float16
                and self.tower.n_run >= 1 and cfg.mm_vision_select_feature == "patch" and (inv is None or n <= cfg.clip_chunk_frames)):
            pooled = torch.empty((B, NV, d), dtype=BF16, device=self.device)
            if inv is not None:
                # de-duplicated frames: ONE tower pass over the distinct frames; the pooling gathers frame (b, t) = distinct frame
                # inv[b*T + t] — the same sums over the same values as the pass over all B*T frames: identical visual tokens
                pooled.copy_(self.tower.forward_tokens(frames, pool=(B, T), frame_index=inv.to(torch.int32).contiguous()))
            else:
                per = max(1, cfg.clip_chunk_frames // T)                 # samples per tower pass
                for b0 in range(0, B, per):
                    b1 = min(B, b0 + per)
                    pooled[b0:b1] = self.tower.forward_tokens(frames[b0 * T:b1 * T], pool=(b1 - b0, T))
            pooled = pooled.view(B * NV, d)
            vis = ops.gemm_nt(pooled, self.proj_w, bias=self.proj_b)
            if keep_for_backward:
                self._vis_saved = dict(pooled=pooled)
            return vis.view(B, NV, self.token_dim)
        tok = torch.empty((n, NT, d), dtype=BF16, device=self.device)          # [n, 257, d]: every chunk writes its slice
        for s in range(0, n, cfg.clip_chunk_frames):
            self.tower.forward_tokens(frames[s:s + cfg.clip_chunk_frames], out=tok[s:s + cfg.clip_chunk_frames], fp8=fp8)
        if inv is not None:
            tok = tok[inv]
        tok = tok.contiguous()
        if cfg.project_after_pool:
            pooled = ops.temporal_spatial_pool_fwd(tok, B, T, drop_cls=True, grid=G).view(B * NV, d)
            vis = ops.gemm_nt(pooled, self.proj_w, bias=self.proj_b)
            if keep_for_backward:
                self._vis_saved = dict(pooled=pooled)
        else:
            feats = tok[:, 1:].contiguous().view(B * T * G * G, d)
            ptok = ops.gemm_nt(feats, self.proj_w, bias=self.proj_b)
            vis = ops.temporal_spatial_pool_fwd(ptok, B, T, drop_cls=False, grid=G).view(B * NV, self.token_dim)
            if keep_for_backward:
                self._vis_saved = dict(feats=feats)
        return vis.view(B, NV, self.token_dim)

    # ------------------------------------------------------------------------------------------
    # splice plan (host)
    # ------------------------------------------------------------------------------------------
    def build_splice_plan(self, input_ids, labels, NV: int, n_hands: int, is_evaluate: bool,
                          has_future_hands: bool, pack: bool = False, attention_mask=None) -> SplicePlan:
        ids_all = input_ids.detach().cpu().numpy().astype(np.int64)
        lab_all = None if labels is None else labels.detach().cpu().numpy().astype(np.int64)
        B, T = ids_all.shape
        # packed rows: the collator's right padding (attention_mask False at the tail, hybrid_dataset.py:151-157) is dropped —
        # exact, because pads sit after every real token under the causal mask and carry label -100
        keep = np.full(B, T)
        if pack and attention_mask is not None:
            am = attention_mask.detach().cpu().numpy().astype(bool)
            keep = np.array([int(np.nonzero(am[b])[0][-1]) + 1 if am[b].any() else 1 for b in range(B)])
        rows_src, rows_lab, rows_slot, scales = [], [], [], []
        for b in range(B):
            idb = ids_all[b, :keep[b]]
            labb = None if lab_all is None else lab_all[b, :keep[b]]
            T = int(keep[b])
            img = np.nonzero(idb == IMAGE_TOKEN_INDEX)[0]
            if img.size == 0:
                src = idb.copy()
                nl = None if labb is None else labb.copy()
                slot = np.full(T, -1, dtype=np.int64)
                scale = 0.0
            else:
                if img.size != 1:
                    raise NotImplementedError("one <image> token per sample (the reference indexes visual tokens "
                                              "with a running counter, handsonvlm.py:231-300)")
                p = int(img[0])
                tail = idb[p + 1:]
                src = np.concatenate([idb[:p], -2 - np.arange(NV), tail])
                nl = None if labb is None else np.concatenate([labb[:p], np.full(NV, IGNORE_INDEX), labb[p + 1:]])
                slot = np.full(src.shape[0], -1, dtype=np.int64)
                ts = p + NV
                scale = 0.0
                if tail.size > 0:
                    hp = np.nonzero(tail == HAND_TOKEN_ID)[0]
                    cnt = hp.size
                    if not is_evaluate:
                        assert cnt <= 4, f"more than 4 <hand_traj> tokens ({cnt})"
                        idx = list(hp) + [0] * (4 - cnt)           # missing ones land on row 0 of the chunk
                        for j in range(4):
                            slot[ts + idx[j]] = j                  # scatter: later write wins (:368-375)
                        scale = cnt / 4.0
                    elif has_future_hands:
                        assert n_hands == cnt, f"gt_hand_num: {n_hands}, hand_token_cnt: {cnt}"
                        for j in range(cnt):
                            slot[ts + hp[j]] = j
                        scale = 1.0
            rows_src.append(src); rows_lab.append(nl); rows_slot.append(slot); scales.append(scale)
        lengths = np.array([r.shape[0] for r in rows_src])
        S = int(lengths.max())
        if pack:
            return self._packed_plan(rows_src, rows_lab, rows_slot, scales, lengths, S, lab_all is not None)
        src = np.full((B, S), -1, dtype=np.int32)
        slot = np.full((B, S), -1, dtype=np.int32)
        lab = lab_all
        nl = None if lab is None else np.full((B, S), IGNORE_INDEX, dtype=np.int64)
        for b in range(B):
            n = lengths[b]
            src[b, :n] = rows_src[b]; slot[b, :n] = rows_slot[b]
            if nl is not None:
                nl[b, :n] = rows_lab[b]
        dev = self.device
        targets = hand_rows = has_hand = None
        if nl is not None:
            tg = np.full((B, S), IGNORE_INDEX, dtype=np.int32)
            tg[:, :-1] = nl[:, 1:]
            targets = torch.from_numpy(tg.reshape(-1)).to(dev)
            vidx = np.nonzero(tg.reshape(-1) != IGNORE_INDEX)[0]
            hr = np.full((B, 4), -1, dtype=np.int32)
            has_hand = np.zeros(B, dtype=bool)
            for b in range(B):
                pos = np.nonzero(nl[b, 1:] == HAND_TOKEN_ID)[0]    # shifted mask (:167-170)
                if pos.size:
                    assert pos.size == 4, f"expected 4 <hand_traj> labels, got {pos.size}"
                    hr[b] = b * S + pos
                    has_hand[b] = True
            hand_rows = torch.from_numpy(hr.reshape(-1)).to(dev)
        seg = None
        if nl is not None and getattr(self, "training", False):
            seg = self._embed_segments(src.reshape(-1))
        return SplicePlan(seg=seg, M=B * S, B=B, S=S, src=torch.from_numpy(src.reshape(-1)).to(dev),
                          hand_slot=torch.from_numpy(slot.reshape(-1)).to(dev),
                          hand_scale=torch.tensor(scales, dtype=torch.float32, device=dev),
                          labels=None if nl is None else torch.from_numpy(nl).to(dev), targets=targets,
                          hand_rows=hand_rows, has_hand=has_hand, lengths=lengths, **self._valid_rows(nl, vidx if nl is not None else None, tg if nl is not None else None, hr if nl is not None else None))

    def _valid_rows(self, nl, vidx, tg, hr) -> dict:
        """Rows that carry a target, resolved on the host with the rest of the plan (config.lm_head_labelled_rows)."""
        if nl is None or vidx is None or not self.config.lm_head_labelled_rows:
            return {}
        dev = self.device
        hf = hr.reshape(-1)
        hv = np.full(hf.shape, -1, dtype=np.int32)
        if vidx.size and (hf >= 0).any():
            at = np.searchsorted(vidx, hf[hf >= 0])
            assert (at < vidx.size).all() and (vidx[at] == hf[hf >= 0]).all(), "a <hand_traj> row without a target"
            hv[hf >= 0] = at.astype(np.int32)
        return dict(rows_v=torch.from_numpy(vidx.astype(np.int64)).to(dev),
                    targets_v=torch.from_numpy(np.ascontiguousarray(tg.reshape(-1)[vidx])).to(dev), n_v=int(vidx.size),
                    hand_rows_v=torch.from_numpy(hv).to(dev))

    def _embed_segments(self, flat_src: np.ndarray):
        """Text rows grouped by token id in a FIXED order (stable sort): the embedding gradient is summed segment by
        segment without atomics (hvlm_embed_grad_segments), so the step is run-to-run bit-identical."""
        dev = self.device
        rows = np.nonzero(flat_src >= 0)[0].astype(np.int32)
        order = np.argsort(flat_src[rows], kind="stable")
        rows = rows[order]
        toks = flat_src[rows]
        starts = np.concatenate([[0], np.nonzero(np.diff(toks))[0] + 1, [rows.size]]).astype(np.int32)
        return (torch.from_numpy(rows).to(dev), torch.from_numpy(starts).to(dev),
                torch.from_numpy(toks[starts[:-1]].astype(np.int32)).to(dev))

    def _packed_plan(self, rows_src, rows_lab, rows_slot, scales, lengths, S, has_labels) -> SplicePlan:
        """The same per-sample layouts stored end to end, without padding rows (the reference unpads around attention
        only, llama_flash_attn_monkey_patch.py:81-96; here every token-wise kernel skips the padding)."""
        dev = self.device
        B = len(rows_src)
        cu = np.zeros(B + 1, dtype=np.int32)
        cu[1:] = np.cumsum(lengths)
        M = int(cu[-1])
        src = np.concatenate(rows_src).astype(np.int32)
        slot = np.concatenate(rows_slot).astype(np.int32)
        pos = np.concatenate([np.arange(n, dtype=np.int32) for n in lengths])
        row_sample = np.repeat(np.arange(B, dtype=np.int32), lengths)
        nl = targets = hand_rows = has_hand = seg = None
        if has_labels:
            nl = np.full((B, S), IGNORE_INDEX, dtype=np.int64)          # padded labels, as the reference returns them
            tg = np.full(M, IGNORE_INDEX, dtype=np.int32)
            hr = np.full((B, 4), -1, dtype=np.int32)
            has_hand = np.zeros(B, dtype=bool)
            for b in range(B):
                n = lengths[b]
                nl[b, :n] = rows_lab[b]
                tg[cu[b]:cu[b] + n - 1] = rows_lab[b][1:]                 # shifted inside the sample; last row ignored
                p = np.nonzero(rows_lab[b][1:] == HAND_TOKEN_ID)[0]
                if p.size:
                    assert p.size == 4, f"expected 4 <hand_traj> labels, got {p.size}"
                    hr[b] = cu[b] + p
                    has_hand[b] = True
            targets = torch.from_numpy(tg).to(dev)
            vidx = np.nonzero(tg != IGNORE_INDEX)[0]
            hand_rows = torch.from_numpy(hr.reshape(-1)).to(dev)
            if getattr(self, "training", False):
                seg = self._embed_segments(src)
        return SplicePlan(B=B, S=S, M=M, src=torch.from_numpy(src).to(dev), hand_slot=torch.from_numpy(slot).to(dev),
                          hand_scale=torch.tensor(scales, dtype=torch.float32, device=dev),
                          labels=None if nl is None else torch.from_numpy(nl).to(dev), targets=targets, hand_rows=hand_rows,
                          has_hand=has_hand, lengths=lengths, seg=seg, cu=torch.from_numpy(cu).to(dev),
                          pos=torch.from_numpy(pos).to(dev), row_sample=torch.from_numpy(row_sample).to(dev),
                          **self._valid_rows(nl, vidx if has_labels else None, tg if has_labels else None, hr if has_labels else None))

    # ------------------------------------------------------------------------------------------
    # forward
    # ------------------------------------------------------------------------------------------
    def forward(self, input_ids=None, attention_mask=None, past_key_values=None, inputs_embeds=None, labels=None,
                image=None, future_hands=None, contact_point=None, future_valid=None, gt_label_valid=None,
                use_cache=None, output_attentions=None, output_hidden_states=None, return_dict=None, **kwargs):
        """handsonvlm.py:70-210.  NOTE on `out.logits` in a training pass: it is a view of the engine's logits buffer,
        which backward() overwrites IN PLACE with d(loss)/d(logits) (no second [M, 32128] buffer: 0.9 GB at B = 16);
        read or clone the logits before calling backward, or pass keep_logits=True to get a detached copy."""
        cfg = self.config
        dev = self.device
        image_frames = kwargs.get("image_frames")            # (distinct frames, frame index, (B, T)) from hvlm.prefetch, instead of `image`
        B = image.shape[0] if image is not None else image_frames[2][0]
        if labels is not None:
            assert future_hands.shape == (B, 2, 5, 2), future_hands.shape
            assert contact_point is None or contact_point.shape == (B, 2), contact_point.shape
            assert future_valid.shape == (B, 2), future_valid.shape
            assert gt_label_valid is None or gt_label_valid.shape == (B,), gt_label_valid.shape
            future_hands = future_hands[:, :, 1:, :]
        if image is not None:
            assert image.dim() == 5 and image.shape[:3] == (B, cfg.num_frames, 3), image.shape
        else:
            assert image_frames[2] == (B, cfg.num_frames) and image_frames[0].shape[1] == 3, image_frames[2]
        n_frames = cfg.num_frames
        assert not use_cache and past_key_values is None, "KV cache is handled inside sample(); forward is cache-less"
        is_evaluate = kwargs.get("is_evaluate", False)
        training_pass = labels is not None and self.training
        T = input_ids.shape[1]

        vis = kwargs.get("visual_tokens")
        G = cfg.vision.image_size // cfg.vision.patch_size
        NV = vis.shape[1] if vis is not None else n_frames + G * G      # temporal + spatial tokens (356 at 224 px)
        n_hands = 0 if future_hands is None else future_hands.shape[2]
        # packed (unpadded) rows: training passes only, opt-in (config.pack_sequences); logits / hidden states are then
        # returned packed too ([M, ...] with out.cu_seqlens), everything else is unchanged
        pack = bool(cfg.pack_sequences) and training_pass
        # the splice plan is host-only work on the token ids: built BEFORE the CLIP tower is enqueued, so that the data-parallel
        # exchange of this step's token ids (dp.EmbedRows: which embedding-gradient rows need reducing at all) happens while the
        # device is still busy with the previous step, not in the middle of this one
        plan = self.build_splice_plan(input_ids, labels, NV, n_hands, is_evaluate, future_hands is not None, pack=pack,
                                      attention_mask=attention_mask)
        if training_pass and self.on_token_ids is not None and plan.seg is not None:
            self.on_token_ids(plan.seg[2])
        if vis is None:
            vis = self.visual_tokens(image.to(dev) if image is not None else None, keep_for_backward=training_pass, image_frames=image_frames)
        assert vis.shape[1] == NV, (vis.shape, NV)
        S = plan.S
        assert S <= cfg.max_position_embeddings, f"spliced length {S} exceeds max_position_embeddings"
        hands_dev = None
        if future_hands is not None and n_hands > 0:
            hands_dev = future_hands.detach().to(dev, torch.float32).contiguous()
        x = ops.embed_splice_fwd(self.embed, vis, plan.src, plan.hand_slot if hands_dev is not None else None,
                                 hands_dev, plan.hand_scale if hands_dev is not None else None, B, S, self.token_dim,
                                 NV, max(n_hands, 1), row_sample=plan.row_sample)
        shape3 = (lambda t: t) if pack else (lambda t: t.view(B, S, -1))
        last_only = kwargs.get("last_logits_only", False) and labels is None
        # labelled rows only (config.lm_head_labelled_rows, training passes that do not hand out the logits): the rows without a
        # target — visual tokens, the prompt, padding — have zero loss and zero gradient, so lm_head, the cross-entropy and their
        # backward run on the gathered [n_v, D] rows; same loss, same gradients, `out.logits` is None.  Without
        # output_hidden_states the LAST decoder layer's o_proj and MLP run on those rows only as well (`pruned`, LlamaStack.forward).
        compact = (training_pass and labels is not None and not last_only and plan.n_v > 0
                   and not kwargs.get("keep_logits", False))
        pruned = compact and not output_hidden_states
        if output_hidden_states:
            hidden, hs = self.stack.forward(x, B, S, save=training_pass, all_hidden=True, cu_seqlens=plan.cu, pos=plan.pos)
            hs = tuple(shape3(h) for h in hs)
        else:
            hidden, hs = self.stack.forward(x, B, S, save=training_pass, cu_seqlens=plan.cu, pos=plan.pos,
                                            last_rows=plan.rows_v if pruned else None), None
        if last_only:
            rows = torch.from_numpy((np.arange(B) * S + plan.lengths - 1).astype(np.int64)).to(dev)
            self._last_hidden = hidden[rows].contiguous()                     # read by sample() on <hand_traj>
            logits_p = ops.gemm_nt(self._last_hidden, self.lm_head)
            logits = logits_p[:, : cfg.vocab_size].unsqueeze(1)
        hidden_v = None
        if last_only:
            pass
        elif compact:
            hidden_v = hidden if pruned else hidden.index_select(0, plan.rows_v)      # (pruned: the stack returned [n_v, D])
            logits_p = ops.gemm_nt(hidden_v, self.lm_head)                   # [n_v, Vpad] bf16
            logits = None
        else:
            logits_p = ops.gemm_nt(hidden, self.lm_head)                     # [M, Vpad] bf16
            logits = logits_p[:, : cfg.vocab_size] if pack else logits_p.view(B, S, -1)[:, :, : cfg.vocab_size]
        loss = comps = None
        if labels is not None:
            acc, row_lse = ops.ce_fwd(logits_p, plan.targets_v if compact else plan.targets, cfg.vocab_size)
            loss_text = acc[0] / acc[1]
            # hand-token gather (:146-187); samples without hand labels get zeros and future_valid[b] = 0
            fv = future_valid
            if not plan.has_hand.all():
                fv[torch.from_numpy(~plan.has_hand).to(fv.device)] = 0      # in place, like the reference (:178)
            emb = ops.hand_gather_fwd(hidden, plan.hand_rows_v if pruned else plan.hand_rows, B, self.token_dim, torch.float32)
            emb.requires_grad_(training_pass)
            with torch.enable_grad():
                tl = self.hand_traj_decoder(emb, future_hands.to(dev, torch.float32), fv.to(dev, torch.float32),
                                            cfg.lambda_traj, cfg.lambda_traj_kl, eps=kwargs.get("cvae_eps"))
                head_loss = cfg.hoi_lambda * tl["total_loss"]
            value = head_loss.detach() + loss_text
            comps = {"text loss": loss_text, "traj_loss": tl["traj_loss"].detach(), "traj_kl_loss": tl["traj_kl_loss"].detach()}
            if training_pass:
                self._ctx = dict(plan=plan, logits_p=logits_p, row_lse=row_lse, n_valid=acc[1:2], hidden=hidden, emb=emb, hidden_v=hidden_v, pruned=pruned,
                                 head_loss=head_loss, B=B, S=S, NV=NV, T=n_frames, M=plan.M)
                loss = _EngineLoss.apply(self._anchor, self, value)
            else:
                loss = value
        if training_pass and kwargs.get("keep_logits", False) and logits is not None:
            logits = logits.clone()
        if compact and not last_only:
            if return_dict is False:
                raise RuntimeError("return_dict=False needs the full logits: set config.lm_head_labelled_rows=False or pass keep_logits=True")
            return LabelledRowsOutput(logits_rows=logits_p[:, : cfg.vocab_size], rows=plan.rows_v, loss=loss, past_key_values=None,
                                      hidden_states=hs, attentions=None, loss_components=comps, cu_seqlens=plan.cu)
        out = CausalLMOutputWithPast(loss=loss, logits=logits, past_key_values=None, hidden_states=hs, attentions=None,
                                     loss_components=comps, cu_seqlens=plan.cu)
        if return_dict is False:
            return (loss, logits) if loss is not None else (logits,)
        return out

    __call__ = forward

    # ------------------------------------------------------------------------------------------
    # backward (explicit engine; also reached through loss.backward())
    # ------------------------------------------------------------------------------------------
    def backward(self, grad_scale=None):
        """Back-propagates d(loss)*grad_scale into every trainable tensor's grad buffer (overwrite)."""
        c = self._ctx
        assert c is not None, "forward with labels (training) must precede backward"
        self._ctx = None
        cfg, dev = self.config, self.device
        plan, B, S, NV = c["plan"], c["B"], c["S"], c["NV"]
        M = c["M"]
        mpad = _round_up(M, 64)
        if grad_scale is None:
            grad_scale = torch.ones((), device=dev)
        grad_scale = grad_scale.to(dev, torch.float32).reshape(())
        gs = (grad_scale / c["n_valid"]).reshape(1).contiguous()
        hand_rows = plan.hand_rows
        if c.get("hidden_v") is not None:         # labelled rows only: d(hidden) of every other row is exactly zero
            dlogits = ops.ce_bwd_(c["logits_p"], plan.targets_v, c["row_lse"], gs, cfg.vocab_size)
            dh_v = ops.gemm_nt(dlogits, self.lm_head_t)                       # [n_v, D]
            if c.get("pruned"):                   # the stack's last layer lives on these rows too: hand d(hidden) [n_v, D] over
                dhidden, hand_rows = dh_v, plan.hand_rows_v
            else:
                dhidden = torch.zeros((M, self.token_dim), dtype=BF16, device=dev)
                dhidden.index_copy_(0, plan.rows_v, dh_v)
            self.stack._dw(dlogits, c["hidden_v"], self.lm_head_g, _round_up(plan.n_v, 64))
            del dh_v
        else:
            dlogits = ops.ce_bwd_(c["logits_p"], plan.targets, c["row_lse"], gs, cfg.vocab_size)
            dhidden = ops.gemm_nt(dlogits, self.lm_head_t)                    # [M, D]
            self.stack._dw(dlogits, c["hidden"], self.lm_head_g, mpad)
        del dlogits
        c["logits_p"] = None
        # CVAE head (PyTorch autograd): grads of the head's own params + d(emb)
        for p in self.hand_traj_decoder.parameters():
            p.grad = None
        with torch.enable_grad():
            (c["head_loss"] * grad_scale).backward()
        ops.hand_gather_bwd_(c["emb"].grad.contiguous(), hand_rows, dhidden, B, self.token_dim)
        if self.on_grads_ready is not None:
            self.stack._join_side()              # lm_head's gradient was computed on the side stream
            self.on_grads_ready("head")
        dx = self.stack.backward(dhidden)
        self.embed_g.zero_()
        # zeros, not empty: the splice backward only writes the rows of samples that contain an <image> token; a text-only
        # sample (forward supports it, like handsonvlm.py:234-244) must contribute ZERO projector gradient
        dvis = torch.zeros((B * NV, self.token_dim), dtype=BF16, device=dev)
        ops.embed_splice_bwd(dx, plan.src, None, dvis, B, S, self.token_dim, NV, row_sample=plan.row_sample)   # visual rows only
        ops.embed_grad_segments(dx, *plan.seg, self.embed_g)
        sv = self._vis_saved
        if cfg.project_after_pool:
            self.stack._dw(dvis, sv["pooled"], self.proj_w_g, _round_up(B * NV, 64))
            ops.cast_f32_to_bf16(ops.colsum(dvis), out=self.proj_b_g)
        else:
            G = cfg.vision.image_size // cfg.vision.patch_size
            dtok = ops.temporal_spatial_pool_bwd(dvis.view(B, NV, -1), B, c["T"], grid=G).view(-1, self.token_dim)
            self.stack._dw(dtok, sv["feats"], self.proj_w_g, _round_up(dtok.shape[0], 64))
            ops.cast_f32_to_bf16(ops.colsum(dtok), out=self.proj_b_g)
        self._vis_saved = None
        self.stack._join_side()                  # every weight gradient is final on the main stream from here on
        if self.on_grads_ready is not None:
            self.on_grads_ready("tail")

    # ------------------------------------------------------------------------------------------
    # generation (custom sampling loop of the reference, batch 1)
    # ------------------------------------------------------------------------------------------
    def prepare_inputs_for_generation(self, input_ids, past_key_values=None, attention_mask=None, inputs_embeds=None,
                                      **kwargs):
        return {"input_ids": input_ids, "past_key_values": None, "use_cache": False, "attention_mask": attention_mask,
                "image": kwargs.get("image"), "is_evaluate": True}

    @torch.no_grad()
    def _prefill(self, input_ids, vis, fh, cache):
        """Prompt pass of KV-cached generation: fills the per-layer caches, returns (last logits [1,V], length)."""
        cfg = self.config
        NV = vis.shape[1]
        n_hands = 0 if fh is None else fh.shape[2]
        plan = self.build_splice_plan(input_ids, None, NV, n_hands, True, fh is not None)
        S = plan.S
        hands_dev = None if fh is None else fh.detach().to(self.device, torch.float32).contiguous()
        x = ops.embed_splice_fwd(self.embed, vis, plan.src, plan.hand_slot if hands_dev is not None else None, hands_dev,
                                 plan.hand_scale if hands_dev is not None else None, 1, S, self.token_dim, NV, max(n_hands, 1))
        hidden = self.stack.forward(x, 1, S, save=False, cache=cache)
        self._last_hidden = hidden[S - 1:S].contiguous()
        logits = ops.gemv(self.lm_head, self._last_hidden.view(-1))
        return logits[: cfg.vocab_size].unsqueeze(0), S

    @torch.no_grad()
    def _decode(self, tok: int, hand: Optional[torch.Tensor], pos: int, cache):
        """One KV-cached step: embed the token appended last (plus the sinusoidal encoding of the hand it produced,
        handsonvlm.py:380-396), run the decoder on that single row, return logits [1,V]."""
        cfg, dev = self.config, self.device
        src = torch.tensor([tok], dtype=torch.int32, device=dev)
        if hand is not None:
            slot = torch.zeros(1, dtype=torch.int32, device=dev)
            hands = hand.detach().to(dev, torch.float32).reshape(1, 2, 1, 2).contiguous()
            sc = torch.ones(1, dtype=torch.float32, device=dev)
            x1 = ops.embed_splice_fwd(self.embed, None, src, slot, hands, sc, 1, 1, self.token_dim, 0, 1)
        else:
            x1 = ops.embed_splice_fwd(self.embed, None, src, None, None, None, 1, 1, self.token_dim, 0, 1)
        hidden = self.stack.decode_step(x1, pos, cache)
        self._last_hidden = hidden
        logits = ops.gemv(self.lm_head, hidden.view(-1))
        return logits[: cfg.vocab_size].unsqueeze(0)

    @torch.no_grad()
    def sample(self, input_ids, logits_processor=None, stopping_criteria=None, logits_warper=None, max_length=None,
               pad_token_id=None, eos_token_id=None, output_attentions=None, output_hidden_states=None,
               output_scores=None, return_dict_in_generate=None, synced_gpus=False, streamer=None, **model_kwargs):
        """Multinomial sampling loop with hand decoding on token 32100 (handsonvlm.py:478-672).
        Batch 1 (the reference calls .item() on the sampled token).  Differences from the reference,
        none of which change results: the CLIP tower runs once per call instead of once per generated
        token (visual tokens do not depend on the text), and only the last position's logits are
        computed.  `hand_noise` (optional list of z tensors [2, latent], already scaled) pins the
        CVAE noise for reproducible tests."""
        cfg, dev = self.config, self.device
        assert input_ids.shape[0] == 1, "sample() is batch-1 like the reference"
        procs = list(logits_processor or []) + list(logits_warper or [])
        pad_token_id = cfg.pad_token_id if pad_token_id is None else pad_token_id
        eos = cfg.eos_token_id if eos_token_id is None else eos_token_id
        eos = [eos] if isinstance(eos, int) else list(eos)
        image = model_kwargs.get("image")
        hand_noise = model_kwargs.get("hand_noise")
        vis = self.visual_tokens(image.to(dev))
        input_ids = input_ids.to(dev)
        pred_hands: List[torch.Tensor] = []
        scores = []
        n_new = 0
        use_kv = model_kwargs.get("use_kv_cache", True)
        # cache rows: the spliced prompt (T - 1 + NV) plus every token that can still be generated (max_length counts
        # ids, like the reference's stopping criterion), capped by the position table like the reference (2048)
        n_gen = 160 if max_length is None else max(1, max_length - input_ids.shape[1])
        cache = self.stack.new_cache(min(cfg.max_position_embeddings, input_ids.shape[1] + vis.shape[1] + n_gen)) if use_kv else None
        cur_len = 0
        tok = -1
        while True:
            fh = torch.stack(pred_hands, dim=2) if pred_hands else None
            if not use_kv:
                out = self.forward(input_ids=input_ids, image=image, future_hands=fh, is_evaluate=True, visual_tokens=vis,
                                   last_logits_only=True, output_hidden_states=False)
                logits_last = out.logits[:, -1, :]
            elif n_new == 0:
                logits_last, cur_len = self._prefill(input_ids, vis, fh, cache)
            else:
                tok_in = tok                                     # (the token drawn in the previous iteration: no second device read-back)
                hand_in = pred_hands[-1] if tok_in == HAND_TOKEN_ID and pred_hands else None
                if cur_len + 1 > cache[0].shape[0]:
                    raise RuntimeError(f"sequence exceeds the KV cache ({cache[0].shape[0]} rows = min(max_position_embeddings, "
                                       "prompt + visual tokens + max_new_tokens)); the reference fails at the same length")
                logits_last = self._decode(tok_in, hand_in, cur_len, cache)
                cur_len += 1
            next_scores = logits_last.float()
            for proc in procs:
                next_scores = proc(input_ids, next_scores)
            if output_scores:
                scores.append(next_scores)
            probs = torch.softmax(next_scores, dim=-1)
            next_tokens = torch.multinomial(probs, num_samples=1).squeeze(1)
            tok = int(next_tokens.item())
            if tok == HAND_TOKEN_ID:
                h = self._last_hidden                                           # [1, D]
                emb = h.float().reshape(1, self.token_dim // 2, 2).permute(0, 2, 1).unsqueeze(2)   # [1,2,1,D/2]
                z = None if hand_noise is None else hand_noise[len(pred_hands)].to(dev, torch.float32)
                pred_hands.append(self.hand_traj_decoder.inference(emb, z).squeeze(2))
            input_ids = torch.cat([input_ids, next_tokens[:, None]], dim=-1)
            n_new += 1
            if streamer is not None:
                streamer.put(next_tokens.cpu())
            done = tok in eos
            if stopping_criteria is not None:
                sc = stopping_criteria(input_ids, None)
                done = done or bool(sc.all() if torch.is_tensor(sc) else sc)
            if max_length is not None and input_ids.shape[1] >= max_length:
                done = True
            if done:
                break
        if streamer is not None:
            streamer.end()
        ph = torch.stack(pred_hands, dim=2) if pred_hands else torch.empty(1, 2, 0, 2, device=dev)
        if return_dict_in_generate is False:
            return input_ids
        return SampleDecoderOnlyOutput(sequences=input_ids, scores=tuple(scores) if output_scores else None,
                                       pred_hands=ph)

    @torch.no_grad()
    def sample_batch(self, input_ids, logits_processor=None, max_length=None, eos_token_id=None, pad_token_id=None,
                     attention_mask=None, image=None, hand_noise=None, fp8: bool = False, fp8_decode: bool = False,
                     fp8_clip: bool = False, output_scores: bool = False):
        """Lock-step generation of B samples (BASELINE config 5: batch-32 evaluation).  The reference's sample() is batch 1
        (`.item()` at handsonvlm.py:609) and re-runs CLIP + the whole prefix per token; this is the same per-sample function
        — multinomial sampling, a CVAE hand decoded from the last hidden state whenever token 32100 is drawn, that hand's
        sinusoidal encoding added to the token's embedding when it is fed back (handsonvlm.py:380-396,543-647) — with one
        batched prefill (bf16, or the e4m3 GEMMs with `fp8`) filling a [B, max_len, 3D] KV cache per layer and one batched
        decode step per generated token (weights streamed once for all B samples; e4m3 weights with `fp8_decode`).
        `fp8_clip` runs the CLIP tower's layer GEMMs on e4m3 operands as well.  Right-padded prompts (attention_mask)
        are supported: a sample's pad rows are never attended (its own length bounds its keys).
        `logits_processor`: callables (input_ids [B, cur], scores [B, V]) -> scores.  `hand_noise[b][j]`: optional z for the
        j-th hand of sample b.  Returns SampleDecoderOnlyOutput(sequences [B, T_max] padded with pad_token_id,
        pred_hands [B, 2, n_max, 2] zero-padded) plus `.lengths` / `.num_hands` lists."""
        cfg, dev = self.config, self.device
        B, T = input_ids.shape
        procs = list(logits_processor or [])
        pad_token_id = cfg.pad_token_id if pad_token_id is None else pad_token_id
        eos = cfg.eos_token_id if eos_token_id is None else eos_token_id
        eos = [eos] if isinstance(eos, int) else list(eos)
        vis = self.visual_tokens(image.to(dev), fp8=fp8_clip)
        NV = vis.shape[1]
        plan = self.build_splice_plan(input_ids, None, NV, 0, True, False, pack=False)
        if attention_mask is not None:          # true prompt lengths (right padding): rows past them are pads
            am = attention_mask.detach().cpu().numpy().astype(bool)
            # (an all-False row = an empty prompt: one row is kept, as build_splice_plan does)
            npad = np.array([T - (int(np.nonzero(am[b])[0][-1]) + 1 if am[b].any() else 1) for b in range(B)])
        else:
            npad = np.zeros(B, dtype=np.int64)
        S = plan.S
        lengths = plan.lengths - npad
        n_gen = 160 if max_length is None else max(1, max_length - T)
        max_len = min(cfg.max_position_embeddings, S + n_gen)
        cache = self.stack.new_cache(max_len, batch=B)
        x = ops.embed_splice_fwd(self.embed, vis, plan.src, None, None, None, B, S, self.token_dim, NV, 1)
        hidden = self.stack.forward(x, B, S, save=False, cache=cache, fp8=fp8)
        rows = torch.from_numpy((np.arange(B) * S + lengths - 1).astype(np.int64)).to(dev)
        last = hidden[rows].contiguous()                                        # [B, D]
        del hidden, x
        lens = torch.from_numpy(lengths.astype(np.int32)).to(dev)
        lens_h = lengths.astype(np.int64).copy()                                # host mirror of `lens`
        seqs = [input_ids[b, : T - npad[b]].tolist() for b in range(B)]
        hands: List[List[torch.Tensor]] = [[] for _ in range(B)]
        done = np.zeros(B, dtype=bool)
        scores_out = []
        cur = input_ids.to(dev)
        while True:
            logits = ops.gemm_nt(last, self.lm_head)[:, : cfg.vocab_size].float()
            for proc in procs:
                logits = proc(cur, logits)
            if output_scores:
                scores_out.append(logits)
            toks = torch.multinomial(torch.softmax(logits, dim=-1), num_samples=1).squeeze(1)
            tl = toks.tolist()
            is_hand = [(not done[b]) and tl[b] == HAND_TOKEN_ID for b in range(B)]
            hb = [b for b in range(B) if is_hand[b]]
            hand_new = torch.zeros(B, 2, 1, 2, dtype=torch.float32, device=dev)
            if hb:
                emb = last[hb].float().reshape(len(hb), self.token_dim // 2, 2).permute(0, 2, 1).unsqueeze(2)   # [n,2,1,D/2]
                z = None
                if hand_noise is not None:
                    z = torch.cat([hand_noise[b][len(hands[b])].to(dev, torch.float32).reshape(2, -1) for b in hb], 0)
                ph = self.hand_traj_decoder.inference(emb, z).squeeze(2)       # [n, 2, 2]
                for i, b in enumerate(hb):
                    hands[b].append(ph[i])
                    hand_new[b, :, 0] = ph[i]
            for b in range(B):
                if not done[b]:
                    seqs[b].append(tl[b])
                    n_new = len(seqs[b]) - (T - int(npad[b]))                # tokens generated by this sample so far
                    if tl[b] in eos or (max_length is not None and n_new >= max_length - T):
                        done[b] = True
            cur = torch.cat([cur, toks[:, None]], dim=1)
            # the cache bounds the samples still generating; a finished sample's rows keep advancing in lock step but are
            # never read again, so only the live ones can end the batch early
            if done.all() or int(lens_h[~done].max()) + 1 >= max_len:
                break
            # feed the drawn tokens back: embedding (+ the sinusoidal encoding of the hand this token just produced)
            src = toks.to(torch.int32)
            slot = torch.tensor([0 if h else -1 for h in is_hand], dtype=torch.int32, device=dev)
            x1 = ops.embed_splice_fwd(self.embed, None, src, slot, hand_new.contiguous(), torch.ones(B, device=dev), B, 1,
                                      self.token_dim, 0, 1)
            # (finished samples stay at their last row: their writes must not run past the cache)
            last = self.stack.decode_step_batched(x1, lens, cache, fp8=fp8_decode)
            lens_h[~done] += 1
            lens = lens + torch.from_numpy((~done).astype(np.int32)).to(dev)
        tmax = max(len(q) for q in seqs)
        out_ids = torch.full((B, tmax), pad_token_id, dtype=torch.long)
        for b in range(B):
            out_ids[b, : len(seqs[b])] = torch.tensor(seqs[b])
        nmax = max((len(h) for h in hands), default=0)
        ph = torch.zeros(B, 2, nmax, 2, device=dev)
        for b in range(B):
            if hands[b]:
                ph[b, :, : len(hands[b])] = torch.stack(hands[b], dim=1)
        out = SampleDecoderOnlyOutput(sequences=out_ids, scores=tuple(scores_out) if output_scores else None, pred_hands=ph)
        out.lengths = [len(q) for q in seqs]
        out.num_hands = [len(h) for h in hands]
        return out

    @torch.no_grad()
    def generate(self, input_ids, image=None, do_sample=True, temperature=1.0, top_p=1.0, num_beams=1,
                 max_new_tokens=100, use_cache=False, output_hidden_states=True, return_dict_in_generate=True, **kw):
        """The call made by handsonvlm/evaluation/handsonvlm_inference.py:99-109."""
        assert num_beams == 1
        procs = _generation_processors(do_sample, temperature, top_p)
        return self.sample(input_ids, logits_processor=procs, max_length=input_ids.shape[1] + max_new_tokens,
                           image=image, return_dict_in_generate=return_dict_in_generate, **kw)

    @torch.no_grad()
    def generate_batch(self, input_ids, image=None, attention_mask=None, do_sample=True, temperature=1.0, top_p=1.0,
                       max_new_tokens=100, fp8: bool = False, **kw):
        """generate() for a right-padded batch of prompts (input_ids [B, T], attention_mask [B, T]): the same decoding recipe,
        B samples in lock step (sample_batch)."""
        procs = _generation_processors(do_sample, temperature, top_p)
        return self.sample_batch(input_ids, logits_processor=procs, max_length=input_ids.shape[1] + max_new_tokens,
                                 attention_mask=attention_mask, image=image, fp8=fp8, **kw)


def _generation_processors(do_sample: bool, temperature: float, top_p: float):
    """Logits processors of the reference's generate() call (temperature, top-p; greedy when do_sample is False); every one
    works on a [B, V] batch of scores."""
    procs = []
    if do_sample and temperature != 1.0:
        procs.append(lambda ids, s: s / temperature)
    if do_sample and top_p < 1.0:
        procs.append(lambda ids, s: _top_p_filter(s, top_p))
    if not do_sample:
        procs.append(lambda ids, s: torch.full_like(s, float("-inf")).scatter(1, s.argmax(-1, keepdim=True), 0.0))
    return procs


def _top_p_filter(scores: torch.Tensor, top_p: float) -> torch.Tensor:
    """TopPLogitsWarper (transformers): keep the smallest set of tokens with cumulative prob >= top_p."""
    sorted_logits, sorted_idx = torch.sort(scores, descending=False)
    cum = sorted_logits.softmax(dim=-1).cumsum(dim=-1)
    remove = cum <= (1 - top_p)
    remove[..., -1:] = False
    remove = remove.scatter(1, sorted_idx, remove)
    return scores.masked_fill(remove, float("-inf"))
