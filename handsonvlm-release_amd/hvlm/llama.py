"""Vicuna/Llama decoder stack as an explicit forward/backward engine over the HIP kernels.

Replaces transformers' LlamaModel.forward as called at handsonvlm.py:119-128 (inputs_embeds in,
last hidden state out, position_ids = arange(S), causal mask; right padding needs no key mask).

MI355X-first choices:
  * token-major activations [M = B*S, features]; q/k/v live in ONE fused [M, 3D] buffer that the
    attention kernel reads in place and whose gradient buffer feeds the qkv dW/dX GEMMs directly —
    no [B,H,S,dh] permutes anywhere;
  * gate|up fused into one [2F, D] GEMM; residual adds fused into the o_proj / down_proj epilogues;
  * activation recompute is OPTIONAL (cfg.activation_recompute): with 288 GB of HBM3E the ~1.9 GB/layer of saved
    activations (7B, B=16, S=868) fit, so by default backward costs 2x forward instead of the reference's 3x
    (gradient checkpointing, scripts/finetune.sh:36); with it only every layer's input is kept (113 MB/layer) and
    the layer forward is re-run inside backward — what lets 13B at S=2048 reach the reference's 16 per device;
  * packed (unpadded) batches: rows of all samples end to end + cu_seqlens, through the whole decoder;
  * every weight keeps a resident transposed bf16 copy (written by the fused AdamW kernel) so dX
    GEMMs are plain NT GEMMs on K-contiguous operands;
  * weight gradients of one layer are views into ONE flat bf16 bucket (the RCCL all-reduce unit).
"""
from __future__ import annotations

import math
from typing import Callable, Dict, List, Optional

import os

import torch

from . import ops
from .config import HandsOnVLMConfig

BF16 = torch.bfloat16
_FUSED_T = os.environ.get("HVLM_FUSED_T", "1") == "1"   # SwiGLU kernels also write the token-contiguous copies the dW GEMMs need
_DECODE_FUSED_FINALIZE = os.environ.get("HVLM_DECODE_FUSED_FINALIZE", "1") == "1"   # lock-step decode: RoPE + cache write / SwiGLU in the GEMM's finalize pass
_FP8_FUSED_NORM = os.environ.get("HVLM_FP8_FUSED_NORM", "1") == "1"     # 0: RMSNorm in bf16 + a rowwise quantise pass (A/B)
_DW_TN = os.environ.get("HVLM_DW_TN", "0") == "1"      # weight-gradient GEMMs on K-strided operands (A/B switch, see _dw)


def _round_up(x: int, m: int) -> int:
    return (x + m - 1) // m * m


class LayerParams:
    """Fused per-layer parameter views over flat buffers (weights, transposes, grads)."""
    NAMES = ("qkv", "o", "gu", "down", "ln1", "ln2")

    def __init__(self, D: int, F: int, device, with_grad: bool, with_transpose: bool):
        shapes = {"qkv": (3 * D, D), "o": (D, D), "gu": (2 * F, D), "down": (D, F), "ln1": (D,), "ln2": (D,)}
        self.shapes = shapes
        n = sum(math.prod(s) for s in shapes.values())
        self.flat_w = torch.empty(n, dtype=BF16, device=device)
        self.flat_g = torch.zeros(n, dtype=BF16, device=device) if with_grad else None
        self.w: Dict[str, torch.Tensor] = {}
        self.g: Dict[str, torch.Tensor] = {}
        self.wt: Dict[str, torch.Tensor] = {}
        off = 0
        for k, s in shapes.items():
            sz = math.prod(s)
            self.w[k] = self.flat_w[off:off + sz].view(*s)
            if with_grad:
                self.g[k] = self.flat_g[off:off + sz].view(*s)
            off += sz
        if with_transpose:
            for k in ("qkv", "o", "gu", "down"):
                r, c = shapes[k]
                self.wt[k] = torch.empty((c, r), dtype=BF16, device=device)

    def refresh_transposes(self):
        for k, t in self.wt.items():
            ops.transpose(self.w[k], out=t)


class SeqInfo:
    """Row layout of one batch: padded (cu is None: sample b = rows [b*S, (b+1)*S)) or packed (cu int32 [B+1])."""
    __slots__ = ("B", "S", "M", "cu", "pos", "fp8")

    def __init__(self, B, S, M, cu, pos, fp8=False):
        self.B, self.S, self.M, self.cu, self.pos, self.fp8 = B, S, M, cu, pos, fp8


class LlamaStack:
    def __init__(self, cfg: HandsOnVLMConfig, device, training: bool = True):
        self.cfg = cfg
        # activation recompute: False / 0 = keep everything; True = every layer; an int n = only the FIRST n layers (their
        # activations would live longest), the rest keep theirs — spend whatever HBM is left on speed
        ar = getattr(cfg, "activation_recompute", False)
        self.recompute = cfg.num_hidden_layers if ar is True else int(ar or 0)
        self.device = torch.device(device)
        self.training = training
        D, F, L = cfg.hidden_size, cfg.intermediate_size, cfg.num_hidden_layers
        self.D, self.F, self.L, self.H, self.dh = D, F, L, cfg.num_attention_heads, cfg.head_dim
        assert D % 64 == 0 and F % 64 == 0, "hidden/intermediate sizes must be multiples of 64 (GEMM K tile)"
        assert self.dh in (64, 128), "head_dim must be 64 or 128 (attention kernel)"
        self.layers = [LayerParams(D, F, self.device, training, training) for _ in range(L)]
        self.norm_w = torch.ones(D, dtype=BF16, device=self.device)
        self.norm_g = torch.zeros(D, dtype=BF16, device=self.device) if training else None
        # RoPE tables: fp32 cos/sin cast to bf16 like the reference's cos_cached.to(x.dtype)
        inv = 1.0 / (cfg.rope_theta ** (torch.arange(0, self.dh, 2, dtype=torch.float32) / self.dh))
        fr = torch.outer(torch.arange(cfg.max_position_embeddings, dtype=torch.float32), inv)
        self.cos = fr.cos().to(BF16).to(self.device).contiguous()
        self.sin = fr.sin().to(BF16).to(self.device).contiguous()
        self.saved: List[dict] = []
        self._pos_cache: Dict[tuple, torch.Tensor] = {}
        self._dec_tables = None
        self._dec_graph = None
        self.on_layer_grad_ready: Optional[Callable[[int], None]] = None   # DP hook: bucket i is final

    # -- parameters -----------------------------------------------------------------------------
    def invalidate_fp8(self) -> None:
        """Drops the e4m3 copies of the weights (and the decode graphs captured on them): they are made lazily from the bf16
        weights at the first fp8 call, so call this whenever the bf16 weights change (loading a checkpoint, a training step)."""
        self._fp8_rw = None
        self.__dict__.pop("_dec_graphs_b", None)

    def init_random(self, seed: int = 0, std: float = 0.02):
        self.invalidate_fp8()
        gen = torch.Generator(device=self.device).manual_seed(seed)
        for lp in self.layers:
            for k in ("qkv", "o", "gu", "down"):
                lp.w[k].copy_((torch.randn(lp.shapes[k], device=self.device, generator=gen) * std).to(BF16))
            lp.w["ln1"].fill_(1.0); lp.w["ln2"].fill_(1.0)
            lp.refresh_transposes()
        self.norm_w.fill_(1.0)

    def load_hf(self, sd: Dict[str, torch.Tensor], prefix: str = "model."):
        """HF-named Llama weights (model.layers.{i}.self_attn.q_proj.weight, ...) -> fused buffers."""
        self.invalidate_fp8()
        dev = self.device
        g = lambda k: sd[prefix + k].to(device=dev, dtype=BF16)  # noqa: E731
        for i, lp in enumerate(self.layers):
            pre = f"layers.{i}."
            lp.w["qkv"].copy_(torch.cat([g(pre + f"self_attn.{n}_proj.weight") for n in "qkv"], 0))
            lp.w["o"].copy_(g(pre + "self_attn.o_proj.weight"))
            lp.w["gu"].copy_(torch.cat([g(pre + "mlp.gate_proj.weight"), g(pre + "mlp.up_proj.weight")], 0))
            lp.w["down"].copy_(g(pre + "mlp.down_proj.weight"))
            lp.w["ln1"].copy_(g(pre + "input_layernorm.weight"))
            lp.w["ln2"].copy_(g(pre + "post_attention_layernorm.weight"))
            lp.refresh_transposes()
        self.norm_w.copy_(g("norm.weight"))

    def export_hf(self, prefix: str = "model.") -> Dict[str, torch.Tensor]:
        D, F = self.D, self.F
        out = {}
        for i, lp in enumerate(self.layers):
            pre = f"{prefix}layers.{i}."
            q, k, v = lp.w["qkv"].split(D, 0)
            out[pre + "self_attn.q_proj.weight"], out[pre + "self_attn.k_proj.weight"], out[pre + "self_attn.v_proj.weight"] = q, k, v
            out[pre + "self_attn.o_proj.weight"] = lp.w["o"]
            out[pre + "mlp.gate_proj.weight"], out[pre + "mlp.up_proj.weight"] = lp.w["gu"].split(F, 0)
            out[pre + "mlp.down_proj.weight"] = lp.w["down"]
            out[pre + "input_layernorm.weight"] = lp.w["ln1"]
            out[pre + "post_attention_layernorm.weight"] = lp.w["ln2"]
        out[prefix + "norm.weight"] = self.norm_w
        return out

    def _positions(self, B: int, S: int) -> torch.Tensor:
        key = (B, S)
        if key not in self._pos_cache:
            self._pos_cache[key] = torch.arange(S, dtype=torch.int32, device=self.device).repeat(B)
        return self._pos_cache[key]

    # -- forward --------------------------------------------------------------------------------
    def new_cache(self, max_len: Optional[int] = None, batch: Optional[int] = None) -> List[torch.Tensor]:
        """Per-layer [max_len, 3D] buffers in the fused qkv layout (k at column D, v at 2D); with `batch`: [B, max_len, 3D]
        (lock-step generation of B samples)."""
        n = max_len or self.cfg.max_position_embeddings
        if batch is not None:
            kept = getattr(self, "_kv_cache_b", None)
            if kept is None or kept[0].shape[0] != batch or kept[0].shape[1] < n:
                kept = None
                self._kv_cache_b = None                     # free the old one before allocating (22 GB at B = 32)
                self.__dict__.pop("_dec_graphs_b", None)    # (and the decode graph captured on it)
                kept = [torch.empty((batch, n, 3 * self.D), dtype=BF16, device=self.device) for _ in range(self.L)]
                self._kv_cache_b = kept
            return kept
        # kept across samples (evaluation generates one sample after the other: 1.6 GB for 7B at n = 1100; allocating it
        # per sample cost 7.8 ms of an 80 ms sample) — and the decode-step pointer tables keyed on it stay valid
        kept = getattr(self, "_kv_cache", None)
        if kept is None or kept[0].shape[0] < n:
            kept = [torch.empty((n, 3 * self.D), dtype=BF16, device=self.device) for _ in range(self.L)]
            self._kv_cache = kept
        return kept

    def _decode_tables(self, cache):
        import ctypes
        from . import _lib
        lib = _lib.load()
        L = self.L
        if self._dec_tables is None or self._dec_tables[0] is not cache:
            arr = lambda ts: (ctypes.c_void_p * L)(*[t.data_ptr() for t in ts])  # noqa: E731
            lw = lambda k: arr([lp.w[k] for lp in self.layers])  # noqa: E731
            ws = torch.zeros((lib.hvlm_llama_decode_ws_bytes(self.D, self.F),), dtype=torch.uint8, device=self.device)
            self._dec_tables = (cache, lw("qkv"), lw("o"), lw("gu"), lw("down"), lw("ln1"), lw("ln2"), arr(cache), ws,
                                self._positions(1, self.cfg.max_position_embeddings))
            self._dec_graph = None
        return self._dec_tables

    def decode_step(self, x1: torch.Tensor, pos: int, cache: List[torch.Tensor]) -> torch.Tensor:
        """One new token (x1 [1, D]) at position `pos` against the KV cache -> final-normed hidden [1, D].
        HBM-bound (every weight is streamed once by the GEMV kernel).  The 5 launches per layer are issued by one native
        call (hvlm_llama_decode_step), asynchronously and far ahead of the GPU.  HVLM_DECODE_GRAPH=1 captures that call ONCE
        in a hipGraph (hvlm_llama_decode_step_dev: the position lives on the device, so the launch sequence is the same for
        every token) and replays it per token — 162 kernel nodes, one graph launch, bit-identical results.  Measured on
        MI355X (tools/decode_bench.py, 7B, cache length 870): 3.11 ms/token either way — the step is bound by the GPU, not by
        launch issue — so the eager loop stays the default and the graph is for hosts whose CPU is the bottleneck."""
        import ctypes
        from . import _lib
        lib = _lib.load()
        _, tq, to, tg, td, t1, t2, tc, ws, posv = self._decode_tables(cache)
        L = self.L
        if os.environ.get("HVLM_DECODE_GRAPH", "0") == "1":
            st = self._dec_graph
            if st is None:
                st = dict(x=torch.empty((1, self.D), dtype=BF16, device=self.device),
                          hidden=torch.empty((1, self.D), dtype=BF16, device=self.device),
                          pos=torch.zeros(1, dtype=torch.int32, device=self.device), next_pos=-1)

                def issue():
                    rc = lib.hvlm_llama_decode_step_dev(tq, to, tg, td, t1, t2, tc, ops._p(self.norm_w), ops._p(self.cos),
                                                        ops._p(self.sin), ops._p(st["pos"]), ops._p(st["x"]), ops._p(st["hidden"]),
                                                        ops._p(ws), L, self.D, self.F, self.H, self.dh, cache[0].shape[0],
                                                        cache[0].stride(0), self.cfg.rms_norm_eps, ops._stream())
                    _lib.check(rc, "llama_decode_step_dev")

                # eager once (code objects loaded before the capture), on the last cache row, which is put back afterwards
                last = cache[0].shape[0] - 1
                keep = [c[last].clone() for c in cache]
                st["x"].zero_()
                st["pos"].fill_(last)
                issue()
                for c, r in zip(cache, keep):
                    c[last].copy_(r)
                torch.cuda.synchronize()
                g = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g):
                    issue()
                st["g"] = g
                self._dec_graph = st
            st["x"].copy_(x1)
            if st["next_pos"] != pos:                       # new sample (or first use): reset the device-side position
                st["pos"].fill_(pos)
            st["g"].replay()
            st["next_pos"] = pos + 1
            return st["hidden"].clone()
        hidden = torch.empty((1, self.D), dtype=BF16, device=self.device)
        rc = lib.hvlm_llama_decode_step(tq, to, tg, td, t1, t2, tc, ops._p(self.norm_w), ops._p(self.cos), ops._p(self.sin),
                                        ctypes.c_void_p(posv.data_ptr() + 4 * pos), ops._p(x1), ops._p(hidden), ops._p(ws), L,
                                        self.D, self.F, self.H, self.dh, pos, cache[0].stride(0), self.cfg.rms_norm_eps,
                                        ops._stream())
        _lib.check(rc, "llama_decode_step")
        return hidden

    def _layer_forward(self, li: int, x: torch.Tensor, seq: "SeqInfo", save: bool, cache=None, rows=None):
        """One decoder layer.  Returns (output, saved-activation dict or None).
        rows (int64 [n], LAST layer of a training pass only): everything after the attention — o_proj, the MLP — runs on these rows
        only and the output is [n, D]: the other rows' outputs of the last layer feed nothing (see forward(last_rows=...))."""
        lp = self.layers[li]
        eps, scale = self.cfg.rms_norm_eps, self.dh ** -0.5
        mm = (lambda a, k, residual=None: self._mm_fp8(li, k, a, residual)) if seq.fp8 else \
             (lambda a, k, residual=None: ops.gemm_nt(a, lp.w[k], residual=residual))
        probe = getattr(self, "_fp8_probe", None)
        # fp8 inference: RMSNorm writes the e4m3 rows + scales of the next GEMM itself (no bf16 copy, no quantise pass)
        fuse8 = seq.fp8 and not save and probe is None and self.D in (2048, 4096, 5120) and _FP8_FUSED_NORM
        if fuse8:
            xn, rstd1 = ops.rmsnorm_fwd_fp8(x, lp.w["ln1"], eps), None
        else:
            xn, rstd1 = ops.rmsnorm_fwd(x, lp.w["ln1"], eps, save_rstd=save)
        if probe is not None:
            probe[li]["qkv"] = float(xn.abs().max())
        qkv = mm(xn, "qkv")
        ops.rope_qk_(qkv, self.cos, self.sin, seq.pos, self.H, self.dh)
        if cache is not None:
            if cache[li].dim() == 2:                       # batch-1 cache [max_len, 3D] (native decode step)
                assert seq.B == 1
                cache[li][:seq.S].copy_(qkv)
            else:                                          # batched cache [B, max_len, 3D]: only the k | v columns are ever read
                if qkv.is_cuda and seq.cu is None:
                    ops.kv_cache_fill(qkv, cache[li], seq.B, seq.S)
                else:
                    cache[li][:, :seq.S, self.D:].copy_(qkv.view(seq.B, seq.S, -1)[:, :, self.D:])
        ao, lse = ops.attn_fwd(qkv, seq.B, seq.S, self.H, self.dh, True, scale, need_lse=save, cu_seqlens=seq.cu)
        ao_full, x_full, mrows = ao, x, seq.M
        if rows is not None:
            ao, x, mrows = ao.index_select(0, rows), x.index_select(0, rows), int(rows.numel())
        h = mm(ao, "o", residual=x)
        if fuse8:
            hn, rstd2 = ops.rmsnorm_fwd_fp8(h, lp.w["ln2"], eps), None
        else:
            hn, rstd2 = ops.rmsnorm_fwd(h, lp.w["ln2"], eps, save_rstd=save)
        gu = mm(hn, "gu")
        actT = None
        if save and _FUSED_T and self.F % 64 == 0:
            # the token-contiguous copy of act (operand of dW_down) is written by the SwiGLU kernel itself; the row-major
            # act is only needed by down_proj right below and is not kept
            act, actT = ops.swiglu_fwd_t(gu, _round_up(mrows, 64))
        elif fuse8 and self.F % 8 == 0 and self.F <= 14336:
            act = ops.swiglu_fwd_fp8(gu)                  # e4m3 rows + scales: the down-projection's operand as it is
        else:
            act = ops.swiglu_fwd(gu)
        if probe is not None:
            probe[li].update(o=float(ao.abs().max()), gu=float(hn.abs().max()), down=float(act.abs().max()))
        xo = mm(act, "down", residual=h)
        sv = dict(x=x_full, rstd1=rstd1, xn=xn, qkv=qkv, lse=lse, ao=ao_full, h=h, rstd2=rstd2, hn=hn, gu=gu,
                  act=None if actT is not None else act, actT=actT) if save else None
        if sv is not None and rows is not None:
            sv.update(rows=rows, ao_v=ao)
        return xo, sv

    def forward(self, x: torch.Tensor, B: int, S: int, save: Optional[bool] = None,
                all_hidden: bool = False, cache: Optional[List[torch.Tensor]] = None,
                cu_seqlens: Optional[torch.Tensor] = None, pos: Optional[torch.Tensor] = None, fp8: bool = False,
                last_rows: Optional[torch.Tensor] = None):
        """x [M, D] bf16 -> final-normed hidden [M, D]  ([n, D] with last_rows, see below).
        Padded batch: M = B*S, sample b owns rows [b*S, (b+1)*S), positions arange(S) (right padding is harmless under the
        causal mask).  Packed batch (cu_seqlens int32 [B+1] on the device + pos int32 [M]): the B samples are stored end
        to end without padding rows, S = the longest sample — the unpadded contract of the reference's flash-attention
        patch (llama_flash_attn_monkey_patch.py:81-96), kept through the WHOLE decoder here, not only around attention.
        Saves activations for backward when `save` (default: self.training); with self.recompute only each layer's input.
        last_rows (int64 [n] row indices, training passes): the rows whose final hidden state is consumed (lm_head on labelled rows,
        the <hand_traj> gather).  Every row feeds the K / V of every layer, but the LAST layer's o_proj and MLP outputs of the other
        rows feed nothing — zero loss, zero gradient — so that layer gathers the n rows after its attention and the stack returns
        [n, D]; backward() takes d(hidden) [n, D].  The reference (LlamaModel.forward) computes all rows."""
        save = self.training if save is None else save
        cfg = self.cfg
        M = x.shape[0]
        assert x.shape == (M, self.D) and x.dtype == BF16 and S <= cfg.max_position_embeddings
        assert (cu_seqlens is None and M == B * S) or (cu_seqlens is not None and pos is not None and pos.numel() == M)
        assert not (fp8 and save), "the fp8 GEMMs are an inference path (BASELINE config 5)"
        assert last_rows is None or (save and not all_hidden and cache is None), "last_rows: training passes without all_hidden"
        if fp8 and getattr(self, "_fp8_rw", None) is None:
            self.fp8_prepare_rowwise()
        seq = SeqInfo(B=B, S=S, M=M, cu=cu_seqlens, pos=self._positions(B, S) if pos is None else pos, fp8=fp8)
        self.saved = []
        hs = [x] if all_hidden else None
        for li in range(self.L):
            keep_all = save and li >= self.recompute
            xo, sv = self._layer_forward(li, x, seq, keep_all, cache, rows=last_rows if li == self.L - 1 else None)
            if save:
                self.saved.append(sv if keep_all else dict(x=x))
            x = xo
            if all_hidden:
                hs.append(x)
        hidden, rstd_f = ops.rmsnorm_fwd(x, self.norm_w, cfg.rms_norm_eps, save_rstd=save)
        if save:
            # (last_rows travels with the saved state of THIS forward: backward() reads it from here, never from a loose attribute)
            self.final = dict(x=x, rstd=rstd_f, last_rows=last_rows)
            self.seq = seq
        if all_hidden:
            hs[-1] = hidden
            return hidden, hs
        return hidden

    # -- fp8 GEMMs with per-output-channel weight scales and per-token activation scales (BASELINE config 5) -----------
    def fp8_prepare_rowwise(self) -> None:
        """e4m3 copies of the four GEMM weights of every layer, one scale per OUTPUT CHANNEL (row of W); activations are
        quantised per TOKEN at run time (ops.quantize_fp8_rowwise), so nothing has to be calibrated."""
        with torch.no_grad():
            self.__dict__.pop("_dec_graphs_b", None)        # a captured decode step points at the old copies
            self._fp8_rw = [{k: ops.quantize_fp8_rowwise(lp.w[k]) for k in ("qkv", "o", "gu", "down")} for lp in self.layers]

    def _mm_fp8(self, li: int, key: str, a: torch.Tensor, residual=None) -> torch.Tensor:
        w8, inv_w = self._fp8_rw[li][key]
        if not isinstance(a, tuple) and a.shape[0] <= 32 and a.shape[1] % 64 == 0 and w8.shape[0] % 4 == 0 and ops._GEMM_SKINNY:
            # decode steps (M = batch): weight-only e4m3 on the weight-streaming kernel — the cost is reading W once, the
            # activations stay bf16 (no per-token quantisation pass, no activation rounding)
            return ops.gemm_w8a16_nt(a, w8, inv_w, residual=residual)
        a8, inv_a = a if isinstance(a, tuple) else ops.quantize_fp8_rowwise(a)     # RMSNorm hands over e4m3 rows directly
        return ops.gemm_fp8_nt_scaled(a8, w8, inv_a, inv_w, residual=residual)

    def decode_step_batched(self, x1: torch.Tensor, lens: torch.Tensor, cache: List[torch.Tensor], fp8: bool = False) -> torch.Tensor:
        """One new token for each of B samples in lock step: x1 [B, D], lens int32 [B] (device) = position of the new token
        of every sample = its current cache length; cache[l] [B, max_len, 3D].  Returns the final-normed hidden [B, D]
        (valid until the next call).  Weights are streamed once per step for all B samples (GEMMs at M = B).

        The step is ~450 short launches (5 - 80 us each) and the generation loop needs its result on the host before it can
        issue the next one, so issuing them from Python is what bounds it once the kernels are fast (7.4 ms issued eagerly
        against 5 - 6 ms of kernels): the launch sequence is captured ONCE per (batch, cache, weight format) in a hipGraph —
        positions and lengths are device tensors, nothing in it depends on the token index — and replayed per token.
        HVLM_DECODE_BATCH_GRAPH=0 issues it eagerly (bit-identical, tests/test_gpu_fp8.py)."""
        if fp8 and getattr(self, "_fp8_rw", None) is None:
            self.fp8_prepare_rowwise()                  # fp8_decode without an fp8 prefill: the e4m3 copies are made here
        if os.environ.get("HVLM_DECODE_BATCH_GRAPH", "1") != "1" or not x1.is_cuda:
            return self._decode_step_batched_eager(x1, lens, cache, fp8)
        wkey = self._fp8_rw[0]["qkv"][0].data_ptr() if fp8 else self.layers[0].w["qkv"].data_ptr()
        key = (x1.shape[0], bool(fp8), cache[0].data_ptr(), cache[0].shape[1], wkey)       # the graph holds raw pointers
        graphs = self.__dict__.setdefault("_dec_graphs_b", {})
        st = graphs.get(key)
        if st is None:
            # a few graphs may stay alive (bf16 / e4m3 weights on the same cache): tearing one down releases its private memory
            # pool with device-wide synchronising frees — measured 0.4 - 0.6 s, landing in the NEXT batch — so nothing is
            # dropped while switching back and forth; a graph is only discarded with the cache or the weights it points at
            while len(graphs) >= 4:
                graphs.pop(next(iter(graphs)))
            sx, sl = x1.clone(), lens.clone()
            hook = ops._gemm_hook
            ops.set_gemm_hook(None)                     # no event records inside the capture
            try:
                cur = torch.cuda.current_stream(x1.device)
                side = torch.cuda.Stream(device=x1.device)
                side.wait_stream(cur)
                with torch.cuda.stream(side):           # warm-up outside the capture: lazy one-time initialisation in the library
                    self._decode_step_batched_eager(sx, sl, cache, fp8)
                cur.wait_stream(side)
                g = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g):
                    out = self._decode_step_batched_eager(sx, sl, cache, fp8)
            finally:
                ops.set_gemm_hook(hook)
            st = (g, sx, sl, out)
            graphs[key] = st
        g, sx, sl, out = st
        sx.copy_(x1)
        sl.copy_(lens)
        g.replay()
        return out

    def _decode_step_batched_eager(self, x1: torch.Tensor, lens: torch.Tensor, cache: List[torch.Tensor], fp8: bool = False) -> torch.Tensor:
        if fp8 and getattr(self, "_fp8_rw", None) is None:
            self.fp8_prepare_rowwise()
        B = x1.shape[0]
        eps, scale = self.cfg.rms_norm_eps, self.dh ** -0.5
        rows = torch.arange(B, device=self.device)
        idx = lens.long()
        lens1 = (lens + 1).contiguous()
        x = x1
        # q|k|v and gate|up: RoPE + cache row write / SwiGLU ride in the split-K finalize pass of the weight-streaming GEMM
        # (bit-identical, three launches fewer per layer); HVLM_DECODE_FUSED_FINALIZE=0 keeps them as separate kernels
        fused = (_DECODE_FUSED_FINALIZE and ops._GEMM_SKINNY and x1.is_cuda and B <= 32 and self.D % 64 == 0 and self.F % 4 == 0
                 and self.D >= 256 and self.dh % 8 == 0)
        for li, lp in enumerate(self.layers):
            mm = (lambda a, k, residual=None: self._mm_fp8(li, k, a, residual)) if fp8 else \
                 (lambda a, k, residual=None: ops.gemm_nt(a, lp.w[k], residual=residual))
            wsel = (lambda k: self._fp8_rw[li][k]) if fp8 else (lambda k: lp.w[k])
            xn, _ = ops.rmsnorm_fwd(x, lp.w["ln1"], eps, save_rstd=False)
            if fused:
                qkv = ops.gemm_skinny_qkv(xn, wsel("qkv"), cache[li], lens, self.cos, self.sin, self.H, self.dh)
            else:
                qkv = mm(xn, "qkv")
                ops.rope_qk_(qkv, self.cos, self.sin, lens, self.H, self.dh)
                cache[li][rows, idx] = qkv                               # the new token's k / v row of every sample
            ao = ops.attn_decode_batched(qkv, cache[li], lens1, self.H, self.dh, scale)
            h = mm(ao, "o", residual=x)
            hn, _ = ops.rmsnorm_fwd(h, lp.w["ln2"], eps, save_rstd=False)
            act = ops.gemm_skinny_swiglu(hn, wsel("gu")) if fused else ops.swiglu_fwd(mm(hn, "gu"))
            x = mm(act, "down", residual=h)
        hidden, _ = ops.rmsnorm_fwd(x, self.norm_w, eps, save_rstd=False)
        return hidden

    # -- fp8 inference forward, per-tensor static scales (round-1 variant, kept for A/B) -------------------------------
    FP8_MAX = 448.0

    def fp8_prepare(self, x: torch.Tensor, B: int, S: int, margin: float = 1.25) -> None:
        """Quantises the four GEMM weights of every layer to e4m3 with per-tensor scales (once) and calibrates static
        per-tensor ACTIVATION scales from one bf16 forward of `x` (amax * margin): no amax pass at run time."""
        with torch.no_grad():
            self._fp8_w = []
            for lp in self.layers:
                ent = {}
                for k in ("qkv", "o", "gu", "down"):
                    w = lp.w[k]
                    sw = self.FP8_MAX / max(float(w.abs().max()), 1e-12)
                    ent[k] = (ops.quantize_fp8(w.contiguous(), sw), sw)
                self._fp8_w.append(ent)
            amax = [dict() for _ in self.layers]
            self._fp8_probe = amax
            self.forward(x, B, S, save=False)
            self._fp8_probe = None
            self._fp8_a = [{k: self.FP8_MAX / max(v * margin, 1e-12) for k, v in d.items()} for d in amax]

    def forward_fp8(self, x: torch.Tensor, B: int, S: int) -> torch.Tensor:
        """Same function as forward() with the QKV / O / gate|up / down GEMMs on e4m3 operands (scaled MFMA, fp32
        accumulation, bf16 outputs); norms, RoPE, attention, SwiGLU and the residual stream stay bf16."""
        assert getattr(self, "_fp8_w", None) is not None, "call fp8_prepare() first"
        pos = self._positions(B, S)
        scale = self.dh ** -0.5
        eps = self.cfg.rms_norm_eps
        for lp, w8, sa in zip(self.layers, self._fp8_w, self._fp8_a):
            def mm(inp, key, residual=None):
                q = ops.quantize_fp8(inp, sa[key])
                return ops.gemm_fp8_nt(q, w8[key][0], 1.0 / (sa[key] * w8[key][1]), residual=residual)
            xn, _ = ops.rmsnorm_fwd(x, lp.w["ln1"], eps, save_rstd=False)
            qkv = mm(xn, "qkv")
            ops.rope_qk_(qkv, self.cos, self.sin, pos, self.H, self.dh)
            ao, _ = ops.attn_fwd(qkv, B, S, self.H, self.dh, True, scale, need_lse=False)
            h = mm(ao, "o", residual=x)
            hn, _ = ops.rmsnorm_fwd(h, lp.w["ln2"], eps, save_rstd=False)
            act = ops.swiglu_fwd(mm(hn, "gu"))
            x = mm(act, "down", residual=h)
        hidden, _ = ops.rmsnorm_fwd(x, self.norm_w, eps, save_rstd=False)
        return hidden

    # -- backward -------------------------------------------------------------------------------
    def _dw(self, dy: torch.Tensor, xin: torch.Tensor, out: torch.Tensor, mpad: int, dyT=None, xT=None):
        """out[N,K] = dy[M,N]^T @ xin[M,K]  via K-contiguous transposed copies (zero-padded to mpad).

        HVLM_DW_SIDE_STREAM=1 issues it on a side stream: the weight gradient is independent of the activation-gradient
        GEMM that follows on the main stream, and the kernel dispatched second can fill the CUs the first one frees in
        its last, partial round of tiles (880 tiles on 256 CUs = 3.44 rounds).  Measured: 0 - 0.7 % on the SFT step (two
        persistent one-workgroup-per-CU kernels mostly just take turns), so it is OFF by default."""
        side = self._side_stream()
        if _DW_TN and dyT is None and xT is None and dy.stride(1) == 1 and xin.stride(1) == 1:
            # dW = dY^T X straight from the token-major activations: the K-strided (TN) instantiation of the ping-pong GEMM
            # gathers its fragments with LDS transpose reads — no transposed copies, no HBM-bound transpose kernels
            ops.gemm(dy, xin, trans_a=True, trans_b=True, out=out)
            return
        if side is None:
            dyT = ops.transpose(dy, r_pad=mpad) if dyT is None else dyT      # (already written by the producing kernel)
            xT = ops.transpose(xin, r_pad=mpad) if xT is None else xT
            ops.gemm_nt(dyT, xT, out=out)
            return
        main = torch.cuda.current_stream()
        side.wait_stream(main)                       # dy / xin are produced on the main stream
        with torch.cuda.stream(side):
            dyT = ops.transpose(dy, r_pad=mpad) if dyT is None else dyT
            xT = ops.transpose(xin, r_pad=mpad) if xT is None else xT
            ops.gemm_nt(dyT, xT, out=out)
        for t_ in (dy, xin, dyT, xT):
            if t_ is not None:
                t_.record_stream(side)

    def _side_stream(self):
        if os.environ.get("HVLM_DW_SIDE_STREAM", "0") != "1":
            return None
        if getattr(self, "_side", None) is None:
            self._side = torch.cuda.Stream()
        return self._side

    def _join_side(self):
        """Main stream waits for the weight gradients issued so far (before they are reduced / consumed)."""
        if getattr(self, "_side", None) is not None:
            torch.cuda.current_stream().wait_stream(self._side)

    def backward(self, dhidden: torch.Tensor) -> torch.Tensor:
        """dhidden [M, D] (grad of the final-normed hidden) -> dx [M, D]; weight grads are written
        (not accumulated) into the per-layer buckets."""
        assert self.training and self.saved, "forward(save=True) must precede backward"
        seq = self.seq
        B, S, M, pos = seq.B, seq.S, seq.M, seq.pos
        mpad_full = _round_up(M, 64)
        scale = self.dh ** -0.5
        last_rows = self.final.get("last_rows")
        n_want = M if last_rows is None else int(last_rows.numel())
        assert dhidden.shape == (n_want, self.D), \
            (f"backward: d(hidden) has {tuple(dhidden.shape)}, the forward returned [{n_want}, {self.D}]"
             + (" (last_rows: the stack's last layer ran on the labelled rows only)" if last_rows is not None else ""))
        dx, _ = ops.rmsnorm_bwd(dhidden, self.final["x"], self.norm_w, self.final["rstd"], dw_out=self.norm_g)
        for i in range(self.L - 1, -1, -1):
            lp, sv = self.layers[i], self.saved[i]
            if i < self.recompute:
                # activation recompute (the reference's gradient checkpointing, scripts/finetune.sh:36): only the layer input
                # was kept; its forward is run again here, bit-identical to the first one (deterministic kernels)
                sv = self._layer_forward(i, sv["x"], seq, True, rows=last_rows if i == self.L - 1 else None)[1]
            rows = sv.get("rows")                        # last layer with last_rows: dx, the MLP and o_proj live on [n, D]
            mpad = mpad_full if rows is None else _round_up(int(rows.numel()), 64)
            # ---- MLP ----
            dact = ops.gemm_nt(dx, lp.wt["down"])                      # [M,F] = dx @ Wd
            self._dw(dx, sv["act"], lp.g["down"], mpad, xT=sv.get("actT"))   # dWd[D,F]
            dguT = None
            if _FUSED_T and self.F % 64 == 0:
                dgu, dguT = ops.swiglu_bwd_t(sv["gu"], dact, mpad)
            else:
                dgu = ops.swiglu_bwd(sv["gu"], dact)
            del dact
            dhn = ops.gemm_nt(dgu, lp.wt["gu"])                        # [M,D]
            self._dw(dgu, sv["hn"], lp.g["gu"], mpad, dyT=dguT)
            del dgu, dguT
            dh, _ = ops.rmsnorm_bwd(dhn, sv["h"], lp.w["ln2"], sv["rstd2"], add=dx, dw_out=lp.g["ln2"])
            # ---- attention ----
            dao = ops.gemm_nt(dh, lp.wt["o"])
            self._dw(dh, sv["ao"] if rows is None else sv["ao_v"], lp.g["o"], mpad)
            if rows is not None:                         # back to all M rows: the other rows' d(attention output) and residual
                dao_v, dh_v = dao, dh                    # gradient are exactly zero
                dao = torch.zeros((M, dao_v.shape[1]), dtype=dao_v.dtype, device=dao_v.device).index_copy_(0, rows, dao_v)
                dh = torch.zeros((M, dh_v.shape[1]), dtype=dh_v.dtype, device=dh_v.device).index_copy_(0, rows, dh_v)
                del dao_v, dh_v
                mpad = mpad_full
            # (the inverse rotary of dq / dk rides in the attention backward's epilogues: no separate pass over dqkv)
            dqkv = ops.attn_bwd(sv["qkv"], sv["ao"], dao, sv["lse"], B, S, self.H, self.dh, True, scale, cu_seqlens=seq.cu,
                                rope=(self.cos, self.sin, pos))
            dxn = ops.gemm_nt(dqkv, lp.wt["qkv"])
            self._dw(dqkv, sv["xn"], lp.g["qkv"], mpad)
            del dqkv, dao
            dx, _ = ops.rmsnorm_bwd(dxn, sv["x"], lp.w["ln1"], sv["rstd1"], add=dh, dw_out=lp.g["ln1"])
            self.saved[i] = None
            del sv
            self._join_side()
            if self.on_layer_grad_ready is not None:
                self.on_layer_grad_ready(i)
        self.saved = []
        self.final = None
        return dx

    # -- bookkeeping ----------------------------------------------------------------------------
    def gemm_params_per_layer(self) -> int:
        return 4 * self.D * self.D + 3 * self.D * self.F

    def flops_forward(self, B: int, S: int, vocab: int) -> float:
        """Algorithmic forward FLOPs (2/MAC, causal attention counted at half)."""
        return B * (2.0 * S * (self.L * self.gemm_params_per_layer() + self.D * vocab) + 0.5 * 4.0 * S * S * self.D * self.L)
