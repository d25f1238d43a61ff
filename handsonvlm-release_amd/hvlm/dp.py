"""Data-parallel gradient exchange: one process per GPU, `torch.distributed` (backend "nccl" = RCCL
over xGMI on ROCm; "gloo" on CPU for tests).

The path shards by sample: forward/backward need no communication; the only exchange
is the gradient all-reduce.  Each decoder layer's weight gradients live in ONE flat bf16 bucket
(~405 MB for Vicuna-7B), and its all-reduce is launched asynchronously the moment that layer's
backward finishes, so all but the last buckets overlap with the remaining backward compute (RCCL runs
on its own stream; the collective is ordered after the kernels already enqueued on the compute
stream).  The SUM is divided by world size inside the optimizer step (AdamW grad_scale), which
reproduces the reference's DeepSpeed gradient averaging.  No ZeRO, no parameter sharding."""
from __future__ import annotations

import os
from typing import List, Optional

import torch
import torch.distributed as dist


def init_from_env(backend: Optional[str] = None):
    """Reads RANK/LOCAL_RANK/WORLD_SIZE/MASTER_* (torch.distributed.run).  Returns (rank, local_rank, world)."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    force = os.environ.get("HVLM_FORCE_DIST", "0") == "1"      # exercise the RCCL path even at world size 1
    if (world > 1 or force) and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        if backend is None:
            # HVLM_DIST_BACKEND=gloo: ranks that share one device (functional checks on a 1-GPU box; RCCL refuses that)
            backend = os.environ.get("HVLM_DIST_BACKEND") or ("nccl" if torch.cuda.is_available() else "gloo")
        if backend == "nccl":
            torch.cuda.set_device(local)
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, local, world


def check_topology(group=None, verbose: bool = True) -> dict:
    """First-step readiness check of an N > 1 run: every rank must drive its OWN GPU.  Two ranks that landed
    on one device (a launcher that does not set LOCAL_RANK, HIP_VISIBLE_DEVICES hiding all but one GPU, ...) still produce a
    "working" run — RCCL refuses it, but a gloo fall-back or HVLM_DIST_BACKEND=gloo does not — whose samples/s says nothing
    about xGMI scaling.  Gathers (host, PCI bus id, device name) from every rank, asserts the (host, bus id) pairs are distinct
    and has rank 0 print ONE line with the world size, the backend, the RCCL version and the device map.  Shared-device runs are
    legitimate only as functional checks and must say so: HVLM_DIST_BACKEND=gloo skips the assertion (the line still prints
    "SHARED")."""
    import socket
    if not dist.is_initialized():
        return {"world": 1, "distinct": True}
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    if torch.cuda.is_available():
        dev = torch.cuda.current_device()
        pr = torch.cuda.get_device_properties(dev)
        bus = getattr(pr, "pci_bus_id", None)
        ident = f"{getattr(pr, 'pci_domain_id', 0):04x}:{bus:02x}:{getattr(pr, 'pci_device_id', 0):02x}" if bus is not None else str(getattr(pr, "uuid", dev))
        mine = (socket.gethostname(), ident, pr.name, dev)
    else:
        mine = (socket.gethostname(), f"cpu-rank{rank}", "cpu", -1)
    allr = [None] * world
    dist.all_gather_object(allr, mine, group=group)
    distinct = len({(h, i) for h, i, _, _ in allr}) == world
    backend = dist.get_backend(group)
    shared_ok = os.environ.get("HVLM_DIST_BACKEND") == "gloo" or not torch.cuda.is_available()
    try:
        ver = ".".join(str(v) for v in torch.cuda.nccl.version()) if backend == "nccl" else "-"
    except Exception:                                         # noqa: BLE001  (a build without the nccl bindings)
        ver = "?"
    if rank == 0 and verbose:
        print(f"hvlm.dp: world {world} backend {backend} (RCCL {ver}) devices "
              + " ".join(f"r{r}={h}/{i}[{d}]" for r, (h, i, _, d) in enumerate(allr))
              + (" DISTINCT" if distinct else " SHARED"), flush=True)
    assert distinct or shared_ok, \
        f"data-parallel ranks share a GPU: {allr} — one process per GPU is the contract (set LOCAL_RANK / do not hide devices)"
    return {"world": world, "distinct": distinct, "backend": backend, "devices": allr}


class GradBuckets:
    """Flat gradient buckets + asynchronous all-reduce handles."""

    def __init__(self, group=None):
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.active = dist.is_initialized() and (self.world > 1 or os.environ.get("HVLM_FORCE_DIST", "0") == "1")
        self.handles: List = []
        self.post: List = []               # callables run once after wait() (see EmbedRows)
        # communication accounting of the CURRENT step (bench.py's `comm` object): collectives issued, bytes handed to them,
        # and the event recorded on the compute stream when the last backward kernel had been enqueued (the "tail" hook)
        self.n_collectives = 0
        self.bytes = 0
        self.backward_end_event = None

    def begin_step(self):
        self.n_collectives, self.bytes, self.backward_end_event = 0, 0, None

    def mark_backward_end(self):
        """Called by the tail hook BEFORE it issues the last all-reduces: everything after this event on the step's critical
        path (tail-bucket all-reduce, the updates that wait for it, whatever earlier all-reduces are still in flight) is
        exposed, i.e. not hidden under backward compute."""
        if torch.cuda.is_available():
            self.backward_end_event = torch.cuda.Event(enable_timing=True)
            self.backward_end_event.record()

    def reduce_async(self, *tensors: torch.Tensor) -> List:
        """Starts the all-reduces; returns their work handles (empty when not distributed)."""
        if not self.active:
            return []
        hs = [dist.all_reduce(t, op=dist.ReduceOp.SUM, group=self.group, async_op=True) for t in tensors]
        self.handles.extend(hs)
        self.n_collectives += len(tensors)
        self.bytes += sum(t.numel() * t.element_size() for t in tensors)
        return hs

    def wait(self):
        for h in self.handles:
            h.wait()
        self.handles = []
        for fn in self.post:               # work that has to follow a collective (EmbedRows.finish); each runs once
            fn()
        self.post = []


class EmbedRows:
    """All-reduce of the text-embedding gradient restricted to the rows that can be non-zero.

    `embed_tokens.weight.grad` is [32101, D] bf16 (263 MB for 7B, 329 MB for 13B) and it is the LAST gradient of the step to
    become final (it hangs off the input of layer 0), so its all-reduce cannot hide under any backward compute.  Only the rows
    of tokens that occur in some rank's minibatch are non-zero (model.backward zeroes the buffer and writes each occurring
    token's row once, hvlm_embed_grad_segments): at most B*T per rank, typically far fewer distinct ones for natural text.
      prepare(ids)  at the START of the step (model.forward calls it with this rank's distinct token ids before the CLIP tower
                    is enqueued): a V-entry occupancy mask is summed over the ranks — 128 KB, and the one host synchronisation
                    it needs falls where the device still has the previous step's tail to run — giving every rank the same
                    sorted union of touched rows;
      reduce(g)     at the tail hook: the union rows are gathered into a compact [n, D] buffer, that buffer is all-reduced, and
      finish()      copies it back (on the stream that waited for the collective).  Rows outside the union are zero on every
                    rank, so the result equals the dense all-reduce bit for bit.  Falls back to the dense all-reduce when the
                    union covers more than half of the vocabulary (uniform synthetic ids at 8 ranks: ~86 %).
    OPT-IN (attach(embed_rows=True) / HVLM_DP_EMBED_ROWS=1): prepare() costs one host synchronisation per step, and whether that is
    cheaper than ~1.5 ms of dense 263-MB all-reduce over xGMI has not been measured on an N > 1 RCCL box yet — until it has, the
    dense all-reduce is the default."""

    def __init__(self, buckets: "GradBuckets", vocab: int):
        self.buckets, self.vocab = buckets, vocab
        self.idx = None
        self._pending = None
        self.rows_last = None              # accounting: rows reduced in the last step (None: dense)

    def prepare(self, token_ids: torch.Tensor):
        if not self.buckets.active:
            self.idx, self.rows_last = None, None
            return
        # the blocking all-reduce below must not be issued between the async bucket all-reduces of a step that is still open (the
        # ranks would disagree on the collective order): every step ends with buckets.wait(), which empties `handles`
        assert not self.buckets.handles and self._pending is None, \
            "EmbedRows.prepare: the previous step's all-reduces were not waited for (GradBuckets.wait())"
        self.idx, self.rows_last = None, None
        mask = torch.zeros(self.vocab, dtype=torch.int32, device=token_ids.device)
        mask[token_ids.long()] = 1
        dist.all_reduce(mask, op=dist.ReduceOp.SUM, group=self.buckets.group)
        idx = mask.nonzero().flatten()                     # (host synchronisation: see the class comment)
        if 2 * idx.numel() <= self.vocab:
            self.idx = idx
            self.rows_last = int(idx.numel())

    def reduce(self, embed_g: torch.Tensor) -> List:
        if self.idx is None or embed_g.dim() != 2:
            return self.buckets.reduce_async(embed_g)
        compact = embed_g.index_select(0, self.idx)
        hs = self.buckets.reduce_async(compact)
        self._pending = (embed_g, self.idx, compact)
        self.buckets.post.append(self.finish)
        return hs

    def finish(self):
        if self._pending is not None:
            embed_g, idx, compact = self._pending
            self._pending = None
            embed_g.index_copy_(0, idx, compact)
            if compact.is_cuda:
                # `compact` was allocated on the compute stream; with the overlapped optimizer this copy runs on the optimizer's
                # stream: tell the caching allocator, or the block could be handed out again before the copy has executed
                compact.record_stream(torch.cuda.current_stream())


def attach(model, buckets: GradBuckets, opt=None, embed_rows: Optional[bool] = None):
    """Wires the model's backward hooks to the buckets: layer i -> its flat bucket; 'tail' -> embed, lm_head,
    projector, final norm and the CVAE head's PyTorch gradients.  With an overlapped optimizer (`opt`, optim.AdamW
    with overlap=True) every bucket is also handed to it the moment it is final, ordered after its all-reduce.
    embed_rows: all-reduce the embedding gradient over the union of touched rows (EmbedRows; default: HVLM_DP_EMBED_ROWS=1,
    else the dense buffer)."""
    stack = model.stack
    ready = opt.bucket_ready if opt is not None else (lambda key, after=(), post=None: None)
    rows = None
    if embed_rows is None:
        embed_rows = os.environ.get("HVLM_DP_EMBED_ROWS", "0") == "1"
    if embed_rows and hasattr(model, "on_token_ids") and getattr(model.embed_g, "dim", lambda: 0)() == 2:
        rows = EmbedRows(buckets, model.embed_g.shape[0])
        model.on_token_ids = rows.prepare
    buckets.embed_rows = rows

    def on_layer(i: int):
        ready(f"layers.{i}", buckets.reduce_async(stack.layers[i].flat_g))

    def on_named(which: str):
        if which == "head":
            # lm_head's gradient (263 MB for 7B) is final before the decoder backward starts: reducing it here hides it
            # under the whole backward instead of leaving it exposed at the end of the step.
            ready("head", buckets.reduce_async(model.lm_head_g))
        if which == "tail":
            buckets.mark_backward_end()
            hs = buckets.reduce_async(stack.norm_g, model.proj_w_g, model.proj_b_g)
            hs += rows.reduce(model.embed_g) if rows is not None else buckets.reduce_async(model.embed_g)
            for p in model.hand_traj_decoder.parameters():
                if p.grad is not None:
                    buckets.reduce_async(p.grad)
            ready("tail", hs, post=rows.finish if rows is not None else None)

    stack.on_layer_grad_ready = on_layer
    model.on_grads_ready = on_named
    return buckets
