"""The host -> device edge of the SFT step: batch i + 1 is uploaded while step i computes.

The reference's collator hands the model a CPU batch whose `image` is 100 frames x 3 x 224 x 224 bf16 per sample — 482 MB per
step per GPU at the per-device batch of 16 (handsonvlm/dataset/hybrid_dataset.py:151-168; `--dataloader_num_workers 4`,
scripts/finetune.sh:37) — and HF's Trainer moves it with a blocking `.to(device)` from pageable memory.  Here:

  * a dedicated copy stream and two sets of PINNED host buffers / device buffers: the upload of batch i + 1 is enqueued right after
    step i has been enqueued on the compute stream and runs beside it; the compute stream only waits on the copy's event
    (`h2d_wait_ms` measures what is left of that wait — the exposed part);
  * the EK100 reader tiles 10 distinct frames 10 x into the 100 the model asserts (handsonvlm/dataset/epic_dataset.py:90-95).  When a
    sample has that structure — checked bit for bit on the host — only the 10 distinct frames and the [B * 100] frame index cross
    the bus (48 MB instead of 482), and the model gets them as `image_frames` (HandsOnVLMForCausalLM.forward), which also
    spares the device-side hash + verify of `config.dedup_frames`.  Results are identical either way: the tower's output for a
    frame does not depend on which other frames share its launch (tests/test_gpu_r5.py::test_prefetch_bit_identical);
  * everything else in the batch (ids, labels, masks, future_hands ...) stays where the model wants it: the splice plan is host
    work on the ids, the small float tensors are a few hundred bytes.

`DevicePrefetcher(loader, device)` is an iterator over the loader's batches with `image` replaced (device tensor, or None +
`image_frames`).  On a CPU device it passes the batches through unchanged (host-logic tests).
"""
from __future__ import annotations

from typing import Dict, Iterable, Iterator, Optional

import torch


def detect_tiling(image: torch.Tensor, period: int = 10) -> bool:
    """True iff every sample's frames are `period` distinct-position frames repeated: image[b, r * period + f] == image[b, f]
    for all r (the layout of tile_epic_frames / epic_dataset.py:90-95).  Exact comparison (memcmp speed, host)."""
    if image.dim() != 5 or image.shape[1] % period != 0 or image.shape[1] == period:
        return False
    B, T = image.shape[:2]
    v = image.reshape(B, T // period, period, *image.shape[2:])
    first = v[:, :1]
    # compare as integers: bit-for-bit (NaN payloads, -0.0) like the device-side verification of visual_tokens
    it = torch.int16 if image.element_size() == 2 else (torch.int32 if image.element_size() == 4 else None)
    if it is None:
        return False
    ref = first[:, 0].reshape(-1).view(it)
    return all(torch.equal(v[:, r].reshape(-1).view(it), ref) for r in range(1, T // period))


class DevicePrefetcher:
    def __init__(self, loader: Iterable[Dict], device, dedup_period: Optional[int] = 10, depth: int = 2):
        self.loader, self.device = loader, torch.device(device)
        self.period = dedup_period
        self.cuda = self.device.type == "cuda"
        self.depth = depth
        self.copy_stream = torch.cuda.Stream(self.device) if self.cuda else None
        self._host = [None] * depth            # pinned staging buffers (frames), grown on demand
        self._dev = [None] * depth             # device buffers (frames)
        self._idx_host = [None] * depth
        self._copied = [None] * depth          # event: the upload from host set k has completed (host buffer reusable)
        self._consumed = [None] * depth        # event on the compute stream: the step that used device set k has been enqueued
        self.h2d_wait_ms = []                  # per delivered batch: compute-stream time spent waiting for the upload
        self.bytes_last = 0
        self.tiled_last = False                # whether the last staged batch was recognised as a period-tiled clip
        self._wait_events = []

    # -- staging ----------------------------------------------------------------------------------------------------
    def _buf(self, store, k, shape, dtype, pinned):
        n = 1
        for s in shape: n *= int(s)
        b = store[k]
        if b is None or b.numel() < n or b.dtype != dtype:
            if pinned:
                b = torch.empty(n, dtype=dtype, pin_memory=True)
            else:
                b = self._dev_alloc(n, dtype)
            store[k] = b
        return b[:n].view(*shape)

    def _dev_alloc(self, n, dtype):
        """Device staging memory comes from the COPY stream's pool.  The caching allocator hands a freed block back to the stream
        it was allocated on without waiting for that stream's queued kernels; a block taken from the compute stream's pool
        could still be read or written by the in-flight step when the copy stream's DMA first touches it.  A block of the
        copy stream's pool was last used by the copy stream (or by a compute-stream reader the allocator already waits for
        through `record_stream`), so its first touch needs no further ordering."""
        with torch.cuda.stream(self.copy_stream):
            return torch.empty(n, dtype=dtype, device=self.device)

    def _stage(self, batch: Dict, k: int) -> Dict:
        img = batch["image"]
        out = dict(batch)
        if not self.cuda or not torch.is_tensor(img) or img.is_cuda:
            return out
        tiled = self.period is not None and detect_tiling(img, self.period)
        src = img[:, : self.period] if tiled else img
        if self._copied[k] is not None:
            self._copied[k].synchronize()                       # the copy that last read this pinned buffer (two batches ago)
        host = self._buf(self._host, k, src.shape, src.dtype, pinned=True)
        host.copy_(src)                                         # pageable -> pinned (host memcpy; the DataLoader's pin thread would do the same)
        dev = self._buf(self._dev, k, src.shape, src.dtype, pinned=False)
        cur = torch.cuda.current_stream(self.device)
        with torch.cuda.stream(self.copy_stream):
            if self._consumed[k] is not None:
                self.copy_stream.wait_event(self._consumed[k])  # the step that read this device buffer has been enqueued and must finish
            dev.copy_(host, non_blocking=True)
            if tiled:
                B, T = img.shape[:2]
                idx = (torch.arange(B, dtype=torch.int32).view(B, 1) * self.period
                       + (torch.arange(T, dtype=torch.int32) % self.period).view(1, T)).reshape(-1)
                ih = idx.pin_memory()                           # 4 * B * T bytes
                idev = torch.empty(B * T, dtype=torch.int32, device=self.device)   # (copy stream current: its pool)
                idev.copy_(ih, non_blocking=True)
                self._idx_host[k] = ih                          # (kept alive until this slot is staged again)
            ev = torch.cuda.Event()
            ev.record(self.copy_stream)
        self._copied[k] = ev
        dev.record_stream(cur)
        self.bytes_last = host.numel() * host.element_size()
        self.tiled_last = bool(tiled)
        if tiled:
            idev.record_stream(cur)
            out["image"] = None
            out["image_frames"] = (dev.view(-1, *img.shape[2:]), idev, (int(img.shape[0]), int(img.shape[1])))
        else:
            out["image"] = dev
        out["_h2d_event"] = ev
        return out

    def _deliver(self, staged: Dict, k: int) -> Dict:
        ev = staged.pop("_h2d_event", None)
        if ev is not None:
            cur = torch.cuda.current_stream(self.device)
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record(cur)
            cur.wait_event(ev)
            b.record(cur)
            self._wait_events.append((a, b))
        return staged

    def _mark_consumed(self, k: int):
        if self.cuda:
            ev = torch.cuda.Event()
            ev.record(torch.cuda.current_stream(self.device))
            self._consumed[k] = ev

    def __iter__(self) -> Iterator[Dict]:
        it = iter(self.loader)
        k = 0
        try:
            nxt = self._stage(next(it), k)
        except StopIteration:
            return
        while nxt is not None:
            cur, ck = nxt, k
            batch = self._deliver(cur, ck)
            yield batch                                         # the consumer enqueues step i ...
            self._mark_consumed(ck)                             # ... and only then is batch i + 1 staged: its upload runs beside step i
            k = (k + 1) % self.depth
            try:
                nxt = self._stage(next(it), k)
            except StopIteration:
                nxt = None

    def __len__(self):
        return len(self.loader)

    def collect_wait_ms(self):
        """Resolves the recorded wait events (call after a synchronize): exposed upload time per delivered batch, ms."""
        for a, b in self._wait_events:
            self.h2d_wait_ms.append(a.elapsed_time(b))
        self._wait_events = []
        return self.h2d_wait_ms
