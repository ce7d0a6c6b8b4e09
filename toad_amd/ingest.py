"""Bag ingest: slide records -> device-resident bags, overlapped with compute.

Reference behaviour being replaced: ``Generic_MIL_MTL_Dataset.__getitem__`` does ``torch.load(<slide_id>.pt)``
(datasets/dataset_mtl_concat.py:369-373), the loader collates one bag per batch (utils/utils.py:30-35,51-55) and
the train loop copies it to the device synchronously (utils/core_utils_mtl_concat.py:201-204). A 100k-patch bag is
410 MB: >= 6.5 ms over PCIe Gen5 x16, more than the 3.6 ms the kernels need for the whole step.

``BagPrefetcher`` keeps ``depth`` bags in flight: worker threads read and pin them, the host->device copy runs on a
dedicated HIP stream, and the consumer's stream only waits on the copy's event. Features stored as fp16/bf16 are copied
at half the bytes and upcast on the device. The wire format stays the reference's: one ``torch.save``d ``[N, 1024]``
tensor per slide.
"""
from __future__ import annotations

import queue
import threading
from concurrent.futures import ThreadPoolExecutor
from typing import Callable, Iterable, Iterator, Optional, Sequence, Tuple, Union

import torch

Record = Tuple[Union[str, Callable[[], torch.Tensor], torch.Tensor], int, int, float]   # (source, label, site, sex)


def _load(source) -> torch.Tensor:
    if isinstance(source, torch.Tensor):
        t = source
    elif callable(source):
        t = source()
    else:
        t = torch.load(source, map_location="cpu")            # the reference's .pt bag
    if t.dim() != 2:
        raise ValueError(f"bag must be [N, features], got {tuple(t.shape)}")
    return t.contiguous()


class BagPrefetcher:
    """Iterate ``(bag, label, site, sex)`` device tensors in record order with ``depth`` bags in flight."""

    def __init__(self, records: Sequence[Record], device: Union[str, torch.device], depth: int = 2, workers: int = 2,
                 dtype: Optional[torch.dtype] = torch.float32, prepare: bool = False, arena_rows: int = 0):
        """``arena_rows`` > 0: consecutive fp32 bags are landed BACK TO BACK in device buffers of that many rows (a new buffer when the next bag
        does not fit; a bag longer than the buffer gets an allocation of its own) and handed out as views. Bags that share a buffer are their own
        concatenation, so ``SlideShardedDP`` / ``ops.mil_multi_step`` batch them into one ragged multi-slide call without copying a row
        (``ops._adjacent_rows``); set it to the DP wrapper's ``batch_rows``. A buffer is released when the last view of it dies - so ONE retained
        bag keeps its whole landing buffer alive (``arena_rows`` x features x 4 bytes: 2 GB at 524,288 rows of 1,024 features); hold copies, not
        views, of bags that must outlive their step. One buffer serves one feature width: a bag of another width starts a fresh buffer.
        ``prepare``: hand the consumer ``ops.PreparedBag`` objects instead of fp32 tensors (toad_bag_prepare_f32, ABI 9): right behind
        its host-to-device copy, on the COPY stream, every bag is brought into the plane-tiled two-piece form the first Linear and its
        weight gradient take by LDS-DMA, and the fp32 copy is released. The training stream then never measures or splits the bag
        (-3 % of a 100k-patch step, more on boxes where the abs-max pass is slower); bags below 64 patches stay fp32 tensors.
        ``dtype``: what the consumer receives. ``torch.float32`` (default) up-casts fp16 / bf16 files on the device;
        ``torch.float16`` hands fp16 bags to the model as they are (its first Linear and weight gradient then run the two-term
        fp16 kernels, toad_mil_*_x16_f32: no up-cast pass, half the HBM reads of the bag); ``None`` keeps every file's own dtype."""
        self.records = list(records)
        self.device = torch.device(device)
        self.depth = max(1, int(depth))
        self.workers = max(1, int(workers))
        self.dtype = dtype
        self.prepare = bool(prepare)
        if self.prepare and dtype is not torch.float32:
            raise ValueError("BagPrefetcher(prepare=True) prepares fp32 bags: leave dtype at torch.float32")
        self.on_gpu = self.device.type == "cuda"
        if self.prepare and not self.on_gpu:
            raise ValueError("BagPrefetcher(prepare=True) needs a HIP device (toad_amd has no CPU path)")
        self.copy_stream = torch.cuda.Stream(device=self.device) if self.on_gpu else None
        self.arena_rows = max(0, int(arena_rows))
        if self.arena_rows and (self.prepare or dtype is not torch.float32):
            raise ValueError("BagPrefetcher(arena_rows=...) lands fp32 bags: leave dtype at torch.float32 and prepare off")
        self._arena: Optional[torch.Tensor] = None              # current landing buffer [arena_rows, features] and the rows already taken
        self._arena_used = 0

    def _land(self, t: torch.Tensor) -> torch.Tensor:
        """Device fp32 copy of host bag ``t`` inside the current landing buffer (called on the copy stream, in record order)."""
        n, k = t.shape
        if n > self.arena_rows or n == 0:
            return t.to(self.device, non_blocking=True).to(torch.float32)
        if self._arena is None or self._arena.shape[1] != k or self._arena_used + n > self.arena_rows:
            self._arena = torch.empty((self.arena_rows, k), dtype=torch.float32, device=self.device)
            self._arena_used = 0
        view = self._arena[self._arena_used:self._arena_used + n]
        view.copy_(t, non_blocking=True)                        # H2D (+ up-cast of fp16 / bf16 files) straight into place
        self._arena_used += n
        return view

    def __len__(self) -> int:
        return len(self.records)

    # -- stage 1 (worker thread): read + pin
    def _stage_host(self, rec: Record):
        src, label, site, sex = rec
        t = _load(src)
        meta = torch.tensor([int(label), int(site)], dtype=torch.int64)
        sx = torch.tensor([float(sex)], dtype=torch.float32)
        if self.on_gpu:                                         # page-locked so every copy is truly asynchronous
            if not t.is_pinned():
                t = t.pin_memory()
            meta, sx = meta.pin_memory(), sx.pin_memory()       # (a pageable source would be a synchronous copy queued
        return t, meta, sx                                      #  behind the bag transfer and stall the host)

    # -- stage 2 (consumer thread, copy stream): H2D + on-device upcast
    def _stage_device(self, host):
        t, meta, sx = host
        if not self.on_gpu:
            if self.arena_rows:
                return (self._land(t), meta[0:1], meta[1:2], sx), None
            return (t if self.dtype is None else t.to(self.dtype), meta[0:1], meta[1:2], sx), None
        with torch.cuda.stream(self.copy_stream):
            if self.arena_rows:
                bag = self._land(t)
            else:
                bag = t.to(self.device, non_blocking=True)
            if self.dtype is not None and bag.dtype != self.dtype:
                bag = bag.to(self.dtype)                        # fp16/bf16 on disk: half the PCIe bytes, upcast here
            from . import ops
            # only bags the prepared-bag kernels take (32-bit row offsets: < 1,048,576 patches): a larger one stays an fp32 tensor and
            # runs on the fp32 path instead of raising in the model after its fp32 copy was dropped
            if self.prepare and bag.dtype == torch.float32 and bag.shape[0] >= 64 and bag.shape[1] == 1024 and ops.x16_ok(bag.shape[0]):
                bag = ops.prepare_bag(bag)                      # on the copy stream (ops use the current stream); the fp32 bag dies here
            meta_d = meta.to(self.device, non_blocking=True)
            sx_d = sx.to(self.device, non_blocking=True)
            ev = torch.cuda.Event()
            ev.record(self.copy_stream)
        return (bag, meta_d[0:1], meta_d[1:2], sx_d), ev

    def __iter__(self) -> Iterator[Tuple[torch.Tensor, torch.Tensor, torch.Tensor, torch.Tensor]]:
        n = len(self.records)
        if n == 0:
            return
        with ThreadPoolExecutor(max_workers=self.workers) as pool:
            host_futs = {}                                       # index -> future of the pinned host bag
            dev_ready = {}                                       # index -> (tensors, event)
            next_host = 0

            def top_up(upto: int):
                nonlocal next_host
                while next_host < min(n, upto):
                    host_futs[next_host] = pool.submit(self._stage_host, self.records[next_host])
                    next_host += 1

            top_up(self.depth + 1)
            for i in range(n):
                # issue the device copies of everything whose host stage is done, up to `depth` ahead
                for jx in range(i, min(n, i + self.depth)):
                    if jx not in dev_ready and jx in host_futs and (jx == i or host_futs[jx].done()):
                        dev_ready[jx] = self._stage_device(host_futs.pop(jx).result())   # .result() re-raises loader errors
                    elif self.arena_rows and jx not in dev_ready:
                        break                                    # landing buffers are filled in record order: never overtake a bag that is not read yet
                top_up(i + self.depth + 2)
                tensors, ev = dev_ready.pop(i)
                if ev is not None:
                    cur = torch.cuda.current_stream(self.device)
                    cur.wait_event(ev)                          # consumer stream waits for the copy, the host does not
                    for t in tensors:                           # allocator: do not recycle while the consumer uses it
                        for u in ((t.planes, t.amax) if getattr(t, "is_prepared_bag", False) else (t,)):
                            u.record_stream(cur)
                yield tensors
