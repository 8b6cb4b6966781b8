"""Host -> device prefetch for the training loop's input batches.

The reference moves every batch to the GPU at the top of the step (`h5data/h5dataloader.py:1334-1341`:
`images.cuda(non_blocking=True)` ... on the compute stream), so the copy -- 154 MB for a 256 x 3 x 224 x 224 fp32 batch,
about 3 ms over PCIe -- sits in front of the forward.  `DevicePrefetcher` wraps any iterable of host batches (tensors,
or tuples / lists / dicts of them, as `H5DataLoader.collate_fn` produces) and keeps `depth` batches in flight on a copy
stream, so the step that consumes batch i overlaps the transfer of batch i + 1; the consumer only waits on an event.

    for images, targets, aux_info, *rest in DevicePrefetcher(data_loader):      # drop-in around train.py's loader
        ...

Host tensors that are not pinned are pinned first (a staging copy; pass `pin_memory=True` to the DataLoader to avoid it).
"""
from __future__ import annotations

from collections import deque
from typing import Any, Iterable, Iterator, Optional

import torch


def _to_device(obj: Any, device: torch.device, hold: list) -> Any:
    if isinstance(obj, torch.Tensor):
        if obj.is_cuda:
            return obj
        src = obj if obj.is_pinned() else obj.pin_memory()
        hold.append(src)  # the pinned source must outlive the asynchronous copy
        return src.to(device, non_blocking=True)
    if isinstance(obj, dict):
        return {k: _to_device(v, device, hold) for k, v in obj.items()}
    if isinstance(obj, tuple):
        return tuple(_to_device(v, device, hold) for v in obj)
    if isinstance(obj, list):
        return [_to_device(v, device, hold) for v in obj]
    return obj


def _record_stream(obj: Any, stream: torch.cuda.Stream) -> None:
    if isinstance(obj, torch.Tensor):
        if obj.is_cuda:
            obj.record_stream(stream)  # the caching allocator must not recycle it while `stream` still reads it
    elif isinstance(obj, dict):
        for v in obj.values():
            _record_stream(v, stream)
    elif isinstance(obj, (tuple, list)):
        for v in obj:
            _record_stream(v, stream)


class DevicePrefetcher:
    def __init__(self, loader: Iterable, device: Optional[torch.device] = None, depth: int = 2):
        if depth < 1:
            raise ValueError("depth must be >= 1")
        self.loader = loader
        self.device = torch.device(device) if device is not None else torch.device("cuda", torch.cuda.current_device())
        if self.device.type != "cuda":
            raise ValueError("DevicePrefetcher copies to a GPU: device must be a cuda device")
        self.depth = depth
        self._copy_stream = torch.cuda.Stream(device=self.device)

    def __len__(self) -> int:
        return len(self.loader)  # type: ignore[arg-type]

    def __iter__(self) -> Iterator[Any]:
        it = iter(self.loader)
        inflight: deque = deque()

        def issue() -> bool:
            try:
                host = next(it)
            except StopIteration:
                return False
            hold: list = []
            with torch.cuda.stream(self._copy_stream):
                dev = _to_device(host, self.device, hold)
                ev = torch.cuda.Event()
                ev.record(self._copy_stream)
            inflight.append((dev, ev, hold))
            return True

        for _ in range(self.depth):
            if not issue():
                break
        while inflight:
            dev, ev, hold = inflight.popleft()
            cur = torch.cuda.current_stream(self.device)
            cur.wait_event(ev)          # no host synchronisation: the consumer's stream waits for this batch's copy only
            _record_stream(dev, cur)
            issue()                     # next transfer goes out before the consumer's kernels are enqueued
            yield dev
            del hold
