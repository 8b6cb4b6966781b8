"""A synthetic data set in ONE flat, memory-mapped file behind the reference's per-sample read contract (config
2's "synthetic HDF5" made literal without h5py, which this image does not have).

`PrefetchingH5Dataset._read_raw_item(idx)` (linnaeus/h5data/prefetching_h5_dataset.py:185-360) returns

    (image [3, S, S] float32 in [0, 1], targets {task: one-hot [n_classes]}, aux_info [D] float32, group_id int,
     subset_ids {name: int}, meta_validity_mask [D] bool)

from an images file (uint8 HWC) and a labels file (integer label per task, label 0 = null -> class index 0; metadata
components in IDX order, a component whose stored vector is all zero is null: zeroed and masked out).  `FlatSyntheticDataset`
returns exactly that tuple from a file `write_synthetic_flat` produces: a JSON header (array names, dtypes, shapes, byte
offsets) followed by the raw arrays, each 4096-byte aligned, so that `np.memmap` reads a sample without touching the rest.
`FlatBatchLoader` yields collated host batches in the layout `H5DataLoader.collate_fn` hands to the GPU mixers
(images, targets, aux_info, meta_validity_masks, group_ids); with `raw_uint8=True` images stay uint8 [B, S, S, 3] for
`linnaeus_amd.aug.u8hwc_to_f32chw` to convert after the transfer (a quarter of the PCIe bytes)."""
from __future__ import annotations

import json
import struct
from typing import Dict, Iterator, List, Optional, Sequence, Tuple

import numpy as np
import torch

MAGIC = b"LNXFLAT1"
ALIGN = 4096


def write_synthetic_flat(path: str, n: int, img_size: int, tasks: Dict[str, int], meta: Sequence[Tuple[str, int]] = (("TEMPORAL", 2), ("SPATIAL", 3)),
                         seed: int = 0, null_fraction: float = 0.05, n_groups: int = 16) -> Dict[str, dict]:
    """Write `n` synthetic samples: uniform random uint8 images (evaluation/synthetic_data.py draws U[0, 1) floats; stored images
    are bytes), labels in 1..C-1 per task with a fraction of nulls (0), metadata U(0, 1) with a fraction of all-zero (null)
    components, a group id per sample."""
    rng = np.random.default_rng(seed)
    arrays: Dict[str, np.ndarray] = {"images": rng.integers(0, 256, (n, img_size, img_size, 3), dtype=np.uint8)}
    for t, c in tasks.items():
        lab = rng.integers(1, c, n, dtype=np.int32)
        lab[rng.random(n) < null_fraction] = 0
        arrays["label/" + t] = lab
    for name, dim in meta:
        m = rng.random((n, dim), dtype=np.float32) * 0.98 + 0.01
        m[rng.random(n) < null_fraction] = 0.0
        arrays["meta/" + name] = m
    arrays["group_ids"] = rng.integers(0, n_groups, n, dtype=np.int64)
    index, off = {}, 0
    for k, a in arrays.items():
        off = (off + ALIGN - 1) // ALIGN * ALIGN
        index[k] = {"dtype": str(a.dtype), "shape": list(a.shape), "offset": off}
        off += a.nbytes
    header = json.dumps({"n": n, "img_size": img_size, "tasks": tasks, "meta": [list(m) for m in meta], "arrays": index}).encode()
    base = (len(MAGIC) + 8 + len(header) + ALIGN - 1) // ALIGN * ALIGN
    with open(path, "wb") as f:
        f.write(MAGIC + struct.pack("<Q", len(header)) + header)
        for k, a in arrays.items():
            f.seek(base + index[k]["offset"])
            f.write(np.ascontiguousarray(a).tobytes())
    return index


class FlatSyntheticDataset(torch.utils.data.Dataset):
    def __init__(self, path: str, tasks: Optional[List[str]] = None):
        with open(path, "rb") as f:
            if f.read(len(MAGIC)) != MAGIC:
                raise ValueError(f"{path}: not a linnaeus_amd flat data file")
            (hl,) = struct.unpack("<Q", f.read(8))
            self.header = json.loads(f.read(hl))
        base = (len(MAGIC) + 8 + hl + ALIGN - 1) // ALIGN * ALIGN
        self._a = {k: np.memmap(path, mode="r", dtype=np.dtype(v["dtype"]), shape=tuple(v["shape"]), offset=base + v["offset"])
                   for k, v in self.header["arrays"].items()}
        self.tasks = list(tasks) if tasks is not None else list(self.header["tasks"].keys())
        for t in self.tasks:
            if "label/" + t not in self._a:
                raise KeyError(f"task {t} is not in {path}")
        self.num_classes = {t: int(self.header["tasks"][t]) for t in self.tasks}
        self.meta = [(str(nm), int(d)) for nm, d in self.header["meta"]]

    def __len__(self) -> int:
        return int(self.header["n"])

    def raw_image(self, idx: int) -> np.ndarray:
        return self._a["images"][idx]

    def _read_raw_item(self, idx: int):
        image = torch.from_numpy(np.array(self._a["images"][idx])).permute(2, 0, 1).float() / 255.0
        targets = {}
        for t in self.tasks:
            one_hot = torch.zeros(self.num_classes[t], dtype=torch.float32)
            one_hot[int(self._a["label/" + t][idx])] = 1.0  # label 0 = null = class index 0
            targets[t] = one_hot
        aux, valid = [], []
        for nm, dim in self.meta:
            v = np.array(self._a["meta/" + nm][idx], dtype=np.float32)
            ok = not bool(np.all(v == 0.0))
            aux.append(v if ok else np.zeros_like(v))
            valid.append(np.full(dim, ok, dtype=np.bool_))
        aux_info = torch.from_numpy(np.concatenate(aux)) if aux else torch.zeros(0)
        mask = torch.from_numpy(np.concatenate(valid)) if valid else torch.zeros(0, dtype=torch.bool)
        return image, targets, aux_info, int(self._a["group_ids"][idx]), {}, mask

    __getitem__ = _read_raw_item


class FlatBatchLoader:
    """Sequential (optionally shuffled per epoch) batches of a FlatSyntheticDataset, collated on the host:
    (images, {task: [B, C] one-hot}, aux_info [B, D], meta_validity_masks [B, D] bool, group_ids [B] int64).
    `workers` threads collate whole batches ahead of the consumer (the reference's DataLoader workers; numpy's gathers release the
    GIL), in order; `pin=True` gathers the images straight into pinned memory, so that DevicePrefetcher copies without re-staging."""

    def __init__(self, ds: FlatSyntheticDataset, batch_size: int, shuffle: bool = False, seed: int = 0, raw_uint8: bool = False, drop_last: bool = True,
                 epochs: Optional[int] = 1, workers: int = 0, pin: bool = False):
        self.ds, self.B, self.shuffle, self.raw, self.drop_last, self.epochs = ds, int(batch_size), shuffle, raw_uint8, drop_last, epochs
        self.workers, self.pin = int(workers), bool(pin)
        self._rng = np.random.default_rng(seed)

    def _collate(self, idx: np.ndarray):
        a = self.ds._a
        B = len(idx)
        img = a["images"]
        raw = torch.empty((B,) + img.shape[1:], dtype=torch.uint8, pin_memory=self.pin)
        np.take(img, idx, axis=0, out=raw.numpy(), mode="clip")  # one gather, in the order of `idx` (mode='raise' would buffer `out`)
        images = raw if self.raw else raw.permute(0, 3, 1, 2).float().div_(255.0)
        targets = {}
        for t in self.ds.tasks:
            lab = torch.from_numpy(np.take(a["label/" + t], idx).astype(np.int64))
            targets[t] = torch.nn.functional.one_hot(lab, self.ds.num_classes[t]).float()
        aux, valid = [], []
        for nm, dim in self.ds.meta:
            v = torch.from_numpy(np.take(a["meta/" + nm], idx, axis=0))
            ok = ~(v == 0).all(dim=1, keepdim=True)
            aux.append(v * ok)
            valid.append(ok.expand(-1, dim))
        aux_info = torch.cat(aux, 1) if aux else torch.zeros(B, 0)
        masks = torch.cat(valid, 1) if valid else torch.zeros(B, 0, dtype=torch.bool)
        gids = torch.from_numpy(np.take(a["group_ids"], idx))
        return images, targets, aux_info, masks.contiguous(), gids

    def _index_batches(self) -> Iterator[np.ndarray]:
        n = len(self.ds)
        ep = 0
        while self.epochs is None or ep < self.epochs:
            perm = self._rng.permutation(n) if self.shuffle else np.arange(n)
            for i in range(0, n - (self.B - 1 if self.drop_last else 0), self.B):
                yield perm[i:i + self.B]
            ep += 1

    def __iter__(self) -> Iterator:
        if self.workers <= 0:
            for idx in self._index_batches():
                yield self._collate(idx)
            return
        from collections import deque
        from concurrent.futures import ThreadPoolExecutor

        with ThreadPoolExecutor(self.workers) as pool:
            pending = deque()
            for idx in self._index_batches():
                pending.append(pool.submit(self._collate, idx))
                if len(pending) > 2 * self.workers:
                    yield pending.popleft().result()
            while pending:
                yield pending.popleft().result()
