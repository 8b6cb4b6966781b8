"""GPU-side batch mixing of the collate step: the reference's group-aware selective Mixup and
CutMix with the same call signature and results, as HIP launches (csrc/collate.hip) instead of per-sample Python.

  GPUSelectiveMixup   <- linnaeus/aug/gpu/selective_mixup.py:14-604
  GPUSelectiveCutMix  <- linnaeus/aug/gpu/selective_cutmix.py:22-543
  exclude_null_samples_from_mixup <- linnaeus/aug/utils.py:46-180
  rand_bbox           <- linnaeus/aug/utils.py:16-43

`batch = (images [B,C,H,W] fp32, targets {task: [B,Cn] soft labels or [B] indices}, aux_info [B,D], meta_validity_masks
[B,D] bool, group_ids [B])`; returns `(mixed_images, mixed_targets, mixed_aux_info, mixed_meta_valids)`.
What differs from the reference is only the mechanics: the in-group permutation is two segmented shuffles on the device
(no per-group Python loop), the metadata "hard pick" is one kernel (the reference loops over samples and chunks with a
host sync each), the probability / lambda / box draws come from the host generator (no `.item()` on a device tensor),
and the caller's aux_info is not modified in place.  There is no CPU path.
"""
from __future__ import annotations

import ctypes as C
import math
import random
from typing import Any, Dict, List, Optional, Tuple

import torch

from . import _lib as L


def rand_bbox(size, lam: float) -> Tuple[int, int, int, int]:
    """(bbx1, bby1, bbx2, bby2): the reference names size[2] "W" and cuts [.., bbx1:bbx2, bby1:bby2]"""
    Wd, Hd = size[2], size[3]
    cut = math.sqrt(1.0 - lam)
    cw, ch = int(Wd * cut), int(Hd * cut)
    cx, cy = random.randint(0, Wd), random.randint(0, Hd)
    return max(0, cx - cw // 2), max(0, cy - ch // 2), min(Wd, cx + cw // 2), min(Hd, cy + ch // 2)


def exclude_null_samples_from_mixup(batch, null_task_keys=None, config=None):
    """group_id -> -1 for samples whose label is null (index 0 / one-hot column 0 > 0.5) in any of the checked tasks"""
    images, targets, aux, masks, gids = batch
    keys = list(targets.keys()) if null_task_keys is None else ([null_task_keys] if isinstance(null_task_keys, str) else list(null_task_keys))
    null = torch.zeros_like(gids, dtype=torch.bool)
    for k in keys:
        if k in targets:
            t = targets[k].to(gids.device)
            null |= (t == 0) if t.dim() == 1 else (t[:, 0] > 0.5)
    return images, targets, aux, masks, torch.where(null, torch.full_like(gids, -1), gids)


def ingroup_permutation(group_ids: torch.Tensor) -> torch.Tensor:
    """Uniform random permutation inside every group of size > 1 (group -1: identity), without host round trips: two
    independent random orders of each group's members are aligned position by position."""
    B = group_ids.numel()
    g = group_ids.to(torch.int64)
    key = g * 2 * B  # group-major; the random tie-break stays inside the group
    o1 = torch.argsort(key + torch.argsort(torch.rand(B, device=g.device)), stable=True)
    o2 = torch.argsort(key + torch.argsort(torch.rand(B, device=g.device)), stable=True)
    perm = torch.empty(B, dtype=torch.int64, device=g.device)
    perm[o1] = o2
    ar = torch.arange(B, device=g.device)
    return torch.where(g == -1, ar, perm)


def _mix_rows(x: torch.Tensor, perm: torch.Tensor, lam: float, mode: int, valid: Optional[torch.Tensor] = None, box=None, hw=None) -> torch.Tensor:
    if not x.is_cuda:
        raise L.LnxError("linnaeus_amd.collate runs on the HIP kernels only: move the batch to the GPU first")
    xf = x.float().contiguous()
    B = xf.shape[0]
    row = xf[0].numel()
    pad = (-row) % 4
    if pad:  # ragged class counts: pad the row (mode 1 never needs it: W % 4 == 0 is required there)
        xf = torch.nn.functional.pad(xf.reshape(B, row), (0, pad))
    out = torch.empty_like(xf)
    a = L.MixArgs()
    a.x, a.perm, a.out = xf.data_ptr(), perm.data_ptr(), out.data_ptr()
    vb = valid.to(torch.uint8).contiguous() if valid is not None else None
    a.valid = vb.data_ptr() if vb is not None else None
    a.B, a.row, a.lam, a.mode = B, row + pad, float(lam), mode
    if mode == 1:
        a.H, a.W = hw
        a.h0, a.h1, a.w0, a.w1 = box
    L.check(L.lib().lnx_mix_rows(C.byref(a), C.c_void_p(torch.cuda.current_stream().cuda_stream)), "lnx_mix_rows")
    if pad:
        out = out[:, :row]
    return out.reshape(x.shape).to(x.dtype if x.is_floating_point() else torch.float32)


class _SelectiveMixBase:
    def __init__(self, mix_config: Dict[str, Any], config=None):
        self.mix_config = mix_config
        self.config = config
        b = mix_config.get("meta_chunk_bounds_list")
        self.chunk_bounds: Optional[List[Tuple[int, int]]] = [tuple(x) for x in b] if isinstance(b, list) else None
        self.last_permutation = None
        self._bounds_dev = {}
        self._inject = None  # tests: dict(perm=, lam=, pick=, box=) recorded from the reference run

    def _mix_meta(self, aux: torch.Tensor, masks: torch.Tensor, perm: torch.Tensor, pick: Optional[torch.Tensor]):
        B, D = aux.shape
        if D == 0:
            return aux.clone(), masks.clone()
        bounds = self.chunk_bounds if self.chunk_bounds is not None else [(0, D)]
        key = (tuple(bounds), str(aux.device))
        bd = self._bounds_dev.get(key)
        if bd is None:
            bd = self._bounds_dev[key] = torch.tensor([v for se in bounds for v in se], dtype=torch.int32, device=aux.device)
        if pick is None:
            pick = torch.rand(B, device=aux.device)
        af = aux.float().contiguous()
        mk = masks.to(torch.uint8).contiguous()
        oa, om = torch.empty_like(af), torch.empty_like(mk)
        st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
        L.check(L.lib().lnx_mix_meta(C.c_void_p(af.data_ptr()), C.c_void_p(mk.data_ptr()), C.c_void_p(perm.data_ptr()), C.c_void_p(pick.float().contiguous().data_ptr()),
                                     C.c_void_p(bd.data_ptr()), len(bounds), B, D, C.c_void_p(oa.data_ptr()), C.c_void_p(om.data_ptr()), st), "lnx_mix_meta")
        # columns outside every chunk keep the sample's own values (the reference leaves them uninitialised)
        covered = torch.zeros(D, dtype=torch.bool, device=aux.device)
        for s, e in bounds:
            covered[s:e] = True
        if not bool(covered.all()):
            oa = torch.where(covered, oa, af)
            om = torch.where(covered, om, mk)
        return oa.to(aux.dtype), om.to(masks.dtype)

    def _prologue(self, batch, exclude_null_samples, null_task_keys):
        if exclude_null_samples:
            batch = exclude_null_samples_from_mixup(batch, null_task_keys, config=self.config)
        return batch

    def _perm(self, gids):
        perm = self._inject["perm"].to(gids.device) if self._inject else ingroup_permutation(gids)
        self.last_permutation = perm
        return perm.contiguous()


class GPUSelectiveMixup(_SelectiveMixBase):
    """mix_config: {"PROB", "ALPHA", "meta_chunk_bounds_list"} (linnaeus/aug/gpu/selective_mixup.py:29-45)"""

    def __call__(self, batch, exclude_null_samples: bool = True, null_task_keys=None):
        images, targets, aux, masks, gids = self._prologue(batch, exclude_null_samples, null_task_keys)
        inj = self._inject
        if inj is None and random.random() > self.mix_config["PROB"]:
            return images, targets, aux, masks
        if bool((gids == -1).all()):
            return images, targets, aux, masks
        perm = self._perm(gids)
        alpha = self.mix_config["ALPHA"]
        lam = float(inj["lam"]) if inj else float(torch.distributions.beta.Beta(alpha, alpha).sample())
        mixed_images = _mix_rows(images, perm, lam, 0)
        mixed_targets = {k: _mix_rows(v, perm, lam, 0) for k, v in targets.items()}
        mixed_aux, mixed_masks = self._mix_meta(aux, masks, perm, inj["pick"].to(aux.device) if inj else None)
        return mixed_images, mixed_targets, mixed_aux, mixed_masks


class GPUSelectiveCutMix(_SelectiveMixBase):
    """mix_config: {"PROB", "ALPHA", "MINMAX", "meta_chunk_bounds_list"} (linnaeus/aug/gpu/selective_cutmix.py:40-90)"""

    def __init__(self, mix_config, config=None):
        super().__init__(mix_config, config)
        self.minmax = mix_config.get("MINMAX", None)

    def __call__(self, batch, exclude_null_samples: bool = True, null_task_keys=None):
        images, targets, aux, masks, gids = self._prologue(batch, exclude_null_samples, null_task_keys)
        inj = self._inject
        if inj is None and random.random() > self.mix_config.get("PROB", 1.0):
            return images, targets, aux, masks
        if bool((gids == -1).all()):
            return images, targets, aux, masks
        perm = self._perm(gids)
        alpha = self.mix_config.get("ALPHA", 1.0)
        lam = float(inj["lam"]) if inj else float(torch.distributions.beta.Beta(alpha, alpha).sample())
        if self.minmax is not None:
            lam = self.minmax[0] + (self.minmax[1] - self.minmax[0]) * lam
        B, Cc, H, W = images.shape
        bbx1, bby1, bbx2, bby2 = inj["box"] if inj else rand_bbox((1, Cc, H, W), lam)
        lam_adj = 1.0 - ((bbx2 - bbx1) * (bby2 - bby1) / (H * W))
        valid = gids != -1
        # the reference cuts images[:, :, bbx1:bbx2, bby1:bby2]: "x" runs over dim 2 (rows), "y" over dim 3 (columns)
        mixed_images = _mix_rows(images, perm, 0.0, 1, valid, (int(bbx1), int(bbx2), int(bby1), int(bby2)), (H, W))
        mixed_targets = {k: _mix_rows(v, perm, lam_adj, 2, valid) for k, v in targets.items()}
        mixed_aux, mixed_masks = self._mix_meta(aux, masks, perm, inj["pick"].to(aux.device) if inj else None)
        return mixed_images, mixed_targets, mixed_aux, mixed_masks
