"""Optimizer + step glue on device: `FusedAdamW` is a `torch.optim.Optimizer` whose
`step()` is two HIP launches for the whole model -- the sum of squares of every gradient, then a multi-tensor AdamW
update with the global-norm clip folded in (csrc/optim.hip) -- instead of `clip_grad_norm_` (the reference makes three
gradient passes with host syncs, train.py:282-308) followed by `torch.optim.AdamW.step`.

Same update rule and `param_groups` / `state_dict` layout as `torch.optim.AdamW` (per-group `lr`, `betas`, `eps`,
`weight_decay`; per-parameter `step`, `exp_avg`, `exp_avg_sq`), so LR schedulers and the reference's parameter-group
builder (optimizers/build.py) work unchanged.  GPU fp32 parameters only.
"""
import ctypes as C
from typing import Optional

import numpy as np
import torch

from . import _lib as L


class FusedAdamW(torch.optim.Optimizer):
    def __init__(self, params, lr: float = 1e-3, betas=(0.9, 0.999), eps: float = 1e-8, weight_decay: float = 1e-2, max_grad_norm: Optional[float] = None):
        if lr < 0 or eps < 0 or not (0 <= betas[0] < 1) or not (0 <= betas[1] < 1) or weight_decay < 0:
            raise ValueError("invalid AdamW hyper-parameters")
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay))
        if len(self.param_groups) > L.ADAMW_MAX_GROUPS:
            raise ValueError(f"at most {L.ADAMW_MAX_GROUPS} parameter groups")
        self.max_grad_norm = max_grad_norm
        self._table = None
        self._key = None
        self._sumsq = None

    # ------------------------------------------------------------------ descriptor table
    def _build(self, items):
        dev = items[0][1].device
        lib = L.lib()
        arr = (L.AdamWDesc * len(items))()
        blk = 0
        for i, (gi, p) in enumerate(items):  # gi: hyper-parameter slot = (param group, step count) combination
            st = self.state[p]
            arr[i].p, arr[i].m, arr[i].v, arr[i].g = p.data_ptr(), st["exp_avg"].data_ptr(), st["exp_avg_sq"].data_ptr(), p.grad.data_ptr()
            arr[i].n, arr[i].group, arr[i].block_start = p.numel(), gi, blk
            blk += lib.lnx_adamw_blocks(C.c_int64(p.numel()))
        host = torch.from_numpy(np.frombuffer(arr, dtype=np.uint8).copy())
        self._table = (host.to(dev), len(items), blk)
        if self._sumsq is None or self._sumsq.device != dev:
            self._sumsq = torch.zeros(1, device=dev, dtype=torch.float32)

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        items = []
        for gi, group in enumerate(self.param_groups):
            for p in group["params"]:
                if p.grad is None:
                    continue
                if not p.is_cuda or p.dtype != torch.float32 or not p.is_contiguous() or p.grad.dtype != torch.float32 or not p.grad.is_contiguous():
                    raise L.LnxError("FusedAdamW needs contiguous fp32 parameters and gradients on the GPU")
                st = self.state[p]
                if len(st) == 0:
                    st["step"] = 0
                    st["exp_avg"] = torch.zeros_like(p, memory_format=torch.preserve_format)
                    st["exp_avg_sq"] = torch.zeros_like(p, memory_format=torch.preserve_format)
                st["step"] = int(st["step"]) + 1  # int() also accepts the tensor a torch.optim.AdamW state_dict carries
                items.append((gi, p))
        if not items:
            return loss
        # torch.optim.AdamW bias-corrects PER PARAMETER (its own step count).  Parameters of one group normally share a
        # step count, but one that was frozen for a while, had no gradient on some steps or came from a checkpoint with
        # mixed steps does not: every distinct (group, step) pair gets its own hyper-parameter slot.
        slots = {}
        for gi, p in items:
            slots.setdefault((gi, int(self.state[p]["step"])), len(slots))
        if len(slots) > L.ADAMW_MAX_GROUPS:
            raise L.LnxError(f"FusedAdamW: {len(slots)} distinct (parameter group, step count) combinations, at most {L.ADAMW_MAX_GROUPS} are supported")
        items = [(slots[(gi, int(self.state[p]["step"]))], p) for gi, p in items]
        key = tuple((si, p.data_ptr(), p.grad.data_ptr(), self.state[p]["exp_avg"].data_ptr(), self.state[p]["exp_avg_sq"].data_ptr()) for si, p in items)
        if key != self._key:
            self._build(items)
            self._key = key
        h = L.AdamWHyper()
        h.ngroups = len(slots)
        for (gi, t), si in slots.items():
            group = self.param_groups[gi]
            b1, b2 = group["betas"]
            h.lr[si], h.beta1[si], h.beta2[si], h.eps[si], h.weight_decay[si] = group["lr"], b1, b2, group["eps"], group["weight_decay"]
            h.bias_c1[si], h.bias_c2[si] = 1.0 - b1 ** float(t), 1.0 - b2 ** float(t)
            h.omb1[si], h.omb2[si] = 1.0 - b1, 1.0 - b2
        table, n, blocks = self._table
        lib = L.lib()
        stream = C.c_void_p(torch.cuda.current_stream().cuda_stream)
        clip = self.max_grad_norm is not None and self.max_grad_norm > 0
        if clip:
            L.check(lib.lnx_grad_sumsq(C.c_void_p(table.data_ptr()), n, blocks, C.c_void_p(self._sumsq.data_ptr()), stream), "lnx_grad_sumsq")
        L.check(lib.lnx_adamw_step(C.c_void_p(table.data_ptr()), n, blocks, C.byref(h), C.c_void_p(self._sumsq.data_ptr()) if clip else None,
                                   C.c_float(self.max_grad_norm if clip else 0.0), stream), "lnx_adamw_step")
        return loss

    def grad_norm(self) -> Optional[torch.Tensor]:
        """total L2 norm of the gradients seen by the last clipped step (device scalar, no sync); None without clipping"""
        if self._sumsq is None or self.max_grad_norm is None:
            return None
        return self._sumsq.sqrt()[0]
