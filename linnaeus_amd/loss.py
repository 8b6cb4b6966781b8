"""Loss on device: the reference's soft-label criterion and the multi-task cross entropy
of the train step as single HIP launches (`lnx_softce`, csrc/loss.hip) behind the reference's module interface.

  TaxonomyAwareLabelSmoothingCE  <- linnaeus/loss/taxonomy_label_smoothing.py:131-408 (same constructor, same
                                    [B] per-sample return, same ignore_index / class-weight behaviour, same errors)
  multitask_cross_entropy        <- the "per-task mean CE, static task weights, summed" loss of the throughput
                                    protocol; forward value and dlogits of all tasks without torch's
                                    log_softmax / nll_loss kernel chain

There is no CPU path: tensors must be on the GPU (LnxError otherwise), like the model itself.
"""
import ctypes as C
from typing import Any, Dict, Optional

import torch
import torch.nn as nn

from . import _lib as L


def _ptr(t):
    return C.c_void_p(t.data_ptr()) if t is not None else None


def _launch(logits, target, soft, smoothing, class_weight, ignore_index, row_scale, scale, loss, loss_sum, dlogits):
    if not logits.is_cuda:
        raise L.LnxError("linnaeus_amd.loss has no CPU path: logits must be on the GPU")
    a = L.SoftCEArgs()
    a.B, a.C = logits.shape
    a.logits, a.ld = _ptr(logits), logits.stride(0)
    a.target = _ptr(target)
    a.soft, a.smoothing = _ptr(soft), float(smoothing)
    a.class_weight = _ptr(class_weight)
    a.ignore_index = -1 if ignore_index is None else int(ignore_index)
    a.row_scale, a.scale = _ptr(row_scale), float(scale)
    a.loss, a.loss_sum = _ptr(loss), _ptr(loss_sum)
    a.dlogits, a.ldd = _ptr(dlogits), (dlogits.stride(0) if dlogits is not None else 0)
    L.check(L.lib().lnx_softce(C.byref(a), C.c_void_p(torch.cuda.current_stream().cuda_stream)), "lnx_softce")


def _as_rows(logits):
    x = logits.float()
    return x if x.stride(-1) == 1 else x.contiguous()


class _SoftCE(torch.autograd.Function):
    """per-sample loss [B]; the unit gradient (d loss[b] / d logits[b, :]) is produced by the same launch"""

    @staticmethod
    def forward(ctx, logits, target, soft, class_weight, ignore_index, smoothing):
        x = _as_rows(logits)
        B, Cn = x.shape
        loss = torch.empty(B, device=x.device, dtype=torch.float32)
        need = logits.requires_grad
        unit = torch.empty(B, Cn, device=x.device, dtype=torch.float32) if need else None
        _launch(x, target, soft, smoothing, class_weight, ignore_index, None, 1.0, loss, None, unit)
        ctx.unit = unit
        ctx.in_dtype = logits.dtype
        return loss

    @staticmethod
    def backward(ctx, go):
        return (ctx.unit * go.unsqueeze(1)).to(ctx.in_dtype), None, None, None, None, None


class TaxonomyAwareLabelSmoothingCE(nn.Module):
    """Label-smoothing cross entropy whose soft labels come from a precomputed [C, C] distribution matrix
    (row c = distribution over classes when the true class is c).  Returns per-sample losses [B]."""

    def __init__(self, soft_label_matrix: torch.Tensor, weight: Optional[torch.Tensor] = None, apply_class_weights: bool = False,
                 ignore_index: Optional[int] = None, config: Optional[Any] = None):
        super().__init__()
        if soft_label_matrix.dim() != 2 or soft_label_matrix.shape[0] != soft_label_matrix.shape[1]:
            raise ValueError("soft_label_matrix must be square [C, C].")
        self.num_classes = soft_label_matrix.shape[0]
        self.register_buffer("soft_labels", soft_label_matrix.clone().float().contiguous())
        self.apply_class_weights = apply_class_weights
        self.ignore_index = ignore_index
        self.config = config
        self.validate_targets = True
        self.weight = None
        if weight is not None:
            if not isinstance(weight, torch.Tensor):
                weight = torch.tensor(weight, dtype=torch.float32)
            self.register_buffer("class_weight", weight.clone().float().contiguous())
            self.weight = self.class_weight

    def forward(self, logits, target: torch.Tensor) -> torch.Tensor:
        if isinstance(logits, dict):  # output of a ConditionalClassifierHead: first [B, num_classes] tensor
            found = None
            for value in logits.values():
                if isinstance(value, torch.Tensor) and value.ndim == 2 and value.shape[1] == self.num_classes:
                    found = value
                    break
            if found is None:
                shapes = {k: v.shape for k, v in logits.items() if isinstance(v, torch.Tensor)}
                raise ValueError(f"Could not find logits tensor with {self.num_classes} classes in input dict. Available shapes: {shapes}")
            logits = found
        elif not isinstance(logits, torch.Tensor):
            raise TypeError(f"Unsupported logits type: {type(logits)}. Expected Tensor or Dict.")
        if logits.shape[1] != self.num_classes:
            raise ValueError(f"Logits dimension mismatch. Expected {self.num_classes} classes, got {logits.shape[1]}.")
        if target.dim() == 2:
            target = target.argmax(dim=1)
        elif target.dim() != 1:
            raise ValueError(f"Target tensor has invalid shape {target.shape}. Expected 1D indices or [B, C] one-hot/soft-representing-one-class.")
        target = target.to(logits.device, torch.long).contiguous()
        if self.soft_labels.device != logits.device:
            self.soft_labels = self.soft_labels.to(logits.device)
        cw = None
        if self.apply_class_weights and self.weight is not None:
            if self.weight.device != logits.device:
                self.weight = self.class_weight = self.weight.to(logits.device)
            cw = self.weight
        loss = _SoftCE.apply(logits, target, self.soft_labels, cw, self.ignore_index, 0.0)
        # Out-of-range targets surface as NaN rows from the kernel; the reference raises IndexError for them
        # (taxonomy_label_smoothing.py:330-343, itself a host sync).  This check is ONE host<->device sync per call:
        # set `validate_targets = False` on the criterion to keep the launch queue asynchronous (NaN rows then
        # propagate into the loss instead of raising).
        if self.validate_targets and torch.isnan(loss).any():
            bad = ((target < 0) | (target >= self.num_classes)).sum().item()
            if bad:
                raise IndexError(f"{bad} target indices out of bounds [0, {self.num_classes - 1}].")
        return loss


class _MultiCE(torch.autograd.Function):
    @staticmethod
    def forward(ctx, weights, targets, smoothing, *logits):
        total = torch.zeros((), device=logits[0].device, dtype=torch.float32)
        grads = []
        for w, t, lg in zip(weights, targets, logits):
            x = _as_rows(lg)
            d = torch.empty_like(x) if lg.requires_grad else None
            _launch(x, t, None, smoothing, None, None, None, w / x.shape[0], None, total, d)
            grads.append(d)
        ctx.grads = grads
        ctx.dtypes = [lg.dtype for lg in logits]
        return total

    @staticmethod
    def backward(ctx, go):
        return (None, None, None) + tuple((g * go).to(dt) if g is not None else None for g, dt in zip(ctx.grads, ctx.dtypes))


def multitask_cross_entropy(outputs: Dict[str, torch.Tensor], targets: Dict[str, torch.Tensor], task_weights: Optional[Dict[str, float]] = None,
                            label_smoothing: float = 0.0) -> torch.Tensor:
    """sum over tasks of task_weight * mean over the batch of cross_entropy(outputs[task], targets[task]).
    One launch per task computes the loss contribution and d(loss)/d(logits); backward is a scalar multiply."""
    tasks = list(outputs.keys())
    ws = [1.0 if task_weights is None else float(task_weights[t]) for t in tasks]
    tg = [targets[t].to(outputs[t].device, torch.long).contiguous() for t in tasks]
    return _MultiCE.apply(ws, tg, float(label_smoothing), *[outputs[t] for t in tasks])
