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


def _fill(a, logits, target, soft, smoothing, class_weight, ignore_index, row_scale, scale, loss, loss_sum, dlogits):
    if not logits.is_cuda:
        raise L.LnxError("linnaeus_amd.loss has no CPU path: logits must be on the GPU")
    a.B, a.C = logits.shape
    a.logits, a.ld = _ptr(logits), logits.stride(0)
    a.target = _ptr(target)
    a.soft, a.smoothing = _ptr(soft), float(smoothing)
    a.class_weight = _ptr(class_weight)
    a.ignore_index = -1 if ignore_index is None else int(ignore_index)
    a.row_scale, a.scale = _ptr(row_scale), float(scale)
    a.loss, a.loss_sum = _ptr(loss), _ptr(loss_sum)
    a.dlogits, a.ldd = _ptr(dlogits), (dlogits.stride(0) if dlogits is not None else 0)


def _launch(*args):
    a = L.SoftCEArgs()
    _fill(a, *args)
    L.check(L.lib().lnx_softce(C.byref(a), C.c_void_p(torch.cuda.current_stream().cuda_stream)), "lnx_softce")


SOFTCE_MAX_TASKS = 8  # == LNX_SOFTCE_MAX_TASKS


def _launch_multi(sets):
    """`sets`: argument tuples of _fill, at most SOFTCE_MAX_TASKS per launch"""
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    for i0 in range(0, len(sets), SOFTCE_MAX_TASKS):
        chunk = sets[i0:i0 + SOFTCE_MAX_TASKS]
        arr = (L.SoftCEArgs * len(chunk))()
        for a, args in zip(arr, chunk):
            _fill(a, *args)
        L.check(L.lib().lnx_softce_multi(arr, len(chunk), st), "lnx_softce_multi")


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
        xs = [_as_rows(lg) for lg in logits]
        # the gradients of all tasks live in one flat buffer: one launch fills them, one multiply scales them in backward
        sizes = [x.numel() if lg.requires_grad else 0 for x, lg in zip(xs, logits)]
        flat = torch.empty(sum(sizes), device=total.device, dtype=torch.float32) if any(sizes) else None
        grads, sets, off = [], [], 0
        for w, t, x, n in zip(weights, targets, xs, sizes):
            d = flat[off:off + n].view_as(x) if n else None
            off += n
            sets.append((x, t, None, smoothing, None, None, None, w / x.shape[0], None, total, d))
            grads.append(d)
        _launch_multi(sets)
        ctx.flat, ctx.sizes = flat, sizes
        ctx.shapes = [x.shape for x in xs]
        ctx.dtypes = [lg.dtype for lg in logits]
        return total

    @staticmethod
    def backward(ctx, go):
        if ctx.flat is None:
            return (None, None, None) + (None,) * len(ctx.sizes)
        scaled = ctx.flat * go
        out, off = [], 0
        for n, shp, dt in zip(ctx.sizes, ctx.shapes, ctx.dtypes):
            out.append(scaled[off:off + n].view(shp).to(dt) if n else None)
            off += n
        return (None, None, None) + tuple(out)


def multitask_cross_entropy(outputs: Dict[str, torch.Tensor], targets: Dict[str, torch.Tensor], task_weights: Optional[Dict[str, float]] = None,
                            label_smoothing: float = 0.0) -> torch.Tensor:
    """sum over tasks of task_weight * mean over the batch of cross_entropy(outputs[task], targets[task]).
    One launch per task computes the loss contribution and d(loss)/d(logits); backward is a scalar multiply."""
    tasks = list(outputs.keys())
    ws = [1.0 if task_weights is None else float(task_weights[t]) for t in tasks]
    tg = [targets[t].to(outputs[t].device, torch.long).contiguous() for t in tasks]
    return _MultiCE.apply(ws, tg, float(label_smoothing), *[outputs[t] for t in tasks])


# ----------------------------------------------------------------------------------------------------------------------
# The whole loss path of the train step on device (remainder): weighted_hierarchical_loss with null
# masking, class weighting and static task weighting, plus the taxonomy smoothing-matrix builder.
#   build_taxonomy_smoothing_matrix <- loss/taxonomy_label_smoothing.py:30-130
#   compute_core_loss               <- loss/core_loss.py:19-100
#   apply_null_masking / apply_class_weighting / apply_loss_masking <- loss/masking.py:19-465, 469-518, 521-700
#   GradientWeighting (static mode) <- loss/gradient_weighting.py:178-365
#   weighted_hierarchical_loss      <- loss/hierarchical_loss.py:24-406
# The per-sample criterion runs in the HIP kernel (lnx_softce); everything after it is arithmetic on [B]-sized device
# vectors.  What changes against the reference is HOW, not WHAT: its per-sample Python loops with .item() (a host sync
# per sample, loss/masking.py:501-503, gradient_weighting.py:338-342) become one gather from a class-weight vector, and
# the valid-sample counts stay on the device.  Finding F13 is reproduced, not fixed: class weights enter three times on
# the scheduled-masking path, twice on the PHASE1 / validation paths, and the mean divides by count(loss != 0).
# ----------------------------------------------------------------------------------------------------------------------
def build_taxonomy_smoothing_matrix(num_classes: int, distances: torch.Tensor, alpha: float = 0.1, beta: float = 1.0,
                                    uniform_roots: bool = True, root_class_ids=None) -> torch.Tensor:
    """[C, C] soft-label matrix: row i = (1 - alpha) on the diagonal, alpha spread over the other classes in proportion
    to exp(-beta * distance) (uniformly for root classes, and for rows whose neighbours are all disconnected)."""
    if not (0.0 <= alpha <= 1.0):
        raise ValueError(f"alpha must be in [0, 1], got {alpha}")
    if beta < 0:
        raise ValueError(f"beta must be non-negative, got {beta}")
    if num_classes <= 0:
        raise ValueError("num_classes must be positive.")
    if distances.shape != (num_classes, num_classes):
        raise ValueError(f"distances must be shape ({num_classes},{num_classes}), got {distances.shape}")
    C_ = num_classes
    d = distances.float()
    w = torch.exp(-beta * d)
    w = torch.where(torch.isinf(d), torch.zeros_like(w), w)
    eye = torch.eye(C_, dtype=torch.bool, device=d.device)
    w = w.masked_fill(eye, 0.0)
    if uniform_roots and root_class_ids and C_ > 1:
        roots = torch.as_tensor(list(root_class_ids), dtype=torch.long, device=d.device)
        w[roots] = torch.full((C_,), 1.0 / (C_ - 1), device=d.device)
        w = w.masked_fill(eye, 0.0)
    elif uniform_roots and root_class_ids and C_ == 1:
        w.zero_()
    rs = w.sum(1, keepdim=True)
    if C_ > 1:
        fallback = torch.full((C_, C_), alpha / (C_ - 1), device=d.device).masked_fill(eye, 0.0)
        probs = torch.where(rs > 1e-9, w * (alpha / rs.clamp_min(1e-30)), fallback)
    else:
        probs = torch.zeros_like(w)
    probs = probs.masked_fill(eye, 1.0 - alpha)
    tot = probs.sum(1, keepdim=True)
    return torch.where((tot - 1.0).abs() > 1e-6, probs / tot, probs)


def _is_null(target: torch.Tensor) -> torch.Tensor:
    return target == 0 if target.dim() == 1 else target[:, 0] > 0.5


def _sorted_tasks(d):
    return sorted(d.keys(), key=lambda k: int(k.split("_L")[-1]))


def compute_core_loss(outputs, targets, criteria, config=None):
    """{task: per-sample loss [B]} in rank order; each criterion returns a [B] vector."""
    return {t: criteria[t](outputs[t], targets[t]) for t in _sorted_tasks(outputs)}


def _class_weight_vector(cw_dict, num_classes: int, device) -> torch.Tensor:
    v = torch.ones(num_classes, dtype=torch.float32)
    for i, w in cw_dict.items():
        if 0 <= int(i) < num_classes:
            v[int(i)] = float(w)
    return v.to(device)


def _sample_weights(cw_dict, target: torch.Tensor, cache: Optional[dict] = None, key=None) -> torch.Tensor:
    """per-sample class weight: cw[label] for hard labels, <soft target, cw> for [B, C] targets; missing classes weigh 1"""
    if target.dim() == 1:
        n = max(int(max(cw_dict.keys(), default=0)) + 1, 1)
        ck = (key, n, str(target.device))
        vec = cache.get(ck) if cache is not None else None
        if vec is None:
            vec = _class_weight_vector(cw_dict, n, target.device)
            if cache is not None:
                cache[ck] = vec
        idx = target.clamp(0, n - 1)
        return torch.where(target < n, vec[idx], torch.ones((), device=target.device))
    vec = _class_weight_vector(cw_dict, target.size(1), target.device)
    return (target.float() * vec.unsqueeze(0)).sum(1)


def apply_class_weighting(per_task_losses, targets, class_weights=None, _cache=None):
    if class_weights is None:
        return per_task_losses
    out = {}
    for t, vec in per_task_losses.items():
        out[t] = vec * _sample_weights(class_weights[t], targets[t], _cache, t).to(vec.dtype) if t in class_weights else vec
    return out


def apply_null_masking(per_task_losses, targets, null_mask_prob: float, logger=None, config=None, _coin=None):
    """Zero the loss of null-labelled samples (label 0) except for a random `null_mask_prob` fraction of them.
    Statistics are device scalars (the reference calls .item() on each).  `_coin`: {task: [B] uniform draws} for tests."""
    masked, stats = {}, {"null_mask_prob": null_mask_prob}
    tot = inc = None
    for t, vec in per_task_losses.items():
        null = _is_null(targets[t])
        if null_mask_prob < 1.0:
            u = _coin[t].to(vec.device) if _coin is not None else torch.rand(vec.shape[0], device=vec.device)
            keep = (~null) | (u < null_mask_prob)
        else:
            keep = torch.ones_like(null)
        masked[t] = torch.where(keep, vec, torch.zeros((), dtype=vec.dtype, device=vec.device))
        n, k = null.sum(), (null & keep).sum()
        tot, inc = (n, k) if tot is None else (tot + n, inc + k)
    stats["null_samples_total"], stats["null_samples_included"] = tot, inc
    stats["inclusion_percentage"] = inc.float() * 100.0 / tot.float().clamp_min(1.0) if tot is not None else 0.0
    return masked, stats


def apply_loss_masking(per_task_losses, targets, ops_schedule, current_step, class_weights=None, is_validation=False, logger=None, config=None,
                       _coin=None, _cache=None):
    if is_validation:
        prob = 1.0
    elif config is not None and getattr(config.TRAIN, "PHASE1_MASK_NULL_LOSS", False):
        prob = 0.0
    else:
        prob = float(ops_schedule.get_null_mask_prob(current_step))
    masked, stats = apply_null_masking(per_task_losses, targets, prob, logger, config, _coin)
    stats["num_valid_samples_per_task"] = {t: (v != 0).sum() for t, v in masked.items()}  # device scalars
    if class_weights is not None:
        return apply_class_weighting(masked, targets, class_weights, _cache), stats
    return masked, stats


class GradientWeighting(nn.Module):
    """Task weighting of the multi-task loss, static mode (the reference's default for fixed weights).  GradNorm (a
    second backward through the backbone per task) is outside the hot path and not provided."""

    def __init__(self, task_keys, config=None, task_weighting_type: str = "static", init_weights=None, class_weights=None,
                 use_subset_weights: bool = False, **kwargs):
        super().__init__()
        if task_weighting_type != "static":
            raise NotImplementedError("linnaeus_amd.loss.GradientWeighting implements static task weights; GradNorm is out of scope")
        self.task_keys = list(task_keys)
        self.config = config
        self.task_weighting_type = task_weighting_type
        if isinstance(init_weights, dict):
            init_weights = [init_weights.get(k, 1.0) for k in self.task_keys]
        self.task_weights = torch.tensor(init_weights or [1.0] * len(self.task_keys), dtype=torch.float32)
        self.gradnorm = None
        self.class_weights = class_weights
        self.use_subset_weights = use_subset_weights
        self._cache: dict = {}

    def _normalize_weights(self, w):
        return w

    def forward(self, per_task_losses, targets, subset_ids=None, mixed_subset_ids=None, num_valid_samples_per_task=None):
        first = next(iter(per_task_losses.values()))
        norm_w = self._normalize_weights(self.task_weights)
        weighted = {}
        for i, t in enumerate(self.task_keys):
            vec = per_task_losses[t]
            nv = vec.shape[0]
            if num_valid_samples_per_task is not None:
                nv = num_valid_samples_per_task.get(t, vec.shape[0])
            if self.class_weights and t in self.class_weights:
                vec = vec * _sample_weights(self.class_weights[t], targets[t], self._cache, t).to(vec.dtype)
            denom = nv.to(vec.dtype).clamp_min(1e-6) if isinstance(nv, torch.Tensor) else max(float(nv), 1e-6)
            weighted[t] = vec.sum() / denom * float(norm_w[i])
        return weighted, dict(zip(self.task_keys, norm_w.tolist()))


def weighted_hierarchical_loss(outputs, targets, criteria, task_weighting, ops_schedule, current_step: int, subset_ids=None, mixed_subset_ids=None,
                               is_validation: bool = False, logger=None, config=None, *, sync_components: bool = True, _coin=None):
    """(total_loss, loss_components, task_weights) of the reference's train / validation step.  With
    `sync_components=False` the logging values stay device scalars (no host sync in the step)."""
    keys = _sorted_tasks(outputs)
    if not isinstance(targets, dict):
        targets = dict(zip(keys, targets))
    per = compute_core_loss(outputs, targets, criteria, config)
    raw = {k: v.detach().clone() for k, v in per.items()}
    phase1 = bool(config is not None and getattr(config.TRAIN, "PHASE1_MASK_NULL_LOSS", False))
    cache = getattr(task_weighting, "_cache", None)
    if phase1 and not is_validation:
        masked = {t: v * (~_is_null(targets[t])).to(v.dtype) for t, v in per.items()}
        zero = torch.zeros((), device=next(iter(per.values())).device)
        stats = {"null_samples_total": zero, "null_samples_included": zero, "inclusion_percentage": 0.0, "null_mask_prob": 0.0}
    else:
        masked, stats = apply_loss_masking(per, targets, ops_schedule, current_step, task_weighting.class_weights, is_validation, logger, config,
                                           _coin=_coin, _cache=cache)
    stats["phase1_active"] = phase1 and not is_validation
    after_cw = masked
    if task_weighting.class_weights:
        try:
            apply_cw = config.LOSS.GRAD_WEIGHTING.CLASS.TRAIN if not is_validation else config.LOSS.GRAD_WEIGHTING.CLASS.VAL
        except Exception:
            apply_cw = True
        if apply_cw:
            after_cw = apply_class_weighting(masked, targets, task_weighting.class_weights, cache)
    weighted, task_weights = task_weighting(after_cw, targets, num_valid_samples_per_task=stats.get("num_valid_samples_per_task", {}))
    total = sum(weighted.values())
    val = (lambda x: x.item()) if sync_components else (lambda x: x.detach())
    comps = {"total": val(total), "tasks": {t: val(per[t].mean()) for t in keys}, "masked_tasks": {t: val(after_cw[t].mean()) for t in keys},
             "weighted_tasks": {t: val(weighted[t]) for t in keys}, "raw_per_sample_losses": raw, "null_masking": stats}
    return total, comps, task_weights
