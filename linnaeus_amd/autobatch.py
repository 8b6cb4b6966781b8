"""AutoBatch for the HIP mFormerV1 (reference: linnaeus/utils/autobatch.py:111-265).

The reference finds the largest per-GPU batch by binary search over REAL training steps, reading
`torch.cuda.max_memory_allocated` and catching out-of-memory errors.  The native plan makes that unnecessary: its
workspace (every saved activation, the operand arena and all scratch) is computed by the planner without allocating, so
the memory of a batch size is

    workspace(B) + dropout keep masks (DROP_RATE / ATTN_DROP_RATE > 0) + logits (+ dlogits) + parameters + gradient arena
    + optimizer state + one input batch

-- every plan-side term exactly as the native planner reports it (`mFormerV1.plan_footprint`).  `auto_find_batch_size` keeps the reference's signature and result (largest batch whose footprint stays
under `target_memory_fraction` of the device, broadcast from rank 0 under DDP), does the search analytically, and then
verifies the winner with `steps_per_trial` real steps, backing off if the measured peak is over budget (allocator
fragmentation, caller-side tensors).  On MI355X the budget is 288 GB per GPU.
"""
from __future__ import annotations

import logging
from typing import Any, Callable, Dict, Optional

import torch

try:
    import torch.distributed as dist
except Exception:  # pragma: no cover
    dist = None


def predicted_bytes(model, batch: int, img_size: Optional[int] = None, *, mode: str = "train", optimizer_state_per_param: int = 2) -> int:
    """Device bytes one step at this per-GPU batch holds: plan workspace + fp32 master weights + fp32 gradient arena +
    `optimizer_state_per_param` fp32 copies (AdamW: 2) + the input batch + logits."""
    m = getattr(model, "module", model)
    n_param = sum(p.numel() for p in m.parameters())
    train = mode == "train"
    H = img_size or m.img_size[0]
    fp = m.plan_footprint(batch, H, H, train=train)
    fixed = 4 * n_param * (1 + ((1 + optimizer_state_per_param) if train else 0))
    io = batch * (m._in_chans * H * H + sum(m.meta_dims)) * 4
    return fp["workspace"] + fp["dropout"] + fp["attn_dropout"] + fp["logits"] + fixed + io


def foreign_bytes(model) -> int:
    """Device bytes currently allocated that are NOT this model's parameters, gradients or buffers (other models, data
    loaders, a previous phase's tensors): they count against the memory budget but are not part of predicted_bytes."""
    m = getattr(model, "module", model)
    m.release_plans()
    dev = next(m.parameters()).device
    own = sum(p.numel() * p.element_size() + (p.grad.numel() * p.grad.element_size() if p.grad is not None else 0) for p in m.parameters())
    own += sum(b.numel() * b.element_size() for b in m.buffers())
    return max(0, torch.cuda.memory_allocated(dev) - own)


def auto_find_batch_size(model, config, mode: str, *, optimizer_main=None, criteria_train=None, grad_weighting_main=None, scaler_main=None,
                         criteria_val=None, target_memory_fraction: float, max_batch_size: int, min_batch_size: int = 1, steps_per_trial: int = 3,
                         log_level: str = "INFO", step_fn: Optional[Callable[[Any, int], None]] = None) -> int:
    """Largest per-GPU batch size within `target_memory_fraction` of the device memory.  `step_fn(model, batch)` (optional)
    runs one real step for the verification trials; without it a forward(+backward of a sum) on random inputs is used."""
    log = logging.getLogger("linnaeus.autobatch")
    log.setLevel(log_level)
    rank = dist.get_rank() if dist is not None and dist.is_available() and dist.is_initialized() else 0
    best = None
    if rank == 0:
        best = _search(model, config, mode, target_memory_fraction, max_batch_size, min_batch_size, steps_per_trial, log, step_fn)
    if dist is not None and dist.is_available() and dist.is_initialized():
        t = torch.tensor(best if best is not None else 0, device="cuda", dtype=torch.int32)
        dist.broadcast(t, src=0)
        best = int(t.item())
    log.info("[auto_find_batch_size] rank=%s found batch size=%s", rank, best)
    return best if best is not None else min_batch_size


def _search(model, config, mode, frac, hi, lo, steps, log, step_fn) -> int:
    m = getattr(model, "module", model)
    dev = next(m.parameters()).device
    if dev.type != "cuda":
        log.info("AutoBatch is intended for GPU devices. Returning min_batch_size for CPU usage.")
        return lo
    total = torch.cuda.get_device_properties(dev).total_memory
    budget = total * frac
    torch.cuda.empty_cache()
    other = foreign_bytes(m)  # what everybody else holds on the device counts against the budget too
    img = int(config.MODEL.IMG_SIZE) if config is not None else m.img_size[0]
    best, low, high = lo, lo, hi
    while low <= high:  # analytic binary search: predicted_bytes is monotone in the batch size
        mid = (low + high) // 2
        need = predicted_bytes(m, mid, img, mode=mode) + other
        if need <= budget:
            best, low = mid, mid + 1
        else:
            high = mid - 1
        log.info("BS=%s => predicted %.2f GB (%s budget %.2f GB)", mid, need / 2**30, "<=" if need <= budget else ">", budget / 2**30)
    # verification with real steps; back off geometrically if the measured peak is over budget
    while best > lo:
        peak = _trial(m, best, img, mode, steps, step_fn)
        if peak is not None and peak <= budget:
            break
        log.info("BS=%s => measured %s over budget; backing off", best, "OOM" if peak is None else f"{peak / 2**30:.2f} GB")
        best = max(lo, int(best * 0.9))
    return best


def _trial(m, batch: int, img: int, mode: str, steps: int, step_fn) -> Optional[int]:
    dev = next(m.parameters()).device
    m.release_plans()
    torch.cuda.empty_cache()
    torch.cuda.reset_peak_memory_stats(dev)
    was_training = m.training
    try:
        m.train(mode == "train")
        for _ in range(max(1, steps)):
            if step_fn is not None:
                step_fn(m, batch)
                continue
            x = torch.rand(batch, m._in_chans, img, img, device=dev)
            meta = torch.rand(batch, sum(m.meta_dims), device=dev) if m.meta_dims else None
            if mode == "train":
                out = m(x, meta)
                sum(v.float().sum() for v in out.values()).backward()
                m.zero_grad(set_to_none=True)
            else:
                with torch.no_grad():
                    m(x, meta)
        torch.cuda.synchronize(dev)
        return int(torch.cuda.max_memory_allocated(dev))
    except torch.cuda.OutOfMemoryError:
        return None
    finally:
        m.train(was_training)
        m.release_plans()
        torch.cuda.empty_cache()
