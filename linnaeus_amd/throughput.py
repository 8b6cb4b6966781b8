"""Inference-throughput benchmark in the reference's own protocol.

Mirror of `linnaeus.evaluation.throughput_tester.throughput_test` (evaluation/throughput_tester.py:13-90; synthetic inputs as
evaluation/synthetic_data.py:6-22): same signature, same `eval_config.THROUGHPUT.{BATCH_SIZES, NUM_ITERATIONS,
WARM_UP_ITERATIONS, META_DIMS}` inputs, same result dictionaries.  The model must be on the GPU (no CPU path)."""
from __future__ import annotations

import time
from typing import Iterable, List, Optional

import torch


def generate_synthetic_data(batch_size: int, img_size: int, in_channels: int, meta_dims):
    images = torch.rand(batch_size, in_channels, img_size, img_size)
    metadata = torch.rand(batch_size, sum(meta_dims))
    return images, metadata


def throughput_test(model: torch.nn.Module, eval_config, *, img_size: int = 224, in_channels: int = 3, meta_dims: Optional[List[int]] = None,
                    device: Optional[torch.device] = None) -> List[dict]:
    device = device or torch.device("cuda")
    model = model.to(device)
    model.eval()
    t_cfg = eval_config.THROUGHPUT
    batch_sizes: Iterable[int] = t_cfg.BATCH_SIZES
    num_iter: int = t_cfg.NUM_ITERATIONS
    warmup: int = t_cfg.WARM_UP_ITERATIONS
    if meta_dims is None:
        meta_dims = getattr(t_cfg, "META_DIMS", [])
    meta_dims = meta_dims or []
    results = []
    for bs in batch_sizes:
        images, meta = generate_synthetic_data(bs, img_size, in_channels, meta_dims)
        images = images.to(device)
        meta = meta.to(device) if sum(meta_dims) > 0 else None
        for _ in range(warmup):
            with torch.no_grad():
                model(images, meta)
        torch.cuda.reset_peak_memory_stats(device)
        torch.cuda.synchronize(device)
        start = time.time()
        for _ in range(num_iter):
            with torch.no_grad():
                model(images, meta)
        torch.cuda.synchronize(device)
        elapsed = time.time() - start
        results.append({"batch_size": bs, "imgs_per_sec": bs * num_iter / elapsed,
                        "memory_used_gb": torch.cuda.max_memory_allocated(device) / 1e9, "gpu_utilization": 0.0})
    return results
