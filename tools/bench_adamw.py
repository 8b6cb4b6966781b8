"""FusedAdamW.step() on one large tensor and on many tensors of the same total size (where does the xl step's 2 ms go?)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from linnaeus_amd.optim import FusedAdamW


def run(shapes, label, max_norm=None):
    params = [torch.nn.Parameter(torch.randn(*s, device="cuda")) for s in shapes]
    for p in params:
        p.grad = torch.randn_like(p)
    opt = FusedAdamW(params, lr=1e-4, weight_decay=0.05, max_grad_norm=max_norm)
    for _ in range(3):
        opt.step()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        opt.step()
    e1.record()
    torch.cuda.synchronize()
    n = sum(p.numel() for p in params)
    t = e0.elapsed_time(e1) / 10 * 1e-3
    print(f"{label}: {n / 1e6:.1f} M parameters in {len(params)} tensors: {t * 1e6:8.1f} us  {n * 28 / t / 1e12:5.2f} TB/s (7 x 4 bytes per parameter)", flush=True)


run([(30_000_000,)], "one tensor, 30 M ")
run([(120_000_000,)], "one tensor, 120 M")
run([(1024, 4096)] * 28, "28 x [1024, 4096]")
run([(1024, 1024)] * 100 + [(1024,)] * 300, "100 x [1024,1024] + 300 x [1024]")
run([(120_000_000,)], "one tensor, 120 M, with clipping", max_norm=1.0)
