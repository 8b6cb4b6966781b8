"""Experiment (round 3): does running the batch as TWO half-batch plans on two HIP streams break the chip-wide phase convoy
(every CU in the K loop, then every CU in the HBM-bound epilogue: profiles/r03_nt_v5_phase_timeline.log)?  Two independent
models of batch B/2 on two streams against one model of batch B, forward + loss + backward, same process."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import TASKS, make_model  # noqa: E402
from linnaeus_amd.loss import multitask_cross_entropy  # noqa: E402


class A:
    arch, img = "sm", 224


def build(B):
    cfg, m = make_model(A)
    m = m.cuda()
    m.set_compute_dtype("bf16")
    m.train()
    m.grad_mode = "direct"
    g = torch.Generator(device="cuda").manual_seed(1)
    x = torch.rand(B, 3, 224, 224, device="cuda", generator=g)
    meta = torch.rand(B, 5, device="cuda", generator=g)
    tg = {t: torch.randint(1, c, (B,), device="cuda", generator=g) for t, c in TASKS}
    return m, x, meta, tg


def step(m, x, meta, tg):
    m.zero_grad(set_to_none=True)
    multitask_cross_entropy(m(x, meta), tg).backward()


def timeit(fn, n=15, warm=4):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
one = build(B)
t1 = timeit(lambda: step(*one))
print(f"one plan, batch {B}: {t1:.2f} ms/step (forward + loss + backward, no optimizer)", flush=True)
del one
torch.cuda.empty_cache()
h1, h2 = build(B // 2), build(B // 2)
th = timeit(lambda: step(*h1))
print(f"one plan, batch {B // 2}: {th:.2f} ms/step", flush=True)
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()


def both():
    cur = torch.cuda.current_stream()
    s1.wait_stream(cur)
    s2.wait_stream(cur)
    with torch.cuda.stream(s1):
        step(*h1)
    with torch.cuda.stream(s2):
        step(*h2)
    cur.wait_stream(s1)
    cur.wait_stream(s2)


t2 = timeit(both)
print(f"two plans of batch {B // 2} on two streams: {t2:.2f} ms per pair = {t2 / t1:.3f} x the single batch-{B} plan", flush=True)
