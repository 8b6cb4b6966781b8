"""Attention forward / backward at the mFormerV1_sm stage-3 / stage-4 shapes (B = 256) and the lg @384 stage-3 shape (B = 64); run it under
`rocprofv3 --kernel-trace --stats` for per-kernel times (the backward is two kernels + the freqs-gradient reduce)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from linnaeus_amd import ops

for B, H, W, E, heads in ((256, 14, 14, 3, 6), (256, 7, 7, 3, 12), (64, 24, 24, 4, 12)):
    N = H * W + E
    Cc = heads * 64
    qkv = torch.randn(B * N, 3 * Cc, device="cuda").bfloat16()
    freqs = torch.randn(2, heads, 32, device="cuda")
    dsin = torch.empty(2, H * W, heads, 32, device="cuda")
    cos = ops.rope_cos_table(freqs, H, W, dsin=dsin)
    o = torch.empty(B * N, Cc, device="cuda", dtype=torch.bfloat16)
    lse = torch.empty(B * heads * N, device="cuda")
    do = torch.randn_like(o)
    dqkv = torch.empty_like(qkv)
    delta = torch.empty_like(lse)
    dfreqs = torch.zeros(2, heads, 32, device="cuda")

    def fwd():
        ops.attn_fwd(qkv, cos, o, lse, B, N, E, heads)

    def bwd():
        ops.attn_bwd(qkv, cos, o, lse, do, dqkv, delta, B, N, E, heads, dsin=dsin, dfreqs=dfreqs)

    for name, fn, fl in (("fwd", fwd, 4.0), ("bwd", bwd, 14.0)):
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            fn()
        e1.record()
        torch.cuda.synchronize()
        t = e0.elapsed_time(e1) / 20 * 1e-3
        print(f"N={N} heads={heads} {name}: {t * 1e6:7.1f} us  {fl * B * heads * N * N * 64 / t / 1e12:6.1f} TFLOP/s", flush=True)
