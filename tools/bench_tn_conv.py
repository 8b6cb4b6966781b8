"""Timing of the conv-stage weight-gradient GEMM shapes in both operand orders (dW vs its transpose)."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
from linnaeus_amd import ops

def t(fn, n=10):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3

for M, C in ((256 * 56 * 56, 96), (256 * 28 * 28, 192)):
    wide = torch.randn(M, 4 * C, device="cuda").bfloat16()
    narrow = torch.randn(M, C, device="cuda").bfloat16()
    ws = None
    for name, dY, A in (("dW1  = dh^T.ln  [4C x C]", wide, narrow), ("dW2  = dz^T.act [C x 4C]", narrow, wide)):
        dW = torch.zeros(dY.shape[1], A.shape[1], device="cuda")
        db = torch.zeros(dY.shape[1], device="cuda")
        print(f"M={M} C={C} {name}: {t(lambda: ops.gemm_tn(dY, A, dW, db=db)):7.1f} us")
