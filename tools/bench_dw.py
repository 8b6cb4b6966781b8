"""A/B timing of the depthwise 7x7 kernels at the bench shapes (env LNX_DWCONV_VALU=1 selects the VALU kernels instead of the MFMA ones)."""
import sys, torch
import os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from linnaeus_amd import ops

def w49(w):
    return w.reshape(w.shape[0], 49).t().contiguous()

def t(fn, n=10):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3

for (B, H, C) in ((256, 56, 96), (256, 28, 192)):
    x = torch.randn(B, H, H, C, device="cuda")
    w = torch.randn(C, 1, 7, 7, device="cuda") / 7
    b = torch.randn(C, device="cuda")
    y = torch.empty(B, H, H, C, device="cuda", dtype=torch.bfloat16)
    dy = torch.randn(B, H, H, C, device="cuda").to(torch.bfloat16)
    g = torch.randn(B, H, H, C, device="cuda")
    dw = torch.zeros(C, 1, 7, 7, device="cuda"); db = torch.zeros(C, device="cuda")
    n = B * H * H * C
    wt = w49(w)
    tf = t(lambda: ops.dwconv7(x, wt, b, y))
    tb = t(lambda: ops.dwconv7(dy, wt, None, g, flip=True, res=g))
    tw = t(lambda: ops.dwconv7_wgrad(x, dy, dw, db))
    print(f"B={B} H={H} C={C}: fwd {tf:7.1f} us ({n * 6 / tf / 1e3:6.0f} GB/s)  dgrad {tb:7.1f} us ({n * 10 / tb / 1e3:6.0f} GB/s)  wgrad {tw:7.1f} us ({n * 6 / tw / 1e3:6.0f} GB/s)")
