"""Diagnostic: cycle shares of the depthwise 7x7 kernel's tile loop (needs tools/libstamp.so via LNX_LIB_PATH)."""
import ctypes as C, sys
import os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
from linnaeus_amd import ops, _lib as L

def w49(w): return w.reshape(w.shape[0], 49).t().contiguous()
B, H, Cc = 256, 56, 96
x = torch.randn(B, H, H, Cc, device="cuda"); w = torch.randn(Cc, 1, 7, 7, device="cuda") / 7; b = torch.randn(Cc, device="cuda")
y = torch.empty(B, H, H, Cc, device="cuda", dtype=torch.bfloat16)
dy = torch.randn(B, H, H, Cc, device="cuda").to(torch.bfloat16); g = torch.randn(B, H, H, Cc, device="cuda")
import os
mf = os.environ.get("LNX_DWCONV_VALU") is None
names = ["issue", "mfma", "outtile", "barrier1", "store", "commit", "barrier2"] if mf else ["issue", "fma", "store", "barrier1", "commit", "barrier2"]
for label, fn in (("fwd", lambda: ops.dwconv7(x, w49(w), b, y)), ("dgrad", lambda: ops.dwconv7(dy, w49(w), None, g, flip=True, res=g))):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    out = (C.c_ulonglong * 8)()
    (L.lib().lnx_dbg_dwconv_mfma_stamps if mf else L.lib().lnx_dbg_dwconv_stamps)(out)
    tot = sum(out[:len(names)])
    print(label, "total clk", tot, " ".join(f"{n} {out[i] / tot * 100:.1f}%" for i, n in enumerate(names)))
