"""Diagnostic: cycle shares of the streamed-weight conv-MLP forward kernel, C = 192 (needs tools/libstamp.so via LNX_LIB_PATH)."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
from linnaeus_amd import ops, _lib as L

M, Cc = 256 * 28 * 28, 192
ln = torch.randn(M, Cc, device="cuda").bfloat16()
x = torch.randn(M, Cc, device="cuda")
w1 = (torch.randn(4 * Cc, Cc, device="cuda") / Cc ** 0.5).bfloat16()
w2 = (torch.randn(Cc, 4 * Cc, device="cuda") / (4 * Cc) ** 0.5).bfloat16()
b1 = torch.randn(4 * Cc, device="cuda"); b2 = torch.randn(Cc, device="cuda"); gamma = torch.rand(Cc, device="cuda")
out = torch.empty(M, Cc, device="cuda"); z = torch.empty(M, Cc, device="cuda", dtype=torch.bfloat16)
def run():
    ops.convmlp_fwd(ln, w1, b1, w2, b2, gamma, x, out, z=z)
for _ in range(3): run()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(10): run()
e1.record(); torch.cuda.synchronize()
print("convmlp_fwd C=192 us:", e0.elapsed_time(e1) * 100)
o8 = (C.c_ulonglong * 8)()
try:
    L.lib().lnx_dbg_convmlp_stamps(o8)
except AttributeError:
    sys.exit(0)  # not a stamp build: timing only
names = ["prologue", "dma wait", "barrier", "dma issue", "prod1|matrix", "gelu|valu", "prod2", "epilogue"]
tot = sum(o8)
print("wave 0: total clk", tot, " ".join(f"{n} {o8[i] / tot * 100:.1f}%" for i, n in enumerate(names)))
