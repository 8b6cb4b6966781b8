"""Diagnostic: cycle shares of the resident attention dk/dv kernel (needs tools/libstamp.so via LNX_LIB_PATH)."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
from linnaeus_amd import ops, _lib as L

B, N, E, heads = 256, 197, 1, 6
Cc = heads * 64
qkv = torch.randn(B * N, 3 * Cc, device="cuda").bfloat16()
freqs = torch.randn(2, heads, 32, device="cuda")
cos = ops.rope_cos_table(freqs, 14, 14)
o = torch.empty(B * N, Cc, device="cuda", dtype=torch.bfloat16)
lse = torch.empty(B * heads * N, device="cuda")
ops.attn_fwd(qkv, cos, o, lse, B, N, E, heads)
do = torch.randn_like(o)
dqkv = torch.empty_like(qkv)
dsin = torch.empty(2, H * W, heads, 32, device="cuda"); cos = ops.rope_cos_table(freqs, H, W, dsin=dsin); dfreqs = torch.zeros(2, heads, 32, device="cuda")
delta = torch.empty_like(lse)
for _ in range(3):
    ops.attn_bwd(qkv, cos, o, lse, do, dqkv, delta, B, N, E, heads, dsin=dsin, dfreqs=dfreqs)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(10):
    ops.attn_bwd(qkv, cos, o, lse, do, dqkv, delta, B, N, E, heads, dsin=dsin, dfreqs=dfreqs)
e1.record(); torch.cuda.synchronize()
print("attn_bwd (dq + dkv) us:", e0.elapsed_time(e1) * 100)
out = (C.c_ulonglong * 8)()
L.lib().lnx_dbg_attn_stamps(out)
names = ["stage", "stats+barrier", "tile fetch", "q loop", "epilogue"]
tot = sum(out[:5])
print("dkv wave 0: total clk", tot, " ".join(f"{n} {out[i] / tot * 100:.1f}%" for i, n in enumerate(names)))
