"""The metadata-head chains alone on the chip (metahead.hip): lnx_meta_heads_fwd / lnx_meta_heads_bwd at the bench shapes, event-timed per call,
against the launch-by-launch numbers of rounds 1-4 (2.6 ms of side-stream kernel time per sm step).  usage: python tools/bench_meta.py [--batch 256]"""
import argparse
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from linnaeus_amd import _lib as L  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=256)
a_ = ap.parse_args()
B = a_.batch
lib = L.lib()
st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
p_ = lambda t: C.c_void_p(t.data_ptr())  # noqa: E731
g = torch.Generator(device="cuda").manual_seed(1)
r = lambda *s, sc=1.0: torch.randn(*s, device="cuda", generator=g) * sc  # noqa: E731
keep = []


def head(C_, dim, off, N, slot, meta, tok, gtok):
    w0 = torch.zeros(C_, 16, device="cuda")
    w0[:, :dim] = r(C_, dim)
    P = {k: r(C_, sc=0.3) for k in ("b0", "b1", "b2", "ln0_b", "ln1_b", "ln2_b")}
    P.update({k: 1 + r(C_, sc=0.1) for k in ("ln0_w", "ln1_w", "ln2_w")})
    P["w1"], P["w2"] = r(C_, C_, sc=C_ ** -0.5), r(C_, C_, sc=C_ ** -0.5)
    bufs = {k: torch.empty(B, C_, device="cuda") for k in ("h0", "x", "h1", "n1", "h2")}
    bufs["t0"] = torch.empty(B, 16, device="cuda")
    bufs.update({k: torch.empty(B, device="cuda") for k in ("m0", "r0", "m1", "r1", "m2", "r2")})
    f = L.MetaHeadArgs()
    f.B, f.C, f.dim, f.off, f.meta, f.meta_width, f.eps = B, C_, dim, off, p_(meta), meta.shape[1], 1e-5
    f.w0, f.ldw0, f.b0, f.ln0_w, f.ln0_b = p_(w0), 16, p_(P["b0"]), p_(P["ln0_w"]), p_(P["ln0_b"])
    f.w1, f.ldw1, f.b1, f.ln1_w, f.ln1_b = p_(P["w1"]), C_, p_(P["b1"]), p_(P["ln1_w"]), p_(P["ln1_b"])
    f.w2, f.ldw2, f.b2, f.ln2_w, f.ln2_b = p_(P["w2"]), C_, p_(P["b2"]), p_(P["ln2_w"]), p_(P["ln2_b"])
    for k, v in bufs.items():
        setattr(f, k, p_(v))
    f.tok, f.tok_row_stride, f.tok_row_offset = p_(tok), N * C_, slot * C_
    w1t, w2t = P["w1"].t().contiguous(), P["w2"].t().contiguous()
    dp = [torch.empty(B, C_, device="cuda") for _ in range(3)]
    part = torch.empty(lib.lnx_meta_heads_bwd_part_floats(B, C_), device="cuda")
    grads = {k: torch.zeros_like(v) for k, v in P.items()}
    grads["w0"] = torch.zeros(C_, dim, device="cuda")
    b = L.MetaHeadBwdArgs()
    b.B, b.C, b.dim, b.g, b.g_row_stride, b.g_row_offset = B, C_, dim, p_(gtok), N * C_, slot * C_
    b.w1t, b.ldw1t, b.w2t, b.ldw2t, b.ln0_w, b.ln1_w, b.ln2_w = p_(w1t), C_, p_(w2t), C_, p_(P["ln0_w"]), p_(P["ln1_w"]), p_(P["ln2_w"])
    for k, v in bufs.items():
        setattr(b, k, p_(v))
    b.dp2, b.dp1, b.dp0, b.part = p_(dp[0]), p_(dp[1]), p_(dp[2]), p_(part)
    for k in ("w0", "b0", "ln0_w", "ln0_b", "w1", "b1", "ln1_w", "ln1_b", "w2", "b2", "ln2_w", "ln2_b"):
        setattr(b, "d_" + k, p_(grads[k]))
    keep.extend([w0, P, bufs, w1t, w2t, dp, part, grads])
    return f, b


meta = torch.rand(B, 5, device="cuda")
fw, bw = [], []
for C_, N in ((384, 199), (768, 52)):
    tok, gtok = torch.zeros(B, N, C_, device="cuda"), torch.randn(B, N, C_, device="cuda")
    keep.extend([tok, gtok])
    for m, (dim, off) in enumerate(((2, 0), (3, 2))):
        f, b = head(C_, dim, off, N, 1 + m, meta, tok, gtok)
        fw.append(f)
        bw.append(b)
F = (L.MetaHeadArgs * 4)(*fw)
Bw3 = (L.MetaHeadBwdArgs * 2)(*bw[:2])
Bw4 = (L.MetaHeadBwdArgs * 2)(*bw[2:])


def timed(fn, n=50):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


t_f = timed(lambda: L.check(lib.lnx_meta_heads_fwd(F, 4, st), "fwd"))
t_b3 = timed(lambda: L.check(lib.lnx_meta_heads_bwd(Bw3, 2, st), "bwd3"))
t_b4 = timed(lambda: L.check(lib.lnx_meta_heads_bwd(Bw4, 2, st), "bwd4"))
print(f"batch {B}: forward (4 heads, 2 launches) {t_f:.1f} us; backward stage 3 (C = 384: chain + weight gradients) {t_b3:.1f} us; stage 4 (C = 768) {t_b4:.1f} us; "
      f"total {t_f + t_b3 + t_b4:.1f} us per step, the chip otherwise idle")
