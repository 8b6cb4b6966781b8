"""Micro-benchmarks of the HBM-bound kernels at the mFormerV1_sm B=256 shapes (achieved GB/s of
algorithmic bytes)."""
import sys
import torch
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from linnaeus_amd import ops

def time_it(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e-3

bf, f32 = torch.bfloat16, torch.float32
which = sys.argv[1] if len(sys.argv) > 1 else "all"
if which in ("all", "ln"):
    for name, M, C, xd, yd in [("s0.blockLN", 802816, 96, bf, bf), ("s1.blockLN", 200704, 192, bf, bf), ("r0.norm", 50944, 384, f32, bf),
                               ("r1.norm", 13312, 768, f32, bf), ("stem.LN", 802816, 96, bf, f32), ("ds0.LN", 802816, 96, f32, bf)]:
        x = torch.randn(M, C, device="cuda").to(xd); y = torch.empty(M, C, device="cuda", dtype=yd)
        w = torch.ones(C, device="cuda"); b = torch.zeros(C, device="cuda")
        mean = torch.empty(M, device="cuda"); rstd = torch.empty(M, device="cuda")
        t = time_it(lambda: ops.layernorm_fwd(x, w, b, y, 1e-6, mean=mean, rstd=rstd))
        by = M * C * (x.element_size() + y.element_size())
        dy = torch.randn(M, C, device="cuda").to(yd); dx = torch.empty(M, C, device="cuda", dtype=xd)
        dw = torch.zeros(C, device="cuda"); db = torch.zeros(C, device="cuda")
        gin = torch.randn(M, C, device="cuda") if xd == f32 else None
        ws = torch.empty(2048 * 2 * C, device='cuda')
        t2 = time_it(lambda: ops.layernorm_bwd(dy, x, w, mean, rstd, dx, gin=gin, dw=dw, db=db, ws=ws))
        by2 = M * C * (dy.element_size() + x.element_size() + dx.element_size() + (4 if gin is not None else 0))
        t3 = time_it(lambda: ops.layernorm_bwd(dy, x, w, mean, rstd, dx, gin=gin))
        print(f"   (bwd without dw/db: {t3*1e6:7.1f}us {by2/t3/1e9:7.0f} GB/s)")
        print(f"LN {name:11s} M={M:7d} C={C:4d} fwd {t*1e6:7.1f}us {by/t/1e9:7.0f} GB/s | bwd {t2*1e6:7.1f}us {by2/t2/1e9:7.0f} GB/s", flush=True)
if which in ("all", "dw"):
    for name, B, H, C in [("s0", 256, 56, 96), ("s1", 256, 28, 192)]:
        x = torch.randn(B, H, H, C, device="cuda"); y = torch.empty(B, H, H, C, device="cuda", dtype=bf)
        w49 = torch.randn(49, C, device="cuda"); bias = torch.randn(C, device="cuda")
        t = time_it(lambda: ops.dwconv7(x, w49, bias, y))
        n = B * H * H * C
        dyb = torch.randn(B, H, H, C, device="cuda").to(bf); g = torch.randn(B, H, H, C, device="cuda")
        t2 = time_it(lambda: ops.dwconv7(dyb, w49, None, g, flip=True, res=g))
        dw = torch.zeros(C, 1, 7, 7, device="cuda"); db = torch.zeros(C, device="cuda")
        t3 = time_it(lambda: ops.dwconv7_wgrad(x, dyb, dw, db))
        print(f"dwconv {name} fwd {t*1e6:7.1f}us {n*6/t/1e9:6.0f} GB/s | dgrad {t2*1e6:7.1f}us {n*10/t2/1e9:6.0f} GB/s | wgrad {t3*1e6:7.1f}us {n*6/t3/1e9:6.0f} GB/s", flush=True)
if which in ("all", "ew"):
    M, C = 802816, 96
    g = torch.randn(M, C, device="cuda"); z = torch.randn(M, C, device="cuda").to(bf); gam = torch.randn(C, device="cuda")
    rs = torch.ones(256, device="cuda"); dz = torch.empty(M, C, device="cuda", dtype=bf); dg = torch.zeros(C, device="cuda")
    t = time_it(lambda: ops.layerscale_bwd(g, z, gam, rs, 3136, dz, dg, M, C))
    print(f"layerscale_bwd s0 {t*1e6:7.1f}us {M*C*8/t/1e9:6.0f} GB/s")
    out = torch.empty(M, C, device="cuda", dtype=bf)
    t = time_it(lambda: ops.scale_cast(g, out, M, C, rowscale=rs, rows_per_sample=3136))
    print(f"scale_cast s0 {t*1e6:7.1f}us {M*C*6/t/1e9:6.0f} GB/s")
if which in ("all", "cm"):
    for name, M, C in [("s0", 802816, 96), ("s1", 200704, 192)]:
        gen = torch.Generator().manual_seed(0)
        ln = torch.randn(M, C, device="cuda").to(bf); w1 = (torch.randn(4 * C, C, device="cuda") / C**0.5).to(bf); b1 = torch.zeros(4 * C, device="cuda")
        w2 = (torch.randn(C, 4 * C, device="cuda") / (4 * C)**0.5).to(bf); b2 = torch.zeros(C, device="cuda"); gam = torch.ones(C, device="cuda")
        x = torch.randn(M, C, device="cuda"); out = torch.empty(M, C, device="cuda"); z = torch.empty(M, C, device="cuda", dtype=bf)
        rs = torch.ones(256, device="cuda")
        t = time_it(lambda: ops.convmlp_fwd(ln, w1, b1, w2, b2, gam, x, out, rowscale=rs, rows_per_sample=M // 256, z=z))
        fl = 2.0 * M * C * 4 * C * 2
        by = M * C * (2 + 4 + 4 + 2)
        g = torch.randn(M, C, device="cuda"); act = torch.empty(M, 4 * C, device="cuda", dtype=bf); dh = torch.empty_like(act)
        dz = torch.empty(M, C, device="cuda", dtype=bf); dln = torch.empty_like(dz); dg = torch.zeros(C, device="cuda")
        w2t = w2.t().contiguous(); w1t = w1.t().contiguous()
        t2 = time_it(lambda: ops.convmlp_bwd(g, ln, z, w1, b1, w2t, w1t, gam, act, dh, dz, dln, dg, rowscale=rs, rows_per_sample=M // 256))
        by2 = M * C * (4 + 2 + 2 + 2 + 2) + 2 * M * 4 * C * 2
        print(f"convmlp {name} fwd {t*1e6:7.1f}us {fl/t/1e12:6.0f} TF/s {by/t/1e9:6.0f} GB/s | bwd {t2*1e6:7.1f}us {1.5*fl/t2/1e12:6.0f} TF/s {by2/t2/1e9:6.0f} GB/s", flush=True)
        # the same with the block LayerNorm inside the kernels (what the plan launches)
        y = torch.randn(M, C, device="cuda").to(bf); lw = torch.ones(C, device="cuda"); lb = torch.zeros(C, device="cuda")
        mean = torch.empty(M, device="cuda"); rstd = torch.empty(M, device="cuda"); dlw = torch.zeros(C, device="cuda"); dlb = torch.zeros(C, device="cuda")
        ws = torch.empty(max(256, (M + 127) // 128) * 2 * C, device="cuda")
        t3 = time_it(lambda: ops.convmlp_fwd(None, w1, b1, w2, b2, gam, x, out, rowscale=rs, rows_per_sample=M // 256, z=z, y=y, ln_w=lw, ln_b=lb, ln_out=ln,
                                             mean=mean, rstd=rstd))
        t4 = time_it(lambda: ops.convmlp_bwd(g, ln, z, w1, b1, w2t, w1t, gam, act, dh, dz, dln, dg, rowscale=rs, rows_per_sample=M // 256, y=y, ln_w=lw,
                                             mean=mean, rstd=rstd, d_ln_w=dlw, d_ln_b=dlb, ws=ws))
        print(f"convmlp {name} +LN  fwd {t3*1e6:7.1f}us | bwd {t4*1e6:7.1f}us", flush=True)
