"""Per-op parity of the HIP kernels (through the C ABI) against the CPU oracle's functions /
plain PyTorch fp32-fp64 on the same seeded inputs.  fp32 mode: tight tolerances; bf16 mode:
tolerance of the bf16 storage rounding, stated per test."""
import numpy as np
import pytest
import torch

from linnaeus_amd import _lib as L
from linnaeus_amd import ops
from oracle import mformer_oracle as O

pytestmark = pytest.mark.gpu
DT = {L.F32: torch.float32, L.BF16: torch.bfloat16}


def g(seed):
    return torch.Generator().manual_seed(seed)


@pytest.mark.parametrize("C_", [32, 96, 192, 384, 768, 1024, 1280, 1536, 2048])
@pytest.mark.parametrize("xd,yd", [(L.F32, L.F32), (L.F32, L.BF16), (L.BF16, L.BF16), (L.BF16, L.F32)])
def test_layernorm_fwd_bwd(C_, xd, yd):
    M = 77
    gen = g(C_ + xd * 3 + yd)
    x = (torch.randn(M, C_, generator=gen) * 2 + 0.5).cuda().to(DT[xd])
    w = (1 + 0.2 * torch.randn(C_, generator=gen)).cuda()
    b = (0.1 * torch.randn(C_, generator=gen)).cuda()
    y = torch.empty(M, C_, device="cuda", dtype=DT[yd])
    mean = torch.empty(M, device="cuda")
    rstd = torch.empty(M, device="cuda")
    ops.layernorm_fwd(x, w, b, y, 1e-6, mean=mean, rstd=rstd)
    xr = x.double().cpu().requires_grad_(True)
    wr = w.double().cpu().requires_grad_(True)
    br = b.double().cpu().requires_grad_(True)
    ref = O.layer_norm_last(xr, wr, br, 1e-6)
    tol = 2e-5 if yd == L.F32 else 1.6e-2
    torch.testing.assert_close(y.double().cpu(), ref.detach(), rtol=tol, atol=tol)
    # backward
    dy = torch.randn(M, C_, generator=gen).cuda().to(DT[yd])
    gin = torch.randn(M, C_, generator=gen).cuda()
    dx = torch.empty(M, C_, device="cuda")
    dw = torch.zeros(C_, device="cuda")
    db = torch.zeros(C_, device="cuda")
    ops.layernorm_bwd(dy, x, w, mean, rstd, dx, gin=gin, dw=dw, db=db)
    # same thing through the partial-sum workspace (two-kernel column reduction, no contended atomics)
    dx2 = torch.empty(M, C_, device="cuda")
    dw2 = torch.zeros(C_, device="cuda")
    db2 = torch.zeros(C_, device="cuda")
    ops.layernorm_bwd(dy, x, w, mean, rstd, dx2, gin=gin, dw=dw2, db=db2, ws=torch.empty(2048 * 2 * C_, device="cuda"))
    torch.testing.assert_close(dx2, dx, rtol=0, atol=0)
    torch.testing.assert_close(dw2, dw, rtol=1e-5, atol=1e-5)
    torch.testing.assert_close(db2, db, rtol=1e-5, atol=1e-5)
    ref.backward(dy.double().cpu())
    torch.testing.assert_close(dx.double().cpu(), xr.grad + gin.double().cpu(), rtol=1e-4, atol=1e-4)
    torch.testing.assert_close(dw.double().cpu(), wr.grad, rtol=1e-4, atol=2e-4)
    torch.testing.assert_close(db.double().cpu(), br.grad, rtol=1e-4, atol=2e-4)


def test_layernorm_rowmaps_add_relu():
    B, HW, E, C_ = 3, 5, 2, 64
    N = HW + E
    gen = g(9)
    tok = torch.randn(B * N, C_, generator=gen).cuda()
    w = (1 + 0.2 * torch.randn(C_, generator=gen)).cuda()
    b = (0.1 * torch.randn(C_, generator=gen)).cuda()
    # read patch rows of a token buffer, write compact
    y = torch.empty(B * HW, C_, device="cuda")
    ops.layernorm_fwd(tok, w, b, y, 1e-5, M=B * HW, x_map=(HW, E, E))
    ref = O.layer_norm_last(tok.view(B, N, C_)[:, E:].reshape(B * HW, C_).cpu(), w.cpu(), b.cpu(), 1e-5)
    torch.testing.assert_close(y.cpu(), ref, rtol=2e-5, atol=2e-5)
    # CLS rows only, with an added skip tensor, written into row 1 of each sample
    add = torch.randn(B, C_, generator=gen).cuda()
    out = torch.zeros(B * N, C_, device="cuda")
    ops.layernorm_fwd(tok, w, b, out, 1e-5, M=B, x_map=(1, N - 1, 0), y_map=(1, N - 1, 1), add=add)
    ref = O.layer_norm_last(tok.view(B, N, C_)[:, 0].cpu(), w.cpu(), b.cpu(), 1e-5) + add.cpu()
    torch.testing.assert_close(out.view(B, N, C_)[:, 1].cpu(), ref, rtol=2e-5, atol=2e-5)
    assert out.view(B, N, C_)[:, 0].abs().sum().item() == 0
    # relu-mask backward
    x = torch.relu(torch.randn(B, C_, generator=gen)).cuda()
    mean = torch.empty(B, device="cuda")
    rstd = torch.empty(B, device="cuda")
    yy = torch.empty(B, C_, device="cuda")
    ops.layernorm_fwd(x, w, b, yy, 1e-5, mean=mean, rstd=rstd)
    dy = torch.randn(B, C_, generator=gen).cuda()
    dx = torch.empty(B, C_, device="cuda")
    ops.layernorm_bwd(dy, x, w, mean, rstd, dx, relu_mask=True)
    pre = torch.randn(B, C_).double()
    pre = (x.double().cpu() > 0).double() * x.double().cpu() - (x.double().cpu() <= 0).double()  # pre-activation with same mask
    pre.requires_grad_(True)
    O.layer_norm_last(torch.relu(pre), w.double().cpu(), b.double().cpu(), 1e-5).backward(dy.double().cpu())
    torch.testing.assert_close(dx.double().cpu(), pre.grad, rtol=1e-4, atol=1e-4)


def _w49(w):  # [C,1,7,7] -> [49][C]
    return w.reshape(w.shape[0], 49).t().contiguous()


@pytest.mark.parametrize("B,H,W,C_", [(2, 6, 5, 32), (1, 16, 16, 64), (2, 28, 28, 96), (1, 56, 56, 32), (3, 9, 23, 64)])
@pytest.mark.parametrize("xd,yd", [(L.F32, L.F32)])
def test_dwconv(B, H, W, C_, xd, yd):
    gen = g(B * H + W + C_)
    x = torch.randn(B, H, W, C_, generator=gen).cuda().to(DT[xd])
    w = (torch.randn(C_, 1, 7, 7, generator=gen) / 7).cuda()
    bias = torch.randn(C_, generator=gen).cuda()
    y = torch.empty(B, H, W, C_, device="cuda", dtype=DT[yd])
    ops.dwconv7(x, _w49(w), bias, y)
    xr = x.double().cpu().permute(0, 3, 1, 2).requires_grad_(True)
    wr = w.double().cpu().requires_grad_(True)
    br = bias.double().cpu().requires_grad_(True)
    ref = O.depthwise_conv7(xr, wr, br)
    tol = 2e-5 if yd == L.F32 else 1.6e-2
    torch.testing.assert_close(y.double().cpu(), ref.detach().permute(0, 2, 3, 1), rtol=tol, atol=tol)
    # data gradient (flip) with residual add, and weight gradient
    dy = torch.randn(B, H, W, C_, generator=gen).cuda().to(DT[xd])
    res = torch.randn(B, H, W, C_, generator=gen).cuda()
    dx = torch.empty(B, H, W, C_, device="cuda")
    ops.dwconv7(dy, _w49(w), None, dx, flip=True, res=res)
    ref.backward(dy.double().cpu().permute(0, 3, 1, 2))
    torch.testing.assert_close(dx.double().cpu(), xr.grad.permute(0, 2, 3, 1) + res.double().cpu(), rtol=1e-4, atol=1e-4)
    dw = torch.zeros(C_, 1, 7, 7, device="cuda")
    db = torch.zeros(C_, device="cuda")
    ops.dwconv7_wgrad(x, dy, dw, db)
    sc = (B * H * W) ** 0.5
    torch.testing.assert_close(dw.double().cpu(), wr.grad, rtol=1e-4, atol=2e-5 * sc)
    torch.testing.assert_close(db.double().cpu(), br.grad, rtol=1e-4, atol=2e-5 * sc)


@pytest.mark.parametrize("B,H,W,C_", [(2, 6, 5, 32), (1, 16, 16, 64), (2, 28, 28, 96), (1, 56, 56, 32), (3, 9, 23, 64), (2, 30, 57, 32), (5, 14, 28, 128)])
def test_dwconv_mfma(B, H, W, C_):
    """bf16 compute: the matrix-core kernels (csrc/dwconv_mfma.hip).  Semantics of conv2d under autocast
    (blocks/convnext.py:56-58 inside train.py's autocast): operands rounded to bf16, fp32 accumulation -- the
    reference is the fp64 oracle on the bf16-rounded operands, so the tolerances stay those of the fp32 kernels."""
    gen = g(7 * B + H * W + C_)
    r16 = lambda t: t.bfloat16().float()
    x = torch.randn(B, H, W, C_, generator=gen).cuda()
    w = (torch.randn(C_, 1, 7, 7, generator=gen) / 7).cuda()
    bias = torch.randn(C_, generator=gen).cuda()
    dy = torch.randn(B, H, W, C_, generator=gen).cuda().bfloat16()
    res = torch.randn(B, H, W, C_, generator=gen).cuda()
    nhwc = lambda t: t.permute(0, 2, 3, 1)

    def oracle(xin):
        xr = r16(xin).double().cpu().permute(0, 3, 1, 2).requires_grad_(True)
        wr = r16(w).double().cpu().requires_grad_(True)
        br = bias.double().cpu().requires_grad_(True)
        ref = O.depthwise_conv7(xr, wr, br)
        ref.backward(dy.double().cpu().permute(0, 3, 1, 2))
        return ref.detach(), xr.grad, wr.grad, br.grad

    ref, gx, gw, gb = oracle(x)
    # forward: fp32 residual stream in, bf16 out (plan.cpp conv_block_fwd); bf16 in / out
    y = torch.empty(B, H, W, C_, device="cuda", dtype=torch.bfloat16)
    ops.dwconv7(x, _w49(w), bias, y)
    torch.testing.assert_close(y.double().cpu(), nhwc(ref), rtol=8e-3, atol=8e-3)
    y2 = torch.empty_like(y)
    ops.dwconv7(x.bfloat16(), _w49(w), bias, y2)
    assert torch.equal(y, y2)
    yf = torch.empty(B, H, W, C_, device="cuda")
    ops.dwconv7(x.bfloat16(), _w49(w), bias, yf)
    torch.testing.assert_close(yf.double().cpu(), nhwc(ref), rtol=1e-4, atol=1e-4)
    # data gradient: bf16 dy, fp32 residual updated in place (plan.cpp conv_block_bwd)
    gbuf = res.clone()
    ops.dwconv7(dy, _w49(w), None, gbuf, flip=True, res=gbuf)
    torch.testing.assert_close(gbuf.double().cpu(), nhwc(gx) + res.double().cpu(), rtol=1e-4, atol=1e-4)
    # weight / bias gradient: fp32 x + bf16 dy (production), bf16 x + bf16 dy
    sc = (B * H * W) ** 0.5
    for xin in (x, x.bfloat16()):
        dw = torch.zeros(C_, 1, 7, 7, device="cuda")
        db = torch.zeros(C_, device="cuda")
        ops.dwconv7_wgrad(xin, dy, dw, db)
        torch.testing.assert_close(dw.double().cpu(), gw, rtol=1e-4, atol=2e-5 * sc)
        torch.testing.assert_close(db.double().cpu(), gb, rtol=1e-4, atol=2e-5 * sc)


def _attn_ref(qkv, freqs, B, N, E, heads, H, W, drop=None):
    """Oracle math of RoPE2DAttention between the qkv Linear and the proj Linear; drop = attn_drop multiplier [B, h, N, N]."""
    C_ = heads * 64
    t = qkv.reshape(B, N, 3, heads, 64).permute(2, 0, 3, 1, 4)
    q, k, v = t[0], t[1], t[2]
    cos = O.rope_cos_table(freqs, H, W).to(qkv.dtype)
    q = torch.cat([q[:, :, :E], O.rope_scale_pairs(q[:, :, E:], cos)], 2) * 0.125
    k = torch.cat([k[:, :, :E], O.rope_scale_pairs(k[:, :, E:], cos)], 2)
    a = torch.softmax(q @ k.transpose(-2, -1), -1)
    if drop is not None:
        a = a * drop
    return (a @ v).transpose(1, 2).reshape(B * N, C_)


def test_rope_cos_tables_of_many_blocks_in_one_launch():
    """lnx_rope_cos_tables (what a plan calls once per forward) writes, for each entry, exactly what lnx_rope_cos_table writes for
    it alone: mixed grids and head counts, with and without the d-cos table, and more entries than one launch carries."""
    shapes = [(6, 14, 14), (12, 7, 7), (3, 5, 9), (6, 14, 14)] * 7  # 28 > LNX_ROPE_TABLES_MAX (24): two launches
    entries, want = [], []
    for i, (heads, H, W) in enumerate(shapes):
        freqs = O.seeded_fill(f"t.cos_tables.{i}", (2, heads, 32), 3 + i).cuda()
        with_dsin = i % 3 != 1
        out = torch.full((H * W, heads, 32), float("nan"), device="cuda")
        dsin = torch.full((2, H * W, heads, 32), float("nan"), device="cuda") if with_dsin else None
        entries.append((freqs, H, W, out, dsin))
        d1 = torch.empty(2, H * W, heads, 32, device="cuda") if with_dsin else None
        want.append((ops.rope_cos_table(freqs, H, W, dsin=d1), d1))
    ops.rope_cos_tables(entries)
    torch.cuda.synchronize()
    for (_, _, _, out, dsin), (c1, d1) in zip(entries, want):
        assert torch.equal(out, c1)
        if dsin is not None:
            assert torch.equal(dsin, d1)


@pytest.mark.parametrize("B,heads,H,W,E", [(2, 2, 3, 5, 3), (1, 6, 14, 14, 3), (2, 4, 7, 7, 3), (1, 2, 12, 12, 4), (2, 1, 2, 2, 1), (1, 2, 24, 24, 4), (2, 2, 20, 20, 4)])
@pytest.mark.parametrize("dtype", [L.F32, L.BF16])
def test_attention_fwd_bwd(B, heads, H, W, E, dtype):
    N = H * W + E
    C_ = heads * 64
    gen = g(B + heads * 5 + N)
    qkv = torch.randn(B * N, 3 * C_, generator=gen).cuda().to(DT[dtype])
    freqs = O.seeded_fill("t.attn.freqs", (2, heads, 32), 7).cuda()
    dsin = torch.empty(2, H * W, heads, 32, device="cuda")
    cos = ops.rope_cos_table(freqs, H, W, dsin=dsin)
    torch.testing.assert_close(cos.cpu(), O.rope_cos_table(freqs.cpu(), H, W), rtol=0, atol=2e-6)
    fd = freqs.double().cpu().requires_grad_(True)  # the second table is d cos(theta) / d freqs[a, h, j], entry by entry
    O.rope_cos_table(fd, H, W).sum().backward()
    torch.testing.assert_close(dsin.double().cpu().sum(1), fd.grad, rtol=1e-5, atol=1e-4)
    o = torch.empty(B * N, C_, device="cuda", dtype=DT[dtype])
    lse = torch.empty(B, heads, N, device="cuda")
    ops.attn_fwd(qkv, cos, o, lse, B, N, E, heads)
    qr = qkv.double().cpu().requires_grad_(True)
    fr = freqs.double().cpu().requires_grad_(True)
    ref = _attn_ref(qr, fr, B, N, E, heads, H, W)
    tol = 3e-5 if dtype == L.F32 else 2e-2
    torch.testing.assert_close(o.double().cpu(), ref.detach(), rtol=tol, atol=tol)
    # backward
    d_o = torch.randn(B * N, C_, generator=gen).cuda().to(DT[dtype])
    dqkv = torch.full((B * N, 3 * C_), float("nan"), device="cuda", dtype=DT[dtype])
    delta = torch.empty(B, heads, N, device="cuda")
    dfreqs = torch.ones(2, heads, 32, device="cuda")  # accumulated into: starts at 1
    ops.attn_bwd(qkv, cos, o, lse, d_o, dqkv, delta, B, N, E, heads, dsin=dsin, dfreqs=dfreqs)
    dfreqs -= 1.0
    ref.backward(d_o.double().cpu())
    tolb = 1e-4 if dtype == L.F32 else 4e-2
    torch.testing.assert_close(dqkv.double().cpu(), qr.grad, rtol=tolb, atol=tolb)
    scale = fr.grad.abs().max().item()
    torch.testing.assert_close(dfreqs.double().cpu(), fr.grad, rtol=tolb, atol=tolb * max(scale, 1.0))


def test_attention_bwd_postponed_freqs_folds_in_one_launch():
    """lnx_attn_bwd_args.defer_freqs + lnx_attn_bwd_flush (what a plan does with the RoPE blocks of a backward segment): the
    freqs gradients of several calls -- different sequence lengths, head counts, kernels (resident, tiled, 8-wave tiled, fp32) --
    folded by ONE launch are what each call folds on its own (same fold order; the partials agree run to run up to the rounding of
    the kernels' LDS float atomics); nothing is written before the flush; a discard drops the pending folds."""
    cases = [(2, 2, 3, 5, 3, L.BF16), (1, 3, 14, 14, 3, L.BF16), (1, 2, 24, 24, 4, L.BF16), (2, 6, 14, 14, 3, L.BF16), (1, 2, 7, 7, 4, L.F32)]
    runs = []
    for i, (B, heads, H, W, E, dtype) in enumerate(cases):
        N, C_ = H * W + E, heads * 64
        gen = g(100 + i)
        qkv = torch.randn(B * N, 3 * C_, generator=gen).cuda().to(DT[dtype])
        freqs = O.seeded_fill(f"t.attn.defer.{i}", (2, heads, 32), 7 + i).cuda()
        dsin = torch.empty(2, H * W, heads, 32, device="cuda")
        cos = ops.rope_cos_table(freqs, H, W, dsin=dsin)
        o = torch.empty(B * N, C_, device="cuda", dtype=DT[dtype])
        lse = torch.empty(B, heads, N, device="cuda")
        ops.attn_fwd(qkv, cos, o, lse, B, N, E, heads)
        d_o = torch.randn(B * N, C_, generator=gen).cuda().to(DT[dtype])
        runs.append(dict(args=(qkv, cos, o, lse, d_o), shape=(B, N, E, heads), dsin=dsin))
    for r in runs:  # each call alone
        B, N, E, heads = r["shape"]
        r["want"] = torch.ones(2, heads, 32, device="cuda")
        ops.attn_bwd(*r["args"], torch.empty_like(r["args"][0]), torch.empty(B, heads, N, device="cuda"), B, N, E, heads, dsin=r["dsin"], dfreqs=r["want"])
    keep = []
    for r in runs:  # postponed
        B, N, E, heads = r["shape"]
        r["got"] = torch.ones(2, heads, 32, device="cuda")
        dq, dl = torch.empty_like(r["args"][0]), torch.empty(B, heads, N, device="cuda")
        keep.append((dq, dl, ops.attn_bwd(*r["args"], dq, dl, B, N, E, heads, dsin=r["dsin"], dfreqs=r["got"], defer_freqs=True)))
    torch.cuda.synchronize()
    assert all(bool((r["got"] == 1).all()) for r in runs)  # nothing folded yet
    ops.attn_bwd_flush()
    torch.cuda.synchronize()
    for r in runs:
        # (the per-workgroup partials themselves carry LDS float atomics: two runs of one call agree to rounding, not to the bit)
        assert not bool((r["got"] == 1).all())
        torch.testing.assert_close(r["got"], r["want"], rtol=1e-5, atol=1e-5 * float(r["want"].abs().max()))
    # error path: pending folds are dropped, the targets stay as they are
    B, N, E, heads = runs[0]["shape"]
    t = torch.ones(2, heads, 32, device="cuda")
    dq, dl = torch.empty_like(runs[0]["args"][0]), torch.empty(B, heads, N, device="cuda")
    ws = ops.attn_bwd(*runs[0]["args"], dq, dl, B, N, E, heads, dsin=runs[0]["dsin"], dfreqs=t, defer_freqs=True)
    assert L.lib().lnx_attn_bwd_discard() == 1
    ops.attn_bwd_flush()  # nothing pending: no launch
    torch.cuda.synchronize()
    assert bool((t == 1).all()) and ws is not None


@pytest.mark.parametrize("B,heads,H,W,E", [(2, 2, 3, 5, 3), (1, 3, 14, 14, 3), (1, 2, 24, 24, 4)])
@pytest.mark.parametrize("dtype", [L.F32, L.BF16])
def test_attention_probability_dropout(B, heads, H, W, E, dtype):
    """attn_drop (rope_2d_mhsa.py:497): the probabilities are multiplied by keep / (1 - p) AFTER the softmax normalisation, in
    forward and backward, with the caller's keep mask [B, heads, N, Np]; one-, four- and ten-tile sequences."""
    N = H * W + E
    Np = (N + 63) // 64 * 64
    C_ = heads * 64
    rate = 0.25
    gen = g(B + heads * 3 + N)
    qkv = torch.randn(B * N, 3 * C_, generator=gen).cuda().to(DT[dtype])
    freqs = O.seeded_fill("t.attn.freqs", (2, heads, 32), 7).cuda()
    dsin = torch.empty(2, H * W, heads, 32, device="cuda")
    cos = ops.rope_cos_table(freqs, H, W, dsin=dsin)
    mask = (torch.rand(B, heads, N, Np, generator=gen) >= rate).to(torch.uint8).cuda()
    o = torch.empty(B * N, C_, device="cuda", dtype=DT[dtype])
    lse = torch.empty(B, heads, N, device="cuda")
    ops.attn_fwd(qkv, cos, o, lse, B, N, E, heads, drop_mask=mask, drop_rate=rate)
    qr = qkv.double().cpu().requires_grad_(True)
    fr = freqs.double().cpu().requires_grad_(True)
    mult = mask[..., :N].double().cpu() / (1.0 - rate)
    ref = _attn_ref(qr, fr, B, N, E, heads, H, W, mult)
    tol = 3e-5 if dtype == L.F32 else 2e-2
    torch.testing.assert_close(o.double().cpu(), ref.detach(), rtol=tol, atol=tol)
    d_o = torch.randn(B * N, C_, generator=gen).cuda().to(DT[dtype])
    dqkv = torch.full((B * N, 3 * C_), float("nan"), device="cuda", dtype=DT[dtype])
    delta = torch.empty(B, heads, N, device="cuda")
    dfreqs = torch.zeros(2, heads, 32, device="cuda")
    ops.attn_bwd(qkv, cos, o, lse, d_o, dqkv, delta, B, N, E, heads, dsin=dsin, dfreqs=dfreqs, drop_mask=mask, drop_rate=rate)
    ref.backward(d_o.double().cpu())
    tolb = 1e-4 if dtype == L.F32 else 4e-2
    torch.testing.assert_close(dqkv.double().cpu(), qr.grad, rtol=tolb, atol=tolb)
    scale = fr.grad.abs().max().item()
    torch.testing.assert_close(dfreqs.double().cpu(), fr.grad, rtol=tolb, atol=tolb * max(scale, 1.0))


def test_attention_spiked_softmax():
    """A key that dominates one query row late in the sequence forces the online-softmax
    rescale path (running max jumps at the last key tile)."""
    B, heads, H, W, E = 1, 1, 12, 12, 1
    N = H * W + E
    qkv = torch.randn(N, 192, generator=g(3)) * 0.3
    qkv[5, 0:64] = 6.0          # query 5
    qkv[N - 2, 64:128] = 6.0    # key N-2 aligned with it
    qkv = qkv.cuda()
    freqs = torch.zeros(2, 1, 32).cuda()  # cos == 1
    cos = ops.rope_cos_table(freqs, H, W)
    o = torch.empty(N, 64, device="cuda")
    lse = torch.empty(1, 1, N, device="cuda")
    ops.attn_fwd(qkv, cos, o, lse, B, N, E, heads)
    ref = _attn_ref(qkv.double().cpu(), freqs.double().cpu(), B, N, E, heads, H, W)
    torch.testing.assert_close(o.double().cpu(), ref, rtol=1e-4, atol=1e-5)


@pytest.mark.parametrize("dtype", [L.F32, L.BF16])
def test_small_kernels(dtype):
    gen = g(21)
    tdt = DT[dtype]
    # im2col
    x = torch.randn(2, 3, 16, 24, generator=gen).cuda()
    pat = torch.full((2 * 4 * 6, 64), float("nan"), device="cuda", dtype=tdt)
    ops.im2col_stem(x, pat)
    ref = torch.nn.functional.unfold(x.cpu(), 4, stride=4).transpose(1, 2).reshape(-1, 48)
    torch.testing.assert_close(pat[:, :48].float().cpu(), ref.to(tdt).float(), rtol=0, atol=0)
    assert pat[:, 48:].abs().sum().item() == 0
    # scale_cast with row map
    B, HW, E, C_ = 3, 4, 2, 32
    tok = torch.randn(B * (HW + E), C_, generator=gen).cuda()
    rs = torch.tensor([0.0, 2.0, 1.0]).cuda()
    out = torch.empty(B * HW, C_, device="cuda", dtype=tdt)
    ops.scale_cast(tok, out, B * HW, C_, in_map=(HW, E, E), rowscale=rs, rows_per_sample=HW)
    ref = (tok.view(B, HW + E, C_)[:, E:] * rs[:, None, None]).reshape(B * HW, C_).to(tdt)
    torch.testing.assert_close(out.float(), ref.float(), rtol=0, atol=0)
    # layerscale backward
    M, C_ = 150, 96
    gg = torch.randn(M, C_, generator=gen).cuda()
    z = torch.randn(M, C_, generator=gen).cuda().to(tdt)
    gam = torch.randn(C_, generator=gen).cuda()
    rs = torch.tensor([0.0, 1.25, 1.25]).cuda()
    dz = torch.empty(M, C_, device="cuda", dtype=tdt)
    dgam = torch.zeros(C_, device="cuda")
    ops.layerscale_bwd(gg, z, gam, rs, 50, dz, dgam, M, C_)
    s = rs.repeat_interleave(50)[:, None]
    torch.testing.assert_close(dz.float(), (s * gam * gg).to(tdt).float(), rtol=1e-2 if dtype else 1e-6, atol=1e-2 if dtype else 1e-6)
    torch.testing.assert_close(dgam, (s * gg * z.float()).sum(0), rtol=1e-4, atol=1e-4)
    # fill_rows / colsum_rows
    vec = torch.randn(C_, generator=gen).cuda()
    buf = torch.zeros(3 * 7, C_, device="cuda")
    ops.fill_rows(vec, buf, C_, (1, 6, 0), 3, C_)
    assert torch.equal(buf.view(3, 7, C_)[:, 0], vec.expand(3, C_)) and buf.view(3, 7, C_)[:, 1:].abs().sum().item() == 0
    src = torch.randn(3 * 7, C_, generator=gen).cuda()
    acc = torch.zeros(C_, device="cuda")
    ops.colsum_rows(src, C_, (1, 6, 2), acc, 3, C_)
    torch.testing.assert_close(acc, src.view(3, 7, C_)[:, 2].sum(0), rtol=1e-5, atol=1e-5)
    # aggregate
    a = torch.randn(5, 64, generator=gen).cuda()
    b = torch.randn(5, 64, generator=gen).cuda()
    w2 = torch.tensor([0.7, -0.3]).cuda()
    b1 = torch.tensor([0.2]).cuda()
    out = torch.empty(5, 64, device="cuda")
    ops.agg2_fwd(a, b, w2, b1, out, 5, 64)
    torch.testing.assert_close(out, 0.7 * a - 0.3 * b + 0.2, rtol=1e-6, atol=1e-6)
    dout = torch.randn(5, 64, generator=gen).cuda()
    da = torch.empty_like(a)
    dbb = torch.empty_like(b)
    dw2 = torch.zeros(2, device="cuda")
    db1 = torch.zeros(1, device="cuda")
    ops.agg2_bwd(dout, a, b, w2, da, dbb, dw2, db1, 5, 64)
    torch.testing.assert_close(da, 0.7 * dout)
    torch.testing.assert_close(dbb, -0.3 * dout)
    torch.testing.assert_close(dw2, torch.stack([(dout * a).sum(), (dout * b).sum()]), rtol=1e-5, atol=1e-5)
    torch.testing.assert_close(db1, dout.sum().reshape(1), rtol=1e-5, atol=1e-5)
    # pack_meta
    meta = torch.randn(4, 5, generator=gen).cuda()
    pm = torch.full((4, 16), float("nan"), device="cuda", dtype=tdt)
    ops.pack_meta(meta, 2, 3, pm)
    assert torch.equal(pm[:, :3].float(), meta[:, 2:5].to(tdt).float()) and pm[:, 3:].abs().sum().item() == 0


@pytest.mark.parametrize("C_,M", [(32, 200), (64, 130), (96, 777), (128, 100), (192, 333), (96, 6272), (96, 50001), (192, 40000), (128, 9000)])
def test_convmlp_fused_fwd_bwd(C_, M):
    """Fused pwconv1 -> GELU -> pwconv2 -> LayerScale -> DropPath -> +x against fp64 math on the same
    bf16 operands (tolerance = bf16 rounding of the on-chip hidden activation and of the outputs)."""
    gen = g(C_ * 3 + M)
    bf = torch.bfloat16
    rps = 50
    nb = (M + rps - 1) // rps
    ln = torch.randn(M, C_, generator=gen).cuda().to(bf)
    w1 = (torch.randn(4 * C_, C_, generator=gen) / C_**0.5).cuda().to(bf)
    b1 = (0.2 * torch.randn(4 * C_, generator=gen)).cuda()
    w2 = (torch.randn(C_, 4 * C_, generator=gen) / (4 * C_) ** 0.5).cuda().to(bf)
    b2 = (0.2 * torch.randn(C_, generator=gen)).cuda()
    gam = (0.5 + 0.3 * torch.randn(C_, generator=gen)).cuda()
    rs = (torch.rand(nb, generator=gen) > 0.3).float().cuda() * 1.25
    x = torch.randn(M, C_, generator=gen).cuda()
    out = torch.empty(M, C_, device="cuda")
    z = torch.empty(M, C_, device="cuda", dtype=bf)
    ops.convmlp_fwd(ln, w1, b1, w2, b2, gam, x, out, rowscale=rs, rows_per_sample=rps, z=z)
    # reference in fp64 with the same rounding points (hidden activation rounded to bf16 before pwconv2)
    lnd, w1d, w2d = ln.double(), w1.double(), w2.double()
    h = lnd @ w1d.T + b1.double()
    act = torch.nn.functional.gelu(h).to(bf).double()
    zr = act @ w2d.T + b2.double()
    s = rs.double().repeat_interleave(rps)[:M, None]
    ref = x.double() + s * gam.double() * zr
    torch.testing.assert_close(out.double(), ref, rtol=2e-3, atol=4e-3)  # one bf16 ulp flip of a hidden unit ~ 2e-3
    torch.testing.assert_close(z.double(), zr, rtol=1.6e-2, atol=1.6e-2)
    # backward
    gout = torch.randn(M, C_, generator=gen).cuda()
    actb = torch.empty(M, 4 * C_, device="cuda", dtype=bf)
    dh = torch.empty(M, 4 * C_, device="cuda", dtype=bf)
    dz = torch.empty(M, C_, device="cuda", dtype=bf)
    dln = torch.empty(M, C_, device="cuda", dtype=bf)
    dgam = torch.zeros(C_, device="cuda")
    ops.convmlp_bwd(gout, ln, z, w1, b1, w2.t().contiguous(), w1.t().contiguous(), gam, actb, dh, dz, dln, dgam, rowscale=rs, rows_per_sample=rps)
    dz_ref = (s * gam.double() * gout.double())
    torch.testing.assert_close(dz.double(), dz_ref, rtol=1e-2, atol=1e-2)
    torch.testing.assert_close(actb.double(), torch.nn.functional.gelu(h), rtol=1.6e-2, atol=1.6e-2)
    hg = h.clone().requires_grad_(True)
    torch.nn.functional.gelu(hg).sum().backward()
    dh_ref = (dz.double() @ w2d) * hg.grad
    torch.testing.assert_close(dh.double(), dh_ref, rtol=2e-2, atol=2e-2)
    torch.testing.assert_close(dln.double(), dh.double() @ w1d, rtol=2e-2, atol=3e-2)
    torch.testing.assert_close(dgam.double(), (s * gout.double() * z.double()).sum(0), rtol=1e-3, atol=1e-2)
    # the same call without materialising act / dH (what the plan launches) gives identical dz / dln / dgamma
    dz2, dln2, dgam2 = torch.empty_like(dz), torch.empty_like(dln), torch.zeros_like(dgam)
    ops.convmlp_bwd(gout, ln, z, w1, b1, w2.t().contiguous(), w1.t().contiguous(), gam, None, None, dz2, dln2, dgam2, rowscale=rs, rows_per_sample=rps)
    assert torch.equal(dz2, dz) and torch.equal(dln2, dln)
    torch.testing.assert_close(dgam2, dgam, rtol=2e-3, atol=2e-3)  # float atomics: summation order differs between launches


@pytest.mark.parametrize("margin", [5, 250])
def test_convmlp_resident_kernels_cover_every_row_for_any_grid(margin):
    """The resident-weight conv-MLP kernels draw 32- / 16-row tiles per WAVE from per-XCD counters: with lnx_set_cu_margin the grid is no
    multiple of 8 (uneven shares) or tiny (6 workgroups for thousands of tiles); forward and z-free backward must equal the default
    grid's results bit for bit (the forward) / to atomics order (column sums)."""
    C_, M = 96, 70001
    gen = g(99)
    bf = torch.bfloat16
    y = (1.5 * torch.randn(M, C_, generator=gen) + 0.3).cuda().to(bf)
    lw, lb = (1.0 + 0.2 * torch.randn(C_, generator=gen)).cuda(), (0.1 * torch.randn(C_, generator=gen)).cuda()
    w1, b1 = (torch.randn(4 * C_, C_, generator=gen) / C_**0.5).cuda().to(bf), (0.2 * torch.randn(4 * C_, generator=gen)).cuda()
    w2, b2 = (torch.randn(C_, 4 * C_, generator=gen) / (4 * C_) ** 0.5).cuda().to(bf), (0.2 * torch.randn(C_, generator=gen)).cuda()
    gam = (0.5 + 0.3 * torch.randn(C_, generator=gen)).cuda()
    x, gout = torch.randn(M, C_, generator=gen).cuda(), torch.randn(M, C_, generator=gen).cuda()
    w2t, w1t = w2.t().contiguous(), w1.t().contiguous()
    ws = torch.empty(256 * 2 * C_, device="cuda")

    def both():
        ln = torch.empty(M, C_, device="cuda", dtype=bf)
        mean, rstd, out = torch.empty(M, device="cuda"), torch.empty(M, device="cuda"), torch.empty(M, C_, device="cuda")
        ops.convmlp_fwd(None, w1, b1, w2, b2, gam, x, out, y=y, ln_w=lw, ln_b=lb, ln_eps=1e-6, ln_out=ln, mean=mean, rstd=rstd)
        act, dh = torch.empty(M, 4 * C_, device="cuda", dtype=bf), torch.empty(M, 4 * C_, device="cuda", dtype=bf)
        dz, dy = torch.empty(M, C_, device="cuda", dtype=bf), torch.full((M, C_), float("nan"), device="cuda", dtype=bf)
        dw, db = torch.zeros(C_, device="cuda"), torch.zeros(C_, device="cuda")
        ops.convmlp_bwd(gout, ln, None, w1, b1, w2t, w1t, gam, act, dh, dz, dy, None, y=y, ln_w=lw, mean=mean, rstd=rstd, d_ln_w=dw, d_ln_b=db, ws=ws, dz_plain=True)
        torch.cuda.synchronize()
        return out, ln, act, dh, dz, dy, dw, db

    ref = both()
    L.check(L.lib().lnx_set_cu_margin(margin), "lnx_set_cu_margin")
    try:
        for _ in range(2):  # twice: the second launch finds the counters the first one left
            got = both()
            for a, b_ in zip(ref[:6], got[:6]):
                assert torch.equal(a, b_)
            torch.testing.assert_close(got[6], ref[6], rtol=1e-4, atol=1e-3 * max(1.0, ref[6].abs().max().item()))
            torch.testing.assert_close(got[7], ref[7], rtol=1e-4, atol=1e-3 * max(1.0, ref[7].abs().max().item()))
    finally:
        L.check(L.lib().lnx_set_cu_margin(0), "lnx_set_cu_margin")


@pytest.mark.parametrize("C_,M", [(32, 200), (64, 130), (96, 777), (128, 100), (192, 333), (96, 50001), (192, 40000)])
def test_convmlp_fused_layernorm(C_, M):
    """The conv-MLP kernels with the block LayerNorm inside (blocks/convnext.py:77,84: `x = self.norm(x)` in front of pwconv1):
    forward = lnx_layernorm_fwd followed by the plain kernel; backward = the plain kernel followed by lnx_layernorm_bwd.  The two
    forms differ only in the summation order of the row statistics, i.e. by single bf16 ulps of the normalised rows."""
    gen = g(C_ * 7 + M)
    bf = torch.bfloat16
    rps = 50
    nb = (M + rps - 1) // rps
    y = (1.5 * torch.randn(M, C_, generator=gen) + 0.3).cuda().to(bf)
    lw = (1.0 + 0.2 * torch.randn(C_, generator=gen)).cuda()
    lb = (0.1 * torch.randn(C_, generator=gen)).cuda()
    w1 = (torch.randn(4 * C_, C_, generator=gen) / C_**0.5).cuda().to(bf)
    b1 = (0.2 * torch.randn(4 * C_, generator=gen)).cuda()
    w2 = (torch.randn(C_, 4 * C_, generator=gen) / (4 * C_) ** 0.5).cuda().to(bf)
    b2 = (0.2 * torch.randn(C_, generator=gen)).cuda()
    gam = (0.5 + 0.3 * torch.randn(C_, generator=gen)).cuda()
    rs = (torch.rand(nb, generator=gen) > 0.3).float().cuda() * 1.25
    x = torch.randn(M, C_, generator=gen).cuda()
    # two-pass form
    ln0 = torch.empty(M, C_, device="cuda", dtype=bf)
    mean0, rstd0 = torch.empty(M, device="cuda"), torch.empty(M, device="cuda")
    ops.layernorm_fwd(y, lw, lb, ln0, eps=1e-6, mean=mean0, rstd=rstd0)
    out0, z0 = torch.empty(M, C_, device="cuda"), torch.empty(M, C_, device="cuda", dtype=bf)
    ops.convmlp_fwd(ln0, w1, b1, w2, b2, gam, x, out0, rowscale=rs, rows_per_sample=rps, z=z0)
    # fused form
    ln1 = torch.full((M, C_), float("nan"), device="cuda", dtype=bf)
    mean1, rstd1 = torch.full((M,), float("nan"), device="cuda"), torch.full((M,), float("nan"), device="cuda")
    out1, z1 = torch.empty(M, C_, device="cuda"), torch.empty(M, C_, device="cuda", dtype=bf)
    ops.convmlp_fwd(None, w1, b1, w2, b2, gam, x, out1, rowscale=rs, rows_per_sample=rps, z=z1, y=y, ln_w=lw, ln_b=lb, ln_eps=1e-6, ln_out=ln1,
                    mean=mean1, rstd=rstd1)
    torch.testing.assert_close(mean1, mean0, rtol=1e-5, atol=1e-6)
    torch.testing.assert_close(rstd1, rstd0, rtol=1e-5, atol=1e-6)
    torch.testing.assert_close(ln1.float(), ln0.float(), rtol=8e-3, atol=1e-6)  # one bf16 ulp
    assert (ln1 != ln0).float().mean().item() < 1e-3
    yd = y.double()
    mu = yd.mean(1, keepdim=True)
    xh = (yd - mu) / torch.sqrt(yd.var(1, unbiased=False, keepdim=True) + 1e-6)
    torch.testing.assert_close(ln1.double(), xh * lw.double() + lb.double(), rtol=8e-3, atol=8e-3)
    torch.testing.assert_close(out1, out0, rtol=2e-3, atol=6e-3)
    # the same without the optional outputs (an inference plan's call)
    out2 = torch.empty_like(out1)
    ops.convmlp_fwd(None, w1, b1, w2, b2, gam, x, out2, rowscale=rs, rows_per_sample=rps, y=y, ln_w=lw, ln_b=lb, ln_eps=1e-6)
    assert torch.equal(out2, out1)

    # backward, both forms from the SAME saved tensors (ln1, z1, mean1, rstd1)
    gout = torch.randn(M, C_, generator=gen).cuda()
    w2t, w1t = w2.t().contiguous(), w1.t().contiguous()
    def bufs():
        return (torch.empty(M, 4 * C_, device="cuda", dtype=bf), torch.empty(M, 4 * C_, device="cuda", dtype=bf), torch.empty(M, C_, device="cuda", dtype=bf),
                torch.full((M, C_), float("nan"), device="cuda", dtype=bf), torch.zeros(C_, device="cuda"))
    act0, dh0, dz0, dln0, dg0 = bufs()
    ops.convmlp_bwd(gout, ln1, z1, w1, b1, w2t, w1t, gam, act0, dh0, dz0, dln0, dg0, rowscale=rs, rows_per_sample=rps)
    dy0 = torch.empty(M, C_, device="cuda", dtype=bf)
    dw0, db0 = torch.zeros(C_, device="cuda"), torch.zeros(C_, device="cuda")
    ops.layernorm_bwd(dln0, y, lw, mean1, rstd1, dy0, dw=dw0, db=db0)
    act1, dh1, dz1, dy1, dg1 = bufs()
    dw1, db1 = torch.full((C_,), 3.0, device="cuda"), torch.full((C_,), -2.0, device="cuda")  # accumulated into
    ws = torch.empty(max(256, (M + 127) // 128) * 2 * C_, device="cuda")
    ops.convmlp_bwd(gout, ln1, z1, w1, b1, w2t, w1t, gam, act1, dh1, dz1, dy1, dg1, rowscale=rs, rows_per_sample=rps, y=y, ln_w=lw, mean=mean1,
                    rstd=rstd1, d_ln_w=dw1, d_ln_b=db1, ws=ws)
    assert torch.equal(act1, act0) and torch.equal(dh1, dh0) and torch.equal(dz1, dz0)
    torch.testing.assert_close(dy1.float(), dy0.float(), rtol=1.6e-2, atol=2e-3 * dy0.float().abs().max().item())
    scale = max(1.0, dw0.abs().max().item())
    torch.testing.assert_close(dw1 - 3.0, dw0, rtol=2e-3, atol=2e-3 * scale)
    torch.testing.assert_close(db1 + 2.0, db0, rtol=2e-3, atol=2e-3 * max(1.0, db0.abs().max().item()))
    # against fp64 math on the stored dln
    dl = dln0.double()
    gvv = dl * lw.double()
    dy_ref = (gvv - gvv.mean(1, keepdim=True) - xh * (gvv * xh).mean(1, keepdim=True)) * rstd1.double()[:, None]
    torch.testing.assert_close(dy1.double(), dy_ref, rtol=1.6e-2, atol=2e-3 * dy_ref.abs().max().item())
    torch.testing.assert_close((dw1 - 3.0).double(), (dl * xh).sum(0), rtol=3e-3, atol=3e-3 * scale)
    # too little scratch for the column sums is an error, not a silent overrun
    with pytest.raises(L.LnxError):
        ops.convmlp_bwd(gout, ln1, z1, w1, b1, w2t, w1t, gam, act1, dh1, dz1, dy1, dg1, rowscale=rs, rows_per_sample=rps, y=y, ln_w=lw, mean=mean1,
                        rstd=rstd1, d_ln_w=dw1, d_ln_b=db1, ws=ws[:C_])

    # ---- round 4: the forms a training plan launches -- z neither written by the forward nor read by the backward ----
    ln3 = torch.full((M, C_), float("nan"), device="cuda", dtype=bf)
    mean3, rstd3 = torch.empty(M, device="cuda"), torch.empty(M, device="cuda")
    out3 = torch.empty_like(out1)
    ops.convmlp_fwd(None, w1, b1, w2, b2, gam, x, out3, rowscale=rs, rows_per_sample=rps, y=y, ln_w=lw, ln_b=lb, ln_eps=1e-6, ln_out=ln3, mean=mean3, rstd=rstd3)
    assert torch.equal(out3, out1) and torch.equal(ln3, ln1) and torch.equal(mean3, mean1) and torch.equal(rstd3, rstd1)
    act3, dh3, dz3, dy3, _ = bufs()
    dw3, db3 = torch.full((C_,), 3.0, device="cuda"), torch.full((C_,), -2.0, device="cuda")
    ops.convmlp_bwd(gout, ln1, None, w1, b1, w2t, w1t, gam, act3, dh3, dz3, dy3, None, rowscale=rs, rows_per_sample=rps, y=y, ln_w=lw, mean=mean1,
                    rstd=rstd1, d_ln_w=dw3, d_ln_b=db3, ws=ws)
    assert torch.equal(act3, act1) and torch.equal(dh3, dh1) and torch.equal(dz3, dz1) and torch.equal(dy3, dy1)
    torch.testing.assert_close(dw3, dw1, rtol=1e-5, atol=1e-4 * scale)  # (column sums through atomics: order differs from launch to launch)
    # ... with dz_plain the dz that reaches memory is rs * g without the LayerScale factor; nothing else changes
    act4, dh4, dz4, dy4, _ = bufs()
    dw4, db4 = torch.zeros(C_, device="cuda"), torch.zeros(C_, device="cuda")
    ops.convmlp_bwd(gout, ln1, None, w1, b1, w2t, w1t, gam, act4, dh4, dz4, dy4, None, rowscale=rs, rows_per_sample=rps, y=y, ln_w=lw, mean=mean1,
                    rstd=rstd1, d_ln_w=dw4, d_ln_b=db4, ws=ws, dz_plain=True)
    assert torch.equal(act4, act1) and torch.equal(dh4, dh1) and torch.equal(dy4, dy1)
    rsr = rs.double().repeat_interleave(rps)[:M, None]
    torch.testing.assert_close(dz4.double(), rsr * gout.double(), rtol=8e-3, atol=1e-6)
    # ... and the LayerScale gradient comes out of the pwconv2 weight gradient (lnx_layerscale_apply_wgrad): S = (rs g)^T act and
    # T = colsum(rs g) as the TN GEMM leaves them in zeroed scratch; dW2 += gamma S, db2 += gamma T, dgamma += rowdot(W2, S) + b2 T.
    # A zero gamma is an ordinary value (no division anywhere).
    gam0 = gam.clone()
    gam0[3] = 0.0
    S = dz4.float().t() @ act1.float()
    T = dz4.float().sum(0)
    dW2, db2g, dgam = torch.full((C_, 4 * C_), 0.5, device="cuda"), torch.full((C_,), -0.5, device="cuda"), torch.full((C_,), 0.75, device="cuda")
    ops.layerscale_apply_wgrad(S, T, w2.float(), b2, gam0, dW2, db2g, dgam)
    torch.cuda.synchronize()
    torch.testing.assert_close(dW2, 0.5 + gam0[:, None] * S, rtol=1e-5, atol=1e-5 * max(1.0, S.abs().max().item()))
    torch.testing.assert_close(db2g, -0.5 + gam0 * T, rtol=1e-5, atol=1e-5 * max(1.0, T.abs().max().item()))
    # reference 1: fp64 on the operands the kernels saw; reference 2: the kernel that reads the saved bf16 z (dg1)
    zz = act1.double() @ w2.double().t() + b2.double()
    dg_ref = (rsr * gout.double() * zz).sum(0)
    gscale = max(1.0, dg_ref.abs().max().item())
    torch.testing.assert_close((dgam - 0.75).double(), dg_ref, rtol=1e-2, atol=1e-2 * gscale)   # dz is bf16(rs g): 2^-9 per term
    torch.testing.assert_close(dgam - 0.75, dg1, rtol=2e-2, atol=2e-2 * gscale)
    assert torch.isfinite(dgam).all()


@pytest.mark.parametrize("M,N,K", [(256, 128, 256), (1000, 384, 512), (4096, 1152, 384 * 2), (513, 208, 1024)])
@pytest.mark.parametrize("xd", [L.BF16, L.F32])
def test_fp8_quantize_and_gemm(M, N, K, xd):
    """fp8 path (BASELINE config 5; the reference has no fp8 code, so the expected values are the definition itself):
    quantisation = torch's own e4m3fn rounding of x * 448 / amax, and the GEMM = the fp64 product of the DEQUANTISED
    operands, i.e. the only error left is fp32 accumulation order."""
    gen = g(M + N + K)
    x = (torch.randn(M, K, generator=gen) * 3).cuda().to(DT[xd])
    w = (torch.randn(N, K, generator=gen) / K ** 0.5).cuda().to(torch.bfloat16)
    x8, sx = ops.quantize_fp8(x)
    w8, sw = ops.quantize_fp8(w)
    amax = x.float().abs().max()
    assert torch.allclose(sx, (amax / 448).reshape(1), rtol=1e-6)
    # 448 / amax as the correctly rounded fp32 quotient (torch's own tensor division on the GPU is a reciprocal-multiply
    # and can be one ulp off, which flips every bf16 input that lands just past a rounding tie); then one fp32 multiply
    # and torch's round-to-nearest-even e4m3 conversion: byte-exact
    inv = torch.tensor(448.0 / amax.double().item(), dtype=torch.float32).item()
    ref8 = (x.float() * inv).clamp(-448, 448).to(torch.float8_e4m3fn)
    assert torch.equal(x8.view(torch.uint8), ref8.view(torch.uint8))
    xd_ = x8.float().double().cpu() * sx.item()
    wd_ = w8.float().double().cpu() * sw.item()
    bias = torch.randn(N, generator=gen).cuda()
    # plain, bias, bias + GELU (+ pre-activation copy), fp32 output with residual and DropPath row scale
    out = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
    ops.gemm_nt_fp8(x8, sx, w8, sw, out)
    ref = xd_ @ wd_.t()
    tol = 8e-3 * max(1.0, ref.abs().max().item())
    torch.testing.assert_close(out.double().cpu(), ref, rtol=8e-3, atol=tol)
    if N % 16 == 0:
        ops.gemm_nt_fp8(x8, sx, w8, sw, out, bias=bias)
        torch.testing.assert_close(out.double().cpu(), ref + bias.double().cpu(), rtol=8e-3, atol=tol)
        pre = torch.empty_like(out)
        ops.gemm_nt_fp8(x8, sx, w8, sw, out, bias=bias, act=L.ACT_GELU, c2=pre)
        h = ref + bias.double().cpu()
        torch.testing.assert_close(pre.double().cpu(), h, rtol=8e-3, atol=tol)
        torch.testing.assert_close(out.double().cpu(), torch.nn.functional.gelu(h), rtol=8e-3, atol=tol)
        res = torch.randn(M, N, generator=gen).cuda()
        rs = (torch.rand(4, generator=gen) + 0.5).cuda()
        rps = (M + 3) // 4
        o32 = torch.empty(M, N, device="cuda")
        ops.gemm_nt_fp8(x8, sx, w8, sw, o32, bias=bias, res=res, rowscale=rs, rows_per_sample=rps)
        rowf = rs.double().cpu()[torch.arange(M) // rps].reshape(M, 1)
        torch.testing.assert_close(o32.double().cpu(), res.double().cpu() + rowf * h, rtol=1e-4, atol=1e-4 * max(1.0, ref.abs().max().item()))


def _mx_reference(x):
    """The definition lnx_quantize_mxfp8 implements, restated with torch integer ops: per 32-element block the smallest
    power of two 2^e with amax * 2^-e <= 448, elements = e4m3fn(x * 2^-e) (torch's round-to-nearest-even conversion)."""
    M, K = x.shape
    xb = x.float().view(M, K // 32, 32)
    bits = xb.abs().amax(-1).contiguous().view(torch.int32)
    e = ((bits >> 23) & 0xFF) - 8 + ((bits & 0x7FFFFF) > 0x600000).int()
    e = e.clamp(0, 254)
    inv = ((254 - e) << 23).view(torch.float32)
    q = (xb * inv.unsqueeze(-1)).clamp(-448, 448).to(torch.float8_e4m3fn).view(M, K)
    return q, e.to(torch.uint8)  # e: [M, K/32]


@pytest.mark.parametrize("M,N,K", [(256, 128, 256), (1000, 384, 512), (4096, 1152, 384 * 2), (513, 208, 1024), (33000, 512, 512), (16500, 1024, 384)])
@pytest.mark.parametrize("xd", [L.BF16, L.F32])
def test_mxfp8_quantize_and_gemm(M, N, K, xd):
    """MXFP8 path (block-scaled e4m3, scales applied inside v_mfma_scale_f32_16x16x128_f8f6f4; the last two shapes -- N a
    multiple of 256, at least 128 tiles -- take the 256x256-tile kernel on v_mfma_scale_f32_32x32x64_f8f6f4, round 3).  Quantisation is checked
    byte for byte against the torch restatement above on data whose blocks span ~2^±20 (plus an all-zero block, a tiny
    block and a block at the top of the fp32/bf16 range); the GEMM against the fp64 product of the DEQUANTISED operands,
    so the only error left is fp32 accumulation order -- which also proves the scale layout and the instruction's k order."""
    gen = g(M * 7 + N + K)
    x = torch.randn(M, K, generator=gen) * torch.exp2(torch.randint(-20, 21, (M, K // 32, 1), generator=gen).float()).expand(M, K // 32, 32).reshape(M, K)
    x[3, 32:64] = 0.0
    x[5, 0:32] = 1e-30
    x[7, 64:96] = 3e38
    x = x.cuda().to(DT[xd])
    w = (torch.randn(N, K, generator=gen) / K ** 0.5 * torch.exp2(torch.randint(-3, 4, (N, K // 32, 1), generator=gen).float()).expand(N, K // 32, 32).reshape(N, K))
    w = w.cuda().to(torch.bfloat16)
    x8, sx = ops.quantize_mxfp8(x)
    w8, sw = ops.quantize_mxfp8(w)
    q, e = _mx_reference(x)
    assert torch.equal(x8.view(torch.uint8), q.view(torch.uint8))
    assert torch.equal(sx.permute(1, 0, 2).reshape(M, K // 32), e)
    # what the format promises: 3 mantissa bits relative to the element, 2^-9 of the block's scale at the bottom
    xd_ = ops.dequantize_mxfp8(x8, sx)
    blk_amax = x.float().abs().view(M, K // 32, 32).amax(-1, keepdim=True).expand(M, K // 32, 32).reshape(M, K)
    assert ((xd_ - x.float()).abs() <= x.float().abs() / 16 + blk_amax / 2 ** 16).all()
    # GEMM on well-scaled activations (the wide-range x above would overflow a bf16 output)
    a = (torch.randn(M, K, generator=gen) * torch.exp2(torch.randint(-4, 5, (M, K // 32, 1), generator=gen).float()).expand(M, K // 32, 32).reshape(M, K)).cuda().to(DT[xd])
    a8, sa = ops.quantize_mxfp8(a)
    ad = ops.dequantize_mxfp8(a8, sa).double().cpu()
    wd = ops.dequantize_mxfp8(w8, sw).double().cpu()
    ref = ad @ wd.t()
    tol = 8e-3 * max(1.0, ref.abs().max().item())
    out = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
    ops.gemm_nt_mxfp8(a8, sa, w8, sw, out)
    torch.testing.assert_close(out.double().cpu(), ref, rtol=8e-3, atol=tol)
    bias = torch.randn(N, generator=gen).cuda()
    ops.gemm_nt_mxfp8(a8, sa, w8, sw, out, bias=bias)
    torch.testing.assert_close(out.double().cpu(), ref + bias.double().cpu(), rtol=8e-3, atol=tol)
    pre = torch.empty_like(out)
    ops.gemm_nt_mxfp8(a8, sa, w8, sw, out, bias=bias, act=L.ACT_GELU, c2=pre)
    h = ref + bias.double().cpu()
    torch.testing.assert_close(pre.double().cpu(), h, rtol=8e-3, atol=tol)
    torch.testing.assert_close(out.double().cpu(), torch.nn.functional.gelu(h), rtol=8e-3, atol=tol)
    if N % 128 == 0:
        # the fc1 -> fc2 hand-over: the epilogue also writes the MXFP8 copy of its bf16 output, bit-identical to quantising it
        out2, pre2 = torch.empty_like(out), torch.empty_like(out)
        c8 = torch.zeros(M, N, device="cuda", dtype=torch.uint8)
        c8s = torch.zeros(N // 128, M, 4, device="cuda", dtype=torch.uint8)
        ops.gemm_nt_mxfp8(a8, sa, w8, sw, out2, bias=bias, act=L.ACT_GELU, c2=pre2, c8=c8, c8_scales=c8s)
        assert torch.equal(out2, out) and torch.equal(pre2, pre)
        r8, rsc = ops.quantize_mxfp8(out)
        assert torch.equal(c8s, rsc)
        assert torch.equal(c8, r8.view(torch.uint8))
    res = torch.randn(M, N, generator=gen).cuda()
    rs = (torch.rand(4, generator=gen) + 0.5).cuda()
    rps = (M + 3) // 4
    o32 = torch.empty(M, N, device="cuda")
    ops.gemm_nt_mxfp8(a8, sa, w8, sw, o32, bias=bias, res=res, rowscale=rs, rows_per_sample=rps)
    rowf = rs.double().cpu()[torch.arange(M) // rps].reshape(M, 1)
    torch.testing.assert_close(o32.double().cpu(), res.double().cpu() + rowf * h, rtol=1e-4, atol=1e-4 * max(1.0, ref.abs().max().item()))
    # quantisation error of the whole product against the unquantised operands: the number a user cares about
    full = a.double().cpu() @ w.double().cpu().t()
    rel = (ref - full).norm() / full.norm()
    print(f"[mxfp8 M={M} N={N} K={K}] relative error of the MXFP8 product vs the unquantised one: {rel:.3e}")
    assert rel < 0.05


@pytest.mark.parametrize("M,N,K", [(128 * 199, 1024, 1024), (128 * 199, 3072, 1024), (128 * 199, 4096, 1024), (128 * 199, 1024, 4096)])
def test_mxfp8_256x256_kernel_at_xl_rows(M, N, K):
    """Round 5: gemm_nt_mx8_kernel (the 256x256-tile MXFP8 kernel) at the shapes `bench.py --arch xl --batch 128 --dtype fp8` gives it
    (M = 25 472; qkv, fc1, fc2 and -- with the opt-in MXFP8 data gradients -- proj), every epilogue it carries, against the fp64 product of
    the DEQUANTISED operands (on the GPU); the dispatcher's own record says which kernel ran."""
    gen = torch.Generator(device="cuda").manual_seed(M + N + K)
    a = (torch.randn(M, K, device="cuda", generator=gen) * torch.exp2(torch.randint(-3, 4, (M, K // 32, 1), device="cuda", generator=gen).float()).expand(M, K // 32, 32).reshape(M, K)).bfloat16()
    w = (torch.randn(N, K, device="cuda", generator=gen) / K ** 0.5).bfloat16()
    a8, sa = ops.quantize_mxfp8(a)
    w8, sw = ops.quantize_mxfp8(w)
    ref = ops.dequantize_mxfp8(a8, sa).double() @ ops.dequantize_mxfp8(w8, sw).double().t()
    tol = 8e-3 * max(1.0, ref.abs().max().item())
    bias = torch.randn(N, device="cuda", generator=gen)
    h = ref + bias.double()
    out = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
    ops.gemm_nt_mxfp8(a8, sa, w8, sw, out)
    assert L.lib().lnx_last_nt_kernel() == L.NT_KERNEL_MX8
    torch.testing.assert_close(out.double(), ref, rtol=8e-3, atol=tol)
    ops.gemm_nt_mxfp8(a8, sa, w8, sw, out, bias=bias)
    torch.testing.assert_close(out.double(), h, rtol=8e-3, atol=tol)
    pre = torch.empty_like(out)
    c8 = torch.zeros(M, N, device="cuda", dtype=torch.uint8)
    c8s = torch.zeros(N // 128, M, 4, device="cuda", dtype=torch.uint8)
    ops.gemm_nt_mxfp8(a8, sa, w8, sw, out, bias=bias, act=L.ACT_GELU, c2=pre, c8=c8, c8_scales=c8s)   # fc1: GELU + pre-activation + MXFP8 copy
    assert L.lib().lnx_last_nt_kernel() == L.NT_KERNEL_MX8
    torch.testing.assert_close(pre.double(), h, rtol=8e-3, atol=tol)
    torch.testing.assert_close(out.double(), torch.nn.functional.gelu(h), rtol=8e-3, atol=tol)
    r8, rsc = ops.quantize_mxfp8(out)
    assert torch.equal(c8s, rsc) and torch.equal(c8, r8.view(torch.uint8))
    aux = torch.randn(M, N, device="cuda", generator=gen).bfloat16()
    ops.gemm_nt_mxfp8(a8, sa, w8, sw, out, act=L.ACT_GELU_BWD, aux=aux)                                   # fc2 data gradient: x GELU'(aux)
    assert L.lib().lnx_last_nt_kernel() == L.NT_KERNEL_MX8
    xa = aux.double().requires_grad_(True)
    torch.nn.functional.gelu(xa).sum().backward()
    torch.testing.assert_close(out.double(), ref * xa.grad, rtol=1e-2, atol=1.25 * tol)
    res = torch.randn(M, N, device="cuda", generator=gen)
    rs = ((torch.rand(128, device="cuda", generator=gen) > 0.2).float() / 0.8)
    o32 = torch.empty(M, N, device="cuda")
    ops.gemm_nt_mxfp8(a8, sa, w8, sw, o32, bias=bias, res=res, rowscale=rs, rows_per_sample=199)          # proj / fc2: fp32 residual + DropPath scale
    assert L.lib().lnx_last_nt_kernel() == L.NT_KERNEL_MX8
    torch.testing.assert_close(o32.double(), res.double() + rs.double().repeat_interleave(199)[:, None] * h, rtol=1e-4, atol=1e-4 * max(1.0, ref.abs().max().item()))


@pytest.mark.parametrize("M,C", [(1000, 384), (777, 768), (300, 1024), (513, 2048), (64, 128), (200, 256), (130, 1536), (99, 1280)])
def test_layernorm_fused_mxfp8_output(M, C):
    """LayerNorm forward with the MXFP8 second output (the producer side of the model's fp8 mode): the bf16 output is what
    it is without the second output, and the fp8 bytes / block scales equal lnx_quantize_mxfp8 of that bf16 output exactly."""
    gen = g(M + C)
    x = (torch.randn(M, C, generator=gen) * torch.exp2(torch.randint(-3, 4, (M, 1), generator=gen).float()) + 0.3).cuda()
    w = (torch.rand(C, generator=gen) + 0.5).cuda()
    b = (torch.randn(C, generator=gen) * 0.1).cuda()
    y0 = torch.empty(M, C, device="cuda", dtype=torch.bfloat16)
    ops.layernorm_fwd(x, w, b, y0, 1e-5)
    y1 = torch.empty_like(y0)
    y8 = torch.zeros(M, C, device="cuda", dtype=torch.uint8)
    sc = torch.zeros(C // 128, M, 4, device="cuda", dtype=torch.uint8)
    mean = torch.empty(M, device="cuda")
    rstd = torch.empty(M, device="cuda")
    ops.layernorm_fwd(x, w, b, y1, 1e-5, mean=mean, rstd=rstd, y8=y8, y8_scales=sc)
    assert torch.equal(y0, y1)
    r8, rs = ops.quantize_mxfp8(y0)
    assert torch.equal(sc, rs)
    assert torch.equal(y8, r8.view(torch.uint8))
    torch.testing.assert_close(mean, x.mean(-1), rtol=1e-5, atol=1e-5)


@pytest.mark.parametrize("M,C,rps", [(1000, 384, 50), (777, 768, 7), (300, 1024, 100), (513, 2048, 53), (64, 128, 64), (130, 1536, 13), (2001, 1280, 200)])
def test_layernorm_bwd_second_output_and_its_mxfp8_copy(M, C, rps):
    """LayerNorm backward with the second output (dx2 = DropPath scale of the row's sample x dx in bf16: the dY the next branch's GEMMs read)
    and, round 4, its MXFP8 copy for fp8 plans' data-gradient products: dx and the column sums are what they are without the second
    output, dx2 equals the scaled dx rounded to bf16, and the fp8 bytes / block scales equal lnx_quantize_mxfp8 of that bf16 tensor exactly
    (every LayerNorm width of the RoPE stages: G = 32 / 64 lanes per row with 3, 4, 6 or 8 slots, a partly filled last slot at C = 1280)."""
    gen = g(M * 3 + C)
    x = (torch.randn(M, C, generator=gen) * 2 + 0.5).cuda()
    dy = (torch.randn(M, C, generator=gen) * torch.exp2(torch.randint(-6, 3, (M, 1), generator=gen).float())).cuda().bfloat16()
    w = (torch.rand(C, generator=gen) + 0.5).cuda()
    gin = torch.randn(M, C, generator=gen).cuda()
    rs = ((torch.rand(-(-M // rps), generator=gen) > 0.2).float() / 0.8).cuda()
    mean = x.mean(-1)
    rstd = (x.var(-1, unbiased=False) + 1e-5).rsqrt()
    ws = torch.empty(2048 * 2 * C, device="cuda")
    dx0, dw0, db0 = torch.empty(M, C, device="cuda"), torch.zeros(C, device="cuda"), torch.zeros(C, device="cuda")
    ops.layernorm_bwd(dy, x, w, mean, rstd, dx0, gin=gin, dw=dw0, db=db0, ws=ws)
    dx1, dw1, db1 = torch.empty(M, C, device="cuda"), torch.zeros(C, device="cuda"), torch.zeros(C, device="cuda")
    d2 = torch.empty(M, C, device="cuda", dtype=torch.bfloat16)
    d8 = torch.zeros(M, C, device="cuda", dtype=torch.uint8)
    sc = torch.zeros(C // 128, M, 4, device="cuda", dtype=torch.uint8)
    ops.layernorm_bwd(dy, x, w, mean, rstd, dx1, gin=gin, dw=dw1, db=db1, ws=ws, dx2=d2, dx2_rowscale=rs, dx2_rows_per_sample=rps, dx2_8=d8, dx2_8_scales=sc)
    assert torch.equal(dx0, dx1)
    torch.testing.assert_close(dw1, dw0, rtol=1e-5, atol=1e-4)
    torch.testing.assert_close(db1, db0, rtol=1e-5, atol=1e-4)
    want = (dx1 * rs.repeat_interleave(rps)[:M, None]).bfloat16()
    assert torch.equal(d2, want)
    r8, rsc = ops.quantize_mxfp8(d2)
    assert torch.equal(sc, rsc)
    assert torch.equal(d8, r8.view(torch.uint8))
    # the plain second output without the copy is the same tensor
    d2b = torch.empty_like(d2)
    ops.layernorm_bwd(dy, x, w, mean, rstd, dx1, gin=gin, ws=ws, dx2=d2b, dx2_rowscale=rs, dx2_rows_per_sample=rps)
    assert torch.equal(d2b, d2)
    # C % 128 != 0 is refused with the copy, accepted without
    if C == 384:
        xs, dys = x[:, :96].contiguous(), dy[:, :96].contiguous()
        with pytest.raises(L.LnxError, match="MXFP8 copy"):
            ops.layernorm_bwd(dys, xs, w[:96].contiguous(), mean, rstd, torch.empty(M, 96, device="cuda"), dx2=torch.empty(M, 96, device="cuda", dtype=torch.bfloat16),
                              dx2_8=d8, dx2_8_scales=sc)


@pytest.mark.parametrize("B,Cin,H,W,Cout", [(3, 3, 56, 72, 96), (2, 3, 32, 32, 192), (1, 4, 16, 20, 128), (2, 1, 64, 64, 256), (40, 3, 224, 224, 96)])
def test_fused_stem_matches_conv_and_layernorm(B, Cin, H, W, Cout):
    """lnx_stem_fwd (round 3) against torch: conv2d 4x4/4 on bf16-rounded operands, output rounded to bf16, channels-first
    LayerNorm in fp32; its patch matrix against lnx_im2col_stem's bit for bit.  The last case is large enough for the
    8-tiles-per-wave launch."""
    g = torch.Generator().manual_seed(B + H + Cout)
    x = torch.randn(B, Cin, H, W, generator=g).cuda()
    w4 = (torch.randn(Cout, Cin, 4, 4, generator=g) / (Cin * 16) ** 0.5).cuda()
    bias, lw, lb = (torch.randn(Cout, generator=g).cuda() for _ in range(3))
    wq = torch.full((Cout, 64), float("nan"), device="cuda", dtype=torch.bfloat16)  # the padding columns must not matter
    wq[:, :Cin * 16] = w4.reshape(Cout, -1).bfloat16()
    M = B * (H // 4) * (W // 4)
    y = torch.empty(M, Cout, device="cuda")
    pre = torch.empty(M, Cout, device="cuda", dtype=torch.bfloat16)
    patches = torch.empty(M, 64, device="cuda", dtype=torch.bfloat16)
    mean, rstd = torch.empty(M, device="cuda"), torch.empty(M, device="cuda")
    ops.stem_fwd(x, wq, bias, lw, lb, y, patches=patches, pre=pre, mean=mean, rstd=rstd)
    ref_p = torch.zeros_like(patches)
    ops.im2col_stem(x, ref_p)
    assert torch.equal(patches[:, :Cin * 16], ref_p[:, :Cin * 16]) and not patches.float().abs()[:, Cin * 16:].any()
    conv = torch.nn.functional.conv2d(x.bfloat16().double(), w4.bfloat16().double(), stride=4) + bias.double()[None, :, None, None]
    conv = conv.permute(0, 2, 3, 1).reshape(M, Cout)
    torch.testing.assert_close(pre.double(), conv, rtol=8e-3, atol=8e-3)
    pf = pre.double()
    mu, var = pf.mean(1, keepdim=True), pf.var(1, unbiased=False, keepdim=True)
    torch.testing.assert_close(y.double(), (pf - mu) / (var + 1e-6).sqrt() * lw.double() + lb.double(), rtol=2e-5, atol=2e-5)
    torch.testing.assert_close(mean.double(), mu[:, 0], rtol=1e-5, atol=1e-6)
    torch.testing.assert_close(rstd.double(), 1 / (var[:, 0] + 1e-6).sqrt(), rtol=2e-5, atol=1e-6)
    y2 = torch.empty_like(y)
    ops.stem_fwd(x, wq, bias, lw, lb, y2)  # inference form: nothing but y
    assert torch.equal(y, y2)


# ----------------------------------------------------------------------------------------------------
# Round 5: metadata-head chains in one launch per direction (metahead.hip) against torch autograd on the reference's chain
# (mFormerV1.py:282-311: Linear -> ReLU -> LayerNorm -> ResNormLayer; res_norm_layer.py:23-30)
# ----------------------------------------------------------------------------------------------------
def _meta_head_params(C, dim, gen, dev="cuda"):
    r = lambda *s, sc=1.0: (torch.randn(*s, generator=gen) * sc).to(dev)  # noqa: E731
    return {"w0": r(C, dim, sc=0.7), "b0": r(C, sc=0.3), "ln0_w": 1 + r(C, sc=0.2), "ln0_b": r(C, sc=0.2),
            "w1": r(C, C, sc=C ** -0.5), "b1": r(C, sc=0.3), "ln1_w": 1 + r(C, sc=0.2), "ln1_b": r(C, sc=0.2),
            "w2": r(C, C, sc=C ** -0.5), "b2": r(C, sc=0.3), "ln2_w": 1 + r(C, sc=0.2), "ln2_b": r(C, sc=0.2)}


def _meta_head_reference(meta, h):
    F = torch.nn.functional
    C = h["w1"].shape[0]
    p = {k: v.double().requires_grad_(True) for k, v in h.items() if torch.is_tensor(v)}
    t = meta.double()[:, h["off"]:h["off"] + h["dim"]]
    x = F.layer_norm(F.relu(t @ p["w0"].t() + p["b0"]), (C,), p["ln0_w"], p["ln0_b"], 1e-5)
    n1 = F.layer_norm(F.relu(x @ p["w1"].t() + p["b1"]), (C,), p["ln1_w"], p["ln1_b"], 1e-5)
    y = x + F.layer_norm(F.relu(n1 @ p["w2"].t() + p["b2"]), (C,), p["ln2_w"], p["ln2_b"], 1e-5)
    return y, p


@pytest.mark.parametrize("B,cfg", [(256, [(384, 2, 0), (384, 3, 2), (768, 2, 0), (768, 3, 2)]), (24, [(768, 3, 2)]), (5, [(128, 10, 5), (256, 2, 0)]),
                                   (130, [(1024, 3, 2), (1024, 2, 0), (512, 2, 0)]), (33, [(128, 1, 4), (128, 10, 5), (128, 2, 0), (128, 3, 2), (128, 16, 0), (768, 16, 0), (256, 2, 0)])])
def test_meta_head_chain_one_launch_matches_autograd(B, cfg):
    """lnx_meta_heads_fwd / lnx_meta_heads_bwd: several heads of different widths in one call (sm's four heads at the benchmark's batch; every
    width the chain carries, C = 128 .. 1024; five heads of one width = two launches; partial last row group; dim up to 16) against fp64
    autograd of the reference chain.  fp32 arithmetic: tokens to 2e-5, every gradient to 1e-4 of its scale; gradients ACCUMULATE onto what the
    buffers held; rows of the token matrix that belong to other tokens are untouched; two runs are bit-identical (no atomics)."""
    gen = g(B * 131 + len(cfg))
    meta = torch.randn(B, 16, generator=gen).cuda()
    N = 7
    heads, toks = [], []
    for i, (C, dim, off) in enumerate(cfg):
        h = _meta_head_params(C, dim, gen)
        h.update(dim=dim, off=off, slot=1 + i % (N - 1))
        heads.append(h)
        toks.append(torch.full((B, N, C), 7.0, device="cuda"))
    saved = ops.meta_heads_fwd(meta, heads, toks)
    torch.cuda.synchronize()
    gouts, refs = [], []
    for h, tok in zip(heads, toks):
        y, p = _meta_head_reference(meta, h)
        got = tok[:, h["slot"]]
        torch.testing.assert_close(got.double(), y.detach(), rtol=2e-5, atol=2e-5 * max(1.0, y.abs().max().item()))
        other = torch.ones(N, dtype=torch.bool)
        other[h["slot"]] = False
        assert (tok[:, other] == 7.0).all()
        gout = torch.randn(B, N, h["w1"].shape[0], generator=gen).cuda()
        (y * gout[:, h["slot"]].double()).sum().backward()
        gouts.append(gout)
        refs.append(p)
    keys = ("w0", "b0", "ln0_w", "ln0_b", "w1", "b1", "ln1_w", "ln1_b", "w2", "b2", "ln2_w", "ln2_b")
    runs = []
    for _ in range(2):
        grads = [{k: torch.full_like(h[k], 0.5) for k in keys} for h in heads]
        ops.meta_heads_bwd(gouts, heads, saved, grads)
        runs.append(grads)
    for h, gr, gr2, p in zip(heads, runs[0], runs[1], refs):
        for k in keys:
            ref = p[k].grad
            scale = max(1e-3, ref.abs().max().item())
            torch.testing.assert_close((gr[k] - 0.5).double(), ref, rtol=1e-4, atol=1e-4 * scale, msg=lambda m, k=k, C=h["w1"].shape[0]: f"{k} (C={C}): {m}")
            assert torch.equal(gr[k], gr2[k]), k


def test_meta_head_chain_refuses_widths_it_does_not_carry():
    """C / 128 must be in {1, 2, 3, 4, 6, 8}: the plan keeps the launch-by-launch chain for wider stages (lg / xl stage 4) and asks first."""
    lib = L.lib()
    assert [lib.lnx_meta_heads_supported(c) for c in (128, 256, 384, 512, 768, 1024)] == [1] * 6
    assert [lib.lnx_meta_heads_supported(c) for c in (64, 192, 640, 896, 1536, 2048)] == [0] * 6
    gen = g(1)
    meta = torch.randn(4, 5, generator=gen).cuda()
    h = _meta_head_params(1536, 2, gen)
    h.update(dim=2, off=0, slot=1)
    with pytest.raises(L.LnxError, match="lnx_meta_heads_supported"):
        ops.meta_heads_fwd(meta, [h], torch.zeros(4, 3, 1536, device="cuda"))


def test_layernorm_bwd_postponed_reductions_in_one_launch():
    """lnx_ln_bwd_args.defer + lnx_layernorm_bwd_flush (round 5): the column-sum second stages of several LayerNorm backward calls -- the
    widths and row counts of a training step's launch-stream LayerNorms: RoPE blocks, downsample layers, the tail -- in ONE launch.  Nothing
    reaches dw / db before the flush; afterwards they hold what the immediate path gives (float atomics over 64 slices either way: 1e-5), dx is
    untouched by the switch, a flush on another stream is refused, discard forgets them, and a 17th pending call flushes the first 16 itself."""
    import ctypes

    gen = g(5)
    cases = [(50944, 384), (13312, 768), (256, 768), (200704, 96), (12544, 384), (256, 384)]
    packs = []
    for M, C in cases:
        x = (torch.randn(M, C, generator=gen) + 0.3).cuda()
        dy = torch.randn(M, C, generator=gen).cuda().bfloat16()
        w = (torch.rand(C, generator=gen) + 0.5).cuda()
        packs.append((x, dy, w, x.mean(-1), (x.var(-1, unbiased=False) + 1e-5).rsqrt()))

    def run(defer):
        outs = []
        for x, dy, w, mean, rstd in packs:
            M, C = x.shape
            dx = torch.empty(M, C, device="cuda")
            dw, db = torch.full((C,), 0.5, device="cuda"), torch.full((C,), -0.5, device="cuda")
            ws = torch.empty(2048 * 2 * C, device="cuda")
            ops.layernorm_bwd(dy, x, w, mean, rstd, dx, dw=dw, db=db, ws=ws, defer=defer)
            outs.append((dx, dw, db, ws))
        return outs

    now = run(False)
    torch.cuda.synchronize()
    later = run(True)
    torch.cuda.synchronize()
    for dx, dw, db, _ in later:  # nothing summed yet
        assert torch.equal(dw, torch.full_like(dw, 0.5)) and torch.equal(db, torch.full_like(db, -0.5))
    other = torch.cuda.Stream()
    with pytest.raises(L.LnxError, match="another stream"):
        L.check(L.lib().lnx_layernorm_bwd_flush(ctypes.c_void_p(other.cuda_stream)), "lnx_layernorm_bwd_flush")
    ops.layernorm_bwd_flush()
    torch.cuda.synchronize()
    for (dx0, dw0, db0, _), (dx1, dw1, db1, _), (x, dy, w, mean, rstd) in zip(now, later, packs):
        assert torch.equal(dx0, dx1)
        scale = max(1.0, dw0.abs().max().item())
        torch.testing.assert_close(dw1, dw0, rtol=1e-5, atol=1e-5 * scale)
        torch.testing.assert_close(db1, db0, rtol=1e-5, atol=1e-5 * max(1.0, db0.abs().max().item()))
        xh = ((x - mean[:, None]) * rstd[:, None]).double()
        torch.testing.assert_close((dw1 - 0.5).double(), (dy.double() * xh).sum(0), rtol=1e-4, atol=1e-4 * scale)
    ops.layernorm_bwd_flush()  # nothing pending: a no-op
    # discard: pending calls are forgotten, dw stays
    x, dy, w, mean, rstd = packs[2]
    dw, db = torch.zeros(768, device="cuda"), torch.zeros(768, device="cuda")
    ws = torch.empty(2048 * 2 * 768, device="cuda")
    ops.layernorm_bwd(dy, x, w, mean, rstd, torch.empty_like(x), dw=dw, db=db, ws=ws, defer=True)
    assert L.lib().lnx_layernorm_bwd_discard() == 1
    ops.layernorm_bwd_flush()
    torch.cuda.synchronize()
    assert float(dw.abs().sum()) == 0.0
    # seventeen pending calls: the 17th makes the library flush the first sixteen
    many = []
    for i in range(17):
        dwi, dbi = torch.zeros(768, device="cuda"), torch.zeros(768, device="cuda")
        wsi = torch.empty(2048 * 2 * 768, device="cuda")
        ops.layernorm_bwd(dy, x, w, mean, rstd, torch.empty_like(x), dw=dwi, db=dbi, ws=wsi, defer=True)
        many.append((dwi, wsi, dbi))  # (everything a pending descriptor points at stays alive until the flush)
    torch.cuda.synchronize()
    assert all(float(m_[0].abs().sum()) > 0 for m_ in many[:16]) and float(many[16][0].abs().sum()) == 0.0
    ops.layernorm_bwd_flush()
    torch.cuda.synchronize()
    for m_ in many:
        torch.testing.assert_close(m_[0], many[0][0], rtol=1e-5, atol=1e-5 * max(1.0, many[0][0].abs().max().item()))
