"""GEMM kernels (lnx_gemm_nt / lnx_gemm_tn) against plain PyTorch fp32 on the same op."""
import ctypes as C
import os

import pytest
import torch

from linnaeus_amd import _lib as L

pytestmark = pytest.mark.gpu


def _stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def _ptr(t):
    return C.c_void_p(t.data_ptr()) if t is not None else None


def run_nt(A, W, dtype, out_f32, bias=None, act=0, aux=None, gamma=None, rowscale=None, rps=0, res=None, want_c2=False,
           c_map=(0, 0, 0), out_rows=None):
    M, K = A.shape
    N = W.shape[0]
    tdt = torch.float32 if dtype == L.F32 else torch.bfloat16
    odt = torch.float32 if out_f32 else tdt
    out = torch.full((out_rows or M, N), float("nan"), device="cuda", dtype=odt) if res is None else res.clone().to(odt)
    c2 = torch.empty(M, N, device="cuda", dtype=tdt) if want_c2 else None
    a = L.GemmArgs()
    a.dtype, a.M, a.N, a.K = dtype, M, N, K
    a.A, a.lda, a.W, a.ldw = _ptr(A), A.stride(0), _ptr(W), W.stride(0)
    a.C, a.ldc, a.out_f32 = _ptr(out), out.stride(0), int(out_f32)
    a.c_map = L.RowMap(*c_map)
    a.bias = _ptr(bias)
    a.c2, a.ldc2 = _ptr(c2), (c2.stride(0) if c2 is not None else 0)
    a.act, a.aux, a.ldaux = act, _ptr(aux), (aux.stride(0) if aux is not None else 0)
    a.gamma, a.rowscale, a.rows_per_sample = _ptr(gamma), _ptr(rowscale), rps
    a.res, a.ldres = _ptr(res), (res.stride(0) if res is not None else 0)
    L.check(L.lib().lnx_gemm_nt(C.byref(a), _stream()), "lnx_gemm_nt")
    torch.cuda.synchronize()
    return out, c2


SHAPES = [(2000, 384, 384), (4096, 1536, 384), (1300, 96, 1536), (1024, 1000, 768), (5000, 200, 128), (128, 128, 64), (256, 384, 96), (200, 96, 384), (333, 1000, 768), (64, 20, 768), (1, 7, 32), (777, 1152, 384)]


@pytest.mark.parametrize("M,N,K", SHAPES)
@pytest.mark.parametrize("dtype", [L.F32, L.BF16])
def test_nt_plain(M, N, K, dtype):
    g = torch.Generator(device="cpu").manual_seed(M * 7 + N * 3 + K)
    A = torch.randn(M, K, generator=g).cuda()
    W = torch.randn(N, K, generator=g).cuda() / K**0.5
    b = torch.randn(N, generator=g).cuda()
    tdt = torch.float32 if dtype == L.F32 else torch.bfloat16
    At, Wt = A.to(tdt), W.to(tdt)
    out, _ = run_nt(At, Wt, dtype, True, bias=b)
    ref = At.double() @ Wt.double().T + b.double()
    tol = 2e-5 if dtype == L.F32 else 2e-5  # operands are already rounded; accumulation is fp32 in both modes
    torch.testing.assert_close(out.double(), ref, rtol=tol, atol=tol * 4)
    if dtype == L.BF16:
        out2, _ = run_nt(At, Wt, dtype, False, bias=b)
        torch.testing.assert_close(out2.float(), ref.float(), rtol=8e-3, atol=8e-3)


@pytest.mark.parametrize("M,N,K", [(256, 1000, 384), (256, 384, 1000), (256, 20, 384), (256, 384, 24), (200, 33, 304), (7, 5, 8), (256, 300, 40), (32, 32, 32)])
@pytest.mark.parametrize("with_res", [False, True])
def test_nt_skinny_batch_rows(M, N, K, with_res):
    """M <= 256 in bf16 (the classification heads and their data gradients) takes the one-wave-per-tile kernel of
    gemm_skinny.hip: K tails that are not a multiple of 32, N tails, bias, fp32 residual, both output types."""
    g = torch.Generator(device="cpu").manual_seed(M * 7 + N * 3 + K)
    A = torch.randn(M, K, generator=g).cuda().bfloat16()
    W = (torch.randn(N, K, generator=g) / K**0.5).cuda().bfloat16()
    b = torch.randn(N, generator=g).cuda()
    res = torch.randn(M, N, generator=g).cuda() if with_res else None
    out, _ = run_nt(A, W, L.BF16, True, bias=b, res=res)
    ref = A.double() @ W.double().T + b.double() + (res.double() if with_res else 0.0)
    torch.testing.assert_close(out.double(), ref, rtol=2e-5, atol=8e-5)
    if not with_res:
        out2, _ = run_nt(A, W, L.BF16, False)
        torch.testing.assert_close(out2.float(), (A.double() @ W.double().T).float(), rtol=8e-3, atol=8e-3)


@pytest.mark.parametrize("M", [256, 128, 100, 7])
def test_nt_group_heads_in_one_launch(M):
    """lnx_gemm_nt_group, the two forms the model's tail uses (mFormerV1.py:536-541, four heads of 1000 / 300 / 80 / 20 classes on
    768 features).  Independent problems: every head's logits block of one flat buffer (padded leading dimensions) is bit for
    bit what lnx_gemm_nt writes for that head alone, and nothing outside the blocks' live columns is touched.  Accumulate: the
    data gradient wrt the features, res + sum_t dlogits_t . W_t over the padded class rows, against fp64 and against the
    head-by-head chain of single launches it replaces."""
    from linnaeus_amd import ops

    g = torch.Generator(device="cpu").manual_seed(11 + M)
    Cf, classes = 768, (1000, 300, 80, 20)
    lds = [(c + 7) // 8 * 8 for c in classes]
    feats = torch.randn(M, Cf, generator=g).cuda().bfloat16()
    Ws = [(torch.randn(c, Cf, generator=g) / Cf**0.5).cuda().bfloat16() for c in classes]
    bs = [torch.randn(c, generator=g).cuda() for c in classes]
    flat = torch.full((M * sum(lds),), float("nan"), device="cuda")
    flat1 = flat.clone()
    views, views1, off = [], [], 0
    for c, ld in zip(classes, lds):
        views.append(flat[off:off + M * ld].view(M, ld)[:, :c])
        views1.append(flat1[off:off + M * ld].view(M, ld)[:, :c])
        off += M * ld
    ops.gemm_nt_group([(feats, W, v, b, None) for W, v, b in zip(Ws, views, bs)])
    for W, v, b in zip(Ws, views1, bs):
        ops.gemm_nt(feats, W, v, bias=b)
    torch.cuda.synchronize()
    assert torch.equal(torch.nan_to_num(flat, nan=-7.0), torch.nan_to_num(flat1, nan=-7.0))  # same bits, same untouched padding
    for W, v, b in zip(Ws, views, bs):
        torch.testing.assert_close(v.double(), feats.double() @ W.double().T + b.double(), rtol=2e-5, atol=8e-5)

    # accumulate: d feats = res + sum_t dl_t . Wt_t^T with Wt_t = W_t^T stored [Cf, ld_t] (zero padding columns, as the plan's arena holds them)
    dls, Wts = [], []
    for c, ld, W in zip(classes, lds, Ws):
        d = torch.zeros(M, ld)
        d[:, :c] = torch.randn(M, c, generator=g) / 30
        dls.append(d.cuda().bfloat16())
        wt = torch.zeros(Cf, ld, device="cuda", dtype=torch.bfloat16)
        wt[:, :c] = W.T
        Wts.append(wt)
    res = torch.randn(M, Cf, generator=g).cuda()
    want = res.double() + sum(d.double() @ wt.double().T for d, wt in zip(dls, Wts))
    for with_res in (True, False):
        out = torch.full((M, Cf), float("nan"), device="cuda")
        probs = [(dls[0], Wts[0], out, None, res if with_res else None)] + [(d, wt, None, None, None) for d, wt in zip(dls[1:], Wts[1:])]
        ops.gemm_nt_group(probs, accumulate=True)
        torch.testing.assert_close(out.double(), want if with_res else want - res.double(), rtol=2e-5, atol=2e-5)
    chain = res.clone()
    for d, wt in zip(dls, Wts):
        ops.gemm_nt(d, wt, chain, res=chain)
    torch.testing.assert_close(out + res, chain, rtol=1e-5, atol=1e-5)  # (one accumulator chain against four fp32 round trips)

    with pytest.raises(L.LnxError):  # fp32 operands: not this kernel's (the plan then launches head by head)
        ops.gemm_nt_group([(feats.float(), Ws[0].float(), views[0], bs[0], None)])


@pytest.mark.parametrize("dtype", [L.F32, L.BF16])
def test_nt_asymmetric_identity(dtype):
    """A = I with an asymmetric W catches row/col swaps in the C write."""
    tdt = torch.float32 if dtype == L.F32 else torch.bfloat16
    K = 128
    A = torch.eye(K, device="cuda", dtype=tdt)
    W = (torch.arange(160 * K, device="cuda").reshape(160, K) % 251).to(tdt)
    out, _ = run_nt(A, W, dtype, True)
    assert torch.equal(out, W.float().T)


@pytest.mark.parametrize("dtype", [L.F32, L.BF16])
def test_nt_epilogues(dtype):
    tdt = torch.float32 if dtype == L.F32 else torch.bfloat16
    g = torch.Generator().manual_seed(5)
    Bsz, rows, K, N = 3, 50, 192, 96
    M = Bsz * rows
    A = torch.randn(M, K, generator=g).cuda().to(tdt)
    W = (torch.randn(N, K, generator=g) / K**0.5).cuda().to(tdt)
    b = torch.randn(N, generator=g).cuda()
    gam = torch.randn(N, generator=g).cuda()
    rs = torch.tensor([0.0, 1.25, 1.25]).cuda()
    res = torch.randn(M, N, generator=g).cuda()
    z = A.double() @ W.double().T + b.double()
    tol = dict(rtol=3e-5, atol=3e-5)
    # GELU with second output (pre-activation)
    out, c2 = run_nt(A, W, dtype, True, bias=b, act=L.ACT_GELU, want_c2=True)
    # bf16 mode evaluates erf by a polynomial with |GELU error| <= 5.8e-5 (common.hpp); fp32 mode uses libm erff
    gtol = tol if dtype == L.F32 else dict(rtol=3e-5, atol=1e-4)
    torch.testing.assert_close(out.double(), torch.nn.functional.gelu(z), **gtol)
    torch.testing.assert_close(c2.double(), z, rtol=1e-5 if dtype == L.F32 else 8e-3, atol=1e-5 if dtype == L.F32 else 8e-3)
    # LayerScale * DropPath + residual, pre-gamma value saved
    out, c2 = run_nt(A, W, dtype, True, bias=b, gamma=gam, rowscale=rs, rps=rows, res=res, want_c2=True)
    ref = res.double() + z * gam.double() * rs.double().repeat_interleave(rows)[:, None]
    torch.testing.assert_close(out.double(), ref, **tol)
    # GELU backward epilogue
    aux = torch.randn(M, N, generator=g).cuda().to(tdt)
    out, _ = run_nt(A, W, dtype, True, act=L.ACT_GELU_BWD, aux=aux)
    x = aux.double().requires_grad_(True)
    torch.nn.functional.gelu(x).sum().backward()
    torch.testing.assert_close(out.double(), (z - b.double()) * x.grad, rtol=1e-4, atol=1e-4)
    # round 3: GELU_D (C = GELU(v), c2 = GELU'(v)) and its backward partner MUL_AUX (v *= aux)
    out, c2 = run_nt(A, W, dtype, True, bias=b, act=L.ACT_GELU_D, want_c2=True)
    zz = z.clone().requires_grad_(True)
    torch.nn.functional.gelu(zz).sum().backward()
    torch.testing.assert_close(out.double(), torch.nn.functional.gelu(z), **gtol)
    torch.testing.assert_close(c2.double(), zz.grad, rtol=1e-5 if dtype == L.F32 else 8e-3, atol=1e-4 if dtype == L.F32 else 8e-3)
    out, _ = run_nt(A, W, dtype, True, act=L.ACT_MUL_AUX, aux=aux)
    torch.testing.assert_close(out.double(), (z - b.double()) * aux.double(), rtol=1e-4, atol=1e-4)
    # ReLU / ReLU backward
    out, _ = run_nt(A, W, dtype, True, bias=b, act=L.ACT_RELU)
    torch.testing.assert_close(out.double(), z.clamp_min(0), **tol)
    out, _ = run_nt(A, W, dtype, True, act=L.ACT_RELU_BWD, aux=aux)
    torch.testing.assert_close(out.double(), (z - b.double()) * (aux.double() > 0), **tol)
    # token row map: each sample gets E=3 extra leading rows
    E = 3
    canvas = torch.zeros(Bsz * (rows + E), N, device="cuda")
    out, _ = run_nt(A, W, dtype, True, bias=b, res=canvas, c_map=(rows, E, E), out_rows=Bsz * (rows + E))
    got = out.view(Bsz, rows + E, N)
    assert torch.equal(got[:, :E], torch.zeros_like(got[:, :E]))
    torch.testing.assert_close(got[:, E:].reshape(M, N).double(), z, **tol)


def _patches_nhwc(x):  # x [B,H,W,C] -> [B*Ho*Wo, 4C] with k = (kh*2+kw)*C + c
    B, H, W, Cc = x.shape
    p = x.view(B, H // 2, 2, W // 2, 2, Cc).permute(0, 1, 3, 2, 4, 5)
    return p.reshape(B * (H // 2) * (W // 2), 4 * Cc)


@pytest.mark.parametrize("small", [True, False])
@pytest.mark.parametrize("dtype", [L.F32, L.BF16])
def test_nt_patch2_modes(dtype, small):
    tdt = torch.float32 if dtype == L.F32 else torch.bfloat16
    g = torch.Generator().manual_seed(11)
    B, H, Wd, Cc, N = (2, 6, 10, 32, 48) if small else (3, 40, 36, 64, 200)
    x = torch.randn(B, H, Wd, Cc, generator=g).cuda().to(tdt)
    W = (torch.randn(N, 4 * Cc, generator=g) / (4 * Cc) ** 0.5).cuda().to(tdt)
    M = B * (H // 2) * (Wd // 2)
    out = torch.empty(M, N, device="cuda")
    a = L.GemmArgs()
    a.dtype, a.M, a.N, a.K = dtype, M, N, 4 * Cc
    a.A, a.W, a.ldw = _ptr(x), _ptr(W), W.stride(0)
    a.a_mode, a.Hin, a.Win, a.Cin = L.ADDR_PATCH2, H, Wd, Cc
    a.C, a.ldc, a.out_f32 = _ptr(out), N, 1
    L.check(L.lib().lnx_gemm_nt(C.byref(a), _stream()), "gemm patch2 A")
    torch.cuda.synchronize()
    ref = _patches_nhwc(x).double() @ W.double().T
    torch.testing.assert_close(out.double(), ref, rtol=3e-5, atol=3e-5)
    # scatter (data gradient of the 2x2 conv): dX_patches = dY . Wt^T, Wt = [4C, N]
    dY = torch.randn(M, N, generator=g).cuda().to(tdt)
    Wt = W.T.contiguous()
    dx = torch.full((B, H, Wd, Cc), float("nan"), device="cuda")
    a = L.GemmArgs()
    a.dtype, a.M, a.N, a.K = dtype, M, 4 * Cc, N
    a.A, a.lda, a.W, a.ldw = _ptr(dY), N, _ptr(Wt), N
    a.c_mode, a.Hin, a.Win, a.Cin = L.ADDR_PATCH2, H, Wd, Cc
    a.C, a.out_f32 = _ptr(dx), 1
    L.check(L.lib().lnx_gemm_nt(C.byref(a), _stream()), "gemm patch2 C")
    torch.cuda.synchronize()
    refp = dY.double() @ Wt.double().T
    torch.testing.assert_close(_patches_nhwc(dx).double(), refp, rtol=3e-5, atol=3e-5)


def run_tn(dY, A, dtype, bias=True, splits=0, k_perm_c=0, patch=None, ws=False, k_store=0, init=0.0):
    M, N = dY.shape
    K = A.shape[1] if patch is None else 4 * patch[2]
    dW = torch.full((N, K), init, device="cuda")
    db = torch.full((N,), init, device="cuda") if bias else None
    a = L.WgradArgs()
    a.dtype, a.M, a.N, a.K = dtype, M, N, K
    a.dY, a.lddy, a.A = _ptr(dY), dY.stride(0), _ptr(A)
    if patch is None:
        a.lda = A.stride(0)
    else:
        a.a_mode, a.Hin, a.Win, a.Cin = L.ADDR_PATCH2, *patch
    a.dW, a.lddw, a.db, a.splits, a.k_perm_c, a.k_store = _ptr(dW), K, _ptr(db), splits, k_perm_c, k_store
    if ws:
        wsb = torch.full((L.TN_WS_FLOATS,), float("nan"), device="cuda")
        a.ws, a.ws_floats = _ptr(wsb), wsb.numel()
    L.check(L.lib().lnx_gemm_tn(C.byref(a), _stream()), "lnx_gemm_tn")
    torch.cuda.synchronize()
    return dW, db


@pytest.mark.parametrize("M,N,K", [(8192, 96, 384), (6400, 1000, 768), (4096, 384, 96), (12800, 384, 1536), (256, 128, 128), (1000, 96, 384), (5000, 384, 96), (199, 1152, 384), (77, 24, 16), (4096, 768, 3072)])
@pytest.mark.parametrize("dtype", [L.F32, L.BF16])
def test_tn(M, N, K, dtype):
    tdt = torch.float32 if dtype == L.F32 else torch.bfloat16
    g = torch.Generator().manual_seed(M + N + K)
    dY = torch.randn(M, N, generator=g).cuda().to(tdt)
    A = torch.randn(M, K, generator=g).cuda().to(tdt)
    dW, db = run_tn(dY, A, dtype)
    ref = dY.double().T @ A.double()
    scale = M**0.5
    torch.testing.assert_close(dW.double(), ref, rtol=1e-4, atol=2e-5 * scale)
    torch.testing.assert_close(db.double(), dY.double().sum(0), rtol=1e-4, atol=2e-5 * scale)


@pytest.mark.parametrize("dtype", [L.F32, L.BF16])
def test_tn_asymmetric(dtype):
    """dY = one-hot rows picks single rows of A: catches transposed/permuted writes exactly."""
    tdt = torch.float32 if dtype == L.F32 else torch.bfloat16
    M, N, K = 128, 128, 256
    A = (torch.arange(M * K, device="cuda").reshape(M, K) % 253).to(tdt)
    dY = torch.zeros(M, N, device="cuda", dtype=tdt)
    idx = torch.arange(N, device="cuda")
    dY[(idx * 37) % M, idx] = 1
    dW, db = run_tn(dY, A, dtype, splits=1)
    assert torch.equal(dW, A.float()[(idx * 37) % M])
    assert torch.equal(db, torch.ones(N, device="cuda"))


@pytest.mark.parametrize("big", [False, True])
@pytest.mark.parametrize("dtype", [L.F32, L.BF16])
def test_tn_patch2_conv_weight_layout(dtype, big):
    """weight gradient of the 2x2 stride-2 patchify conv; `big` (M = 8960 patch rows, odd aspect) takes the pipelined
    LDS-DMA kernel with the gather in its source addresses, the small case the register-staged one"""
    tdt = torch.float32 if dtype == L.F32 else torch.bfloat16
    g = torch.Generator().manual_seed(3)
    B, H, Wd, Cc, N = (5, 56, 128, 48, 200) if big else (2, 8, 6, 32, 64)
    x = torch.randn(B, H, Wd, Cc, generator=g).cuda().to(tdt)
    M = B * (H // 2) * (Wd // 2)
    dY = torch.randn(M, N, generator=g).cuda().to(tdt)
    dW, db = run_tn(dY, x, dtype, k_perm_c=Cc, patch=(H, Wd, Cc), ws=big)
    torch.testing.assert_close(db.double(), dY.double().sum(0), rtol=1e-4, atol=2e-3)
    # reference: conv2d weight gradient in torch layout [N, C, 2, 2]
    xn = x.double().permute(0, 3, 1, 2).requires_grad_(False)
    w = torch.zeros(N, Cc, 2, 2, device="cuda", dtype=torch.double, requires_grad=True)
    y = torch.nn.functional.conv2d(xn, w, stride=2)
    y.backward(dY.double().view(B, H // 2, Wd // 2, N).permute(0, 3, 1, 2))
    torch.testing.assert_close(dW.double().view(N, Cc, 2, 2), w.grad, rtol=1e-4, atol=1e-4 if not big else 3e-3)


@pytest.mark.parametrize("M", [4096, 640])
@pytest.mark.parametrize("dtype", [L.F32, L.BF16])
def test_tn_padded_logits_rows(M, dtype):
    """dY rows padded to a multiple of 8 (N = 300 classes in 304 columns), as the head wgrad uses."""
    tdt = torch.float32 if dtype == L.F32 else torch.bfloat16
    g = torch.Generator().manual_seed(M)
    N, K = 300, 768
    buf = torch.zeros(M, 304, device="cuda", dtype=tdt)
    buf[:, :N] = torch.randn(M, N, generator=g).cuda().to(tdt)
    dY = buf[:, :N]
    A = torch.randn(M, K, generator=g).cuda().to(tdt)
    dW = torch.zeros(N, K, device="cuda")
    db = torch.zeros(N, device="cuda")
    a = L.WgradArgs()
    a.dtype, a.M, a.N, a.K = dtype, M, N, K
    a.dY, a.lddy, a.A, a.lda, a.dW, a.lddw, a.db = _ptr(dY), 304, _ptr(A), K, _ptr(dW), K, _ptr(db)
    L.check(L.lib().lnx_gemm_tn(C.byref(a), _stream()), "lnx_gemm_tn")
    torch.cuda.synchronize()
    torch.testing.assert_close(dW.double(), dY.double().T @ A.double(), rtol=1e-4, atol=2e-5 * M**0.5)
    torch.testing.assert_close(db.double(), dY.double().sum(0), rtol=1e-4, atol=2e-5 * M**0.5)


@pytest.mark.parametrize("N", [128, 256])
@pytest.mark.parametrize("kind", ["plain", "gelu_c2", "gelu_bwd", "res_f32", "bias_bf16"])
def test_nt_specialised_epilogues(kind, N):
    """Large aligned problems take the pipelined kernels with an epilogue compiled for its feature set (gemm2.hip,
    gemm_epilogue_fast; gemm3.hip): every form the plan launches, on a grid of 523 tiles (more than two per CU) and with K = 384
    (six K slices: the shortest loop the persistent kernel accepts).  N = 256 takes the 256x256-tile kernel; N = 128 the
    persistent 256x128 kernel (gemm_nt_v7) for the forms it is preferred for and the one-shot 256x128 kernel for the others."""
    M, K = 256 * 523, 384
    g = torch.Generator().manual_seed(11)
    A = (torch.randn(M, K, generator=g)).cuda().bfloat16()
    W = (torch.randn(N, K, generator=g) / K**0.5).cuda().bfloat16()
    b = torch.randn(N, generator=g).cuda()
    z = A.double() @ W.double().T + b.double()
    if kind == "plain":
        out, _ = run_nt(A, W, L.BF16, True, bias=b)
        torch.testing.assert_close(out.double(), z, rtol=2e-5, atol=1e-4)
    elif kind == "bias_bf16":
        out, _ = run_nt(A, W, L.BF16, False, bias=b)
        torch.testing.assert_close(out.float(), z.float(), rtol=8e-3, atol=8e-3)
        out, _ = run_nt(A, W, L.BF16, False)
        torch.testing.assert_close(out.float(), (z - b.double()).float(), rtol=8e-3, atol=8e-3)
    elif kind == "gelu_c2":
        out, c2 = run_nt(A, W, L.BF16, False, bias=b, act=L.ACT_GELU, want_c2=True)
        torch.testing.assert_close(c2.float(), z.float(), rtol=8e-3, atol=8e-3)
        torch.testing.assert_close(out.float(), torch.nn.functional.gelu(z).float(), rtol=8e-3, atol=8e-3)
        out, c2 = run_nt(A, W, L.BF16, False, bias=b, act=L.ACT_GELU_D, want_c2=True)  # derivative as the second output
        zz = z.clone().requires_grad_(True)
        torch.nn.functional.gelu(zz).sum().backward()
        torch.testing.assert_close(c2.float(), zz.grad.float(), rtol=8e-3, atol=8e-3)
        torch.testing.assert_close(out.float(), torch.nn.functional.gelu(z).float(), rtol=8e-3, atol=8e-3)
    elif kind == "gelu_bwd":
        aux = torch.randn(M, N, generator=g).cuda().bfloat16()
        out, _ = run_nt(A, W, L.BF16, False, act=L.ACT_GELU_BWD, aux=aux)
        x = aux.double().requires_grad_(True)
        torch.nn.functional.gelu(x).sum().backward()
        torch.testing.assert_close(out.float(), ((z - b.double()) * x.grad).float(), rtol=1e-2, atol=1e-2)
        out, _ = run_nt(A, W, L.BF16, False, act=L.ACT_MUL_AUX, aux=aux)
        torch.testing.assert_close(out.float(), ((z - b.double()) * aux.double()).float(), rtol=1e-2, atol=1e-2)
    else:
        res = torch.randn(M, N, generator=g).cuda()
        rs = (torch.rand(523, generator=g) > 0.3).float().cuda() / 0.7
        ref = res.double() + z * rs.double().repeat_interleave(256)[:, None]
        out, _ = run_nt(A, W, L.BF16, True, bias=b, rowscale=rs, rps=256, res=res)
        torch.testing.assert_close(out.double(), ref, rtol=2e-5, atol=2e-4)
    # the dispatcher's own record; which of the pipelined families a form takes is pinned at the benchmark's shapes below
    assert L.lib().lnx_last_nt_kernel() in (L.NT_KERNEL_V2, L.NT_KERNEL_V4, L.NT_KERNEL_V7, L.NT_KERNEL_V9)


# ----------------------------------------------------------------------------------------------------
# Round 4: the persistent kernel the benchmark's RoPE-block products run on (gemm3.hip: gemm_nt_v7 / gemm_nt_v9), at the
# benchmark's own M = 256 x 199 = 50 944 rows, every epilogue form x every (N, K) of mFormerV1_sm's stage 3, against fp64.
# Reference math: rope_2d_mhsa.py:432 (qkv), :500 (proj), blocks/mlp.py:46-66 (fc1 / fc2) and their autograd data gradients.
# ----------------------------------------------------------------------------------------------------
M_SM = 256 * 199
# (form, N, K) the B = 256 plan launches and the default dispatch is expected to give to the persistent kernels
PLAN_PERSISTENT = {("plain", 384, 384), ("plain", 384, 1152), ("plain", 384, 1536), ("bias", 1152, 384), ("res_f32", 384, 384),
                   ("res_f32", 384, 1536), ("mul_aux", 1536, 384), ("fc1d", 1536, 384)}


def _gpu_randn(shape, seed, scale=1.0):
    g = torch.Generator(device="cuda").manual_seed(seed)
    return torch.randn(*shape, generator=g, device="cuda") * scale


def _check_form(form, M, N, K, rows_per_sample=199):
    """One NT product of the given epilogue form against the fp64 product of the same (bf16-rounded) operands."""
    A = _gpu_randn((M, K), 1 + M + N + K).bfloat16()
    W = _gpu_randn((N, K), 2 + N * 3 + K, K**-0.5).bfloat16()
    b = _gpu_randn((N,), 3)
    z0 = A.double() @ W.double().T
    z = z0 + b.double()
    bt = dict(rtol=8e-3, atol=8e-3)  # bf16 output: one rounding of an O(1) value
    if form == "plain":
        out, _ = run_nt(A, W, L.BF16, False)
        torch.testing.assert_close(out.double(), z0, **bt)
    elif form == "bias":
        out, _ = run_nt(A, W, L.BF16, False, bias=b)
        torch.testing.assert_close(out.double(), z, **bt)
    elif form == "res_f32":  # proj / fc2: x + DropPath_scale[sample] * (A W^T + b), fp32 residual stream
        res = _gpu_randn((M, N), 4)
        nb = -(-M // rows_per_sample)
        rs = (torch.rand(nb, device="cuda", generator=torch.Generator(device="cuda").manual_seed(5)) > 0.2).float() / 0.8
        out, _ = run_nt(A, W, L.BF16, True, bias=b, rowscale=rs, rps=rows_per_sample, res=res)
        ref = res.double() + z * rs.double().repeat_interleave(rows_per_sample)[:M, None]
        torch.testing.assert_close(out.double(), ref, rtol=2e-5, atol=3e-4)
    elif form == "mul_aux":  # fc2 data gradient: dH = (dY W2) * GELU'(h), the factor saved by the forward
        aux = _gpu_randn((M, N), 6).bfloat16()
        out, _ = run_nt(A, W, L.BF16, False, act=L.ACT_MUL_AUX, aux=aux)
        torch.testing.assert_close(out.double(), z0 * aux.double(), rtol=1e-2, atol=1e-2)
    elif form == "gelu_bwd":  # the same with the derivative evaluated in the epilogue (fp8 blocks, inference-free plans)
        aux = _gpu_randn((M, N), 6).bfloat16()
        out, _ = run_nt(A, W, L.BF16, False, act=L.ACT_GELU_BWD, aux=aux)
        x = aux.double().requires_grad_(True)
        torch.nn.functional.gelu(x).sum().backward()
        torch.testing.assert_close(out.double(), z0 * x.grad, rtol=1e-2, atol=1e-2)
    elif form == "fc1":  # GELU(h) and the pre-activation h
        out, c2 = run_nt(A, W, L.BF16, False, bias=b, act=L.ACT_GELU, want_c2=True)
        torch.testing.assert_close(c2.double(), z, **bt)
        torch.testing.assert_close(out.double(), torch.nn.functional.gelu(z), rtol=1.2e-2, atol=1.2e-2)
    elif form == "fc1d":  # what the training plan's fc1 launches: GELU(h) and GELU'(h)
        out, c2 = run_nt(A, W, L.BF16, False, bias=b, act=L.ACT_GELU_D, want_c2=True)
        zz = z.clone().requires_grad_(True)
        torch.nn.functional.gelu(zz).sum().backward()
        torch.testing.assert_close(c2.double(), zz.grad, **bt)
        torch.testing.assert_close(out.double(), torch.nn.functional.gelu(z), **bt)
    elif form == "bias_gelu":  # an inference plan's fc1: one output
        out, _ = run_nt(A, W, L.BF16, False, bias=b, act=L.ACT_GELU)
        torch.testing.assert_close(out.double(), torch.nn.functional.gelu(z), **bt)
    else:
        raise AssertionError(form)
    return L.lib().lnx_last_nt_kernel()


PERSISTENT = (L.NT_KERNEL_V7, L.NT_KERNEL_V9)


@pytest.mark.parametrize("K", [384, 1152, 1536])
@pytest.mark.parametrize("N", [384, 1152, 1536])
@pytest.mark.parametrize("form", ["plain", "bias", "res_f32", "mul_aux", "gelu_bwd", "fc1", "fc1d"])
def test_nt_persistent_kernel_at_benchmark_rows(form, N, K, monkeypatch):
    """Default dispatch at M = 50 944: the products of PLAN_PERSISTENT must really run on a persistent kernel (the dispatcher's
    own record, lnx_last_nt_kernel), every other (form, N, K) on whatever the dispatcher prefers -- all against fp64."""
    monkeypatch.delenv("LNX_NT_V7", raising=False)
    monkeypatch.delenv("LNX_NT_V9", raising=False)
    kind = _check_form(form, M_SM, N, K)
    if (form, N, K) in PLAN_PERSISTENT:
        assert kind in PERSISTENT, (form, N, K, kind)
    else:
        assert kind in (L.NT_KERNEL_V2, L.NT_KERNEL_V4) + PERSISTENT, kind


@pytest.mark.parametrize("M,N,K", [(M_SM, 384, 384), (M_SM, 1152, 384), (M_SM, 1536, 384), (M_SM, 384, 1536), (M_SM, 1536, 1536), (M_SM, 1152, 1152),
                                   (M_SM - 37, 448, 384), (M_SM - 37, 448, 1152), (128 * 199, 384, 384), (13312, 768, 768), (13312, 2304, 768), (13312, 768, 3072)])
@pytest.mark.parametrize("form", ["plain", "bias", "res_f32", "mul_aux", "gelu_bwd", "fc1"])
def test_nt_v7_forced_every_form(form, M, N, K, monkeypatch):
    """LNX_NT_V7=1: gemm_nt_v7 wherever it can run -- both peelings (K / 64 < 17 and >= 17), a ragged last row tile with a column
    count that is not a multiple of 128 (M = 50 907, N = 448), the 128-image batch of BASELINE config 3 and the stage-4 shapes."""
    monkeypatch.setenv("LNX_NT_V7", "1")
    kind = _check_form(form, M, N, K, rows_per_sample=199 if M != 13312 else 52)
    assert kind == L.NT_KERNEL_V7, kind


@pytest.mark.parametrize("M,N,K,kpc", [(8192, 96, 384, 0), (12800, 384, 1536, 0), (50944, 1536, 384, 0), (6400, 1000, 768, 0), (8192, 192, 384, 96)])
def test_tn_workspace_reduction(M, N, K, kpc):
    """Split-K through the workspace (partial tiles + fixed-order reduce kernel) instead of atomics: same result as
    fp64, accumulates onto what dW/db already hold, honours the conv-weight column permutation, and is bit-reproducible."""
    g = torch.Generator().manual_seed(M + N + K)
    dY = torch.randn(M, N, generator=g).cuda().bfloat16()
    A = torch.randn(M, K, generator=g).cuda().bfloat16()
    dW, db = run_tn(dY, A, L.BF16, ws=True, k_perm_c=kpc, init=0.5)
    ref = dY.double().T @ A.double()
    if kpc:
        ref = ref.reshape(N, K // kpc, kpc).transpose(1, 2).reshape(N, K)
    torch.testing.assert_close(dW.double(), ref + 0.5, rtol=1e-4, atol=2e-5 * M**0.5)
    torch.testing.assert_close(db.double(), dY.double().sum(0) + 0.5, rtol=1e-4, atol=2e-5 * M**0.5)
    dW2, db2 = run_tn(dY, A, L.BF16, ws=True, k_perm_c=kpc, init=0.5)
    if N * K >= 0.7 * (-(-N // 256) * 256) * (-(-K // 128) * 128) or N * K >= 0.7 * (-(-N // 128) * 128) * (-(-K // 256) * 256):
        assert torch.equal(dW, dW2) and torch.equal(db, db2)  # well-filled tiles take the workspace path: fixed summation order
    else:
        torch.testing.assert_close(dW, dW2, rtol=1e-5, atol=1e-3)  # poorly filled tiles stay on atomics (fewer bytes)


@pytest.mark.parametrize("M,N,K", [(M_SM, 1536, 384), (M_SM, 1536, 1536), (M_SM, 256, 256), (M_SM - 37, 512, 384), (128 * 199, 1536, 384), (13312, 768, 768),
                                   (13312, 2304, 768), (13312, 3072, 768), (13312, 768, 3072), (1024, 256, 256), (2048, 512, 1024)])
@pytest.mark.parametrize("form", ["plain", "bias", "res_f32", "mul_aux", "gelu_bwd", "fc1", "fc1d", "bias_gelu"])
def test_nt_v9_forced_every_form(form, M, N, K, monkeypatch):
    """LNX_NT_V9=1: the persistent 256x256 kernel (gemm5.hip) wherever it can run -- every epilogue form, the shortest K loop it
    accepts (8 slices of 32) and long ones, a ragged last row tile (which drains its stores instead of counting them), launches with
    fewer than 8 workgroups (fewer tile shares than XCDs) and the stage-4 / xl-like shapes; against fp64."""
    monkeypatch.setenv("LNX_NT_V9", "1")
    monkeypatch.setenv("LNX_NT_V7", "0")
    kind = _check_form(form, M, N, K, rows_per_sample=199 if M % 199 == 0 or M == M_SM - 37 else 52)
    assert kind == L.NT_KERNEL_V9, kind


# ----------------------------------------------------------------------------------------------------
# Round 5: the (N, K) pairs of mFormerV1_xl @224 at B = 128 (M = 128 x 199 = 25 472, C = 1024) and of
# mFormerV1_lg @384 at B = 64 (M = 64 x 580 = 37 120, C = 768) -- the shapes `bench.py --arch xl --batch 128` and `--arch lg --img 384
# --batch 64` time -- under the default dispatch and with gemm_nt_v9 forced, every epilogue form, against fp64.
# ----------------------------------------------------------------------------------------------------
M_XL, M_LG = 128 * 199, 64 * 580
XL_LG_NK = [(M_XL, 1024, 1024), (M_XL, 3072, 1024), (M_XL, 4096, 1024), (M_XL, 1024, 4096), (M_XL, 1024, 3072),
            (M_LG, 768, 768), (M_LG, 2304, 768), (M_LG, 3072, 768), (M_LG, 768, 3072), (M_LG, 768, 2304)]


@pytest.mark.parametrize("M,N,K", XL_LG_NK)
@pytest.mark.parametrize("form", ["plain", "bias", "res_f32", "mul_aux", "gelu_bwd", "fc1", "fc1d", "bias_gelu"])
def test_nt_default_dispatch_at_xl_lg_rows(form, M, N, K, monkeypatch):
    """Default dispatch: every N here is a multiple of 256 on 400+ tiles, so the persistent kernels take all of them -- gemm_nt_v9 except
    the GELU'-multiply data gradients, which gemm2.hip's nt_v7_preferred gives to gemm_nt_v7 (the dispatcher's own record)."""
    monkeypatch.delenv("LNX_NT_V7", raising=False)
    monkeypatch.delenv("LNX_NT_V9", raising=False)
    kind = _check_form(form, M, N, K, rows_per_sample=199 if M == M_XL else 580)
    assert kind == (L.NT_KERNEL_V7 if form == "mul_aux" else L.NT_KERNEL_V9), (form, M, N, K, kind)


@pytest.mark.parametrize("M,N,K", XL_LG_NK)
@pytest.mark.parametrize("form", ["plain", "bias", "res_f32", "mul_aux", "gelu_bwd", "fc1", "fc1d", "bias_gelu"])
def test_nt_v9_forced_at_xl_lg_rows(form, M, N, K, monkeypatch):
    monkeypatch.setenv("LNX_NT_V9", "1")
    monkeypatch.setenv("LNX_NT_V7", "0")
    kind = _check_form(form, M, N, K, rows_per_sample=199 if M == M_XL else 580)
    assert kind == L.NT_KERNEL_V9, kind


@pytest.mark.parametrize("M,N,K", XL_LG_NK)
def test_tn_at_xl_lg_rows(M, N, K):
    """The weight-gradient products of the same layers (dW[N, K] = dY^T A over M rows; gemm_tn_v2 + the workspace reduce, as the plan calls
    it) against fp64: accumulates onto what dW / db hold, fixed summation order (bit-reproducible)."""
    dY = _gpu_randn((M, N), 11 + N).bfloat16()
    A = _gpu_randn((M, K), 12 + K).bfloat16()
    dW, db = run_tn(dY, A, L.BF16, ws=True, init=0.25)
    ref = dY.double().T @ A.double()
    torch.testing.assert_close(dW.double(), ref + 0.25, rtol=1e-4, atol=2e-5 * M**0.5)
    torch.testing.assert_close(db.double(), dY.double().sum(0) + 0.25, rtol=1e-4, atol=2e-5 * M**0.5)
    dW2, db2 = run_tn(dY, A, L.BF16, ws=True, init=0.25)
    assert torch.equal(dW, dW2) and torch.equal(db, db2)


@pytest.mark.parametrize("margin", [0, 5, 248])
def test_persistent_kernels_cover_every_tile_once_for_any_grid(margin, monkeypatch):
    """The drawn-tile schedulers (common.hpp: per-XCD shares and counters) under odd tile counts and odd grids: lnx_set_cu_margin(5) gives a grid
    that is not a multiple of 8 (uneven shares), 248 leaves 8 workgroups for hundreds of tiles (many draws per workgroup), random M / N put
    ragged row tiles and 1..7 tiles into a share.  Every output element must come out right -- a tile drawn twice is harmless, a tile never
    drawn is not -- and the launches that follow must find their counters at zero (the second loop re-uses the same stream's counter set)."""
    import random

    rnd = random.Random(1234 + margin)
    L.check(L.lib().lnx_set_cu_margin(margin), "lnx_set_cu_margin")
    try:
        for force in ("LNX_NT_V7", "LNX_NT_V9"):
            monkeypatch.setenv("LNX_NT_V7", "1" if force == "LNX_NT_V7" else "0")
            monkeypatch.setenv("LNX_NT_V9", "1" if force == "LNX_NT_V9" else "0")
            for _ in range(6):
                M = rnd.randrange(1024, 40000)
                N = rnd.choice([256, 512, 768, 1024]) if force == "LNX_NT_V9" else 64 * rnd.randrange(2, 20)
                K = rnd.choice([384, 512, 1152])
                A = _gpu_randn((M, K), M + N).bfloat16()
                W = _gpu_randn((N, K), N + K, K**-0.5).bfloat16()
                b = _gpu_randn((N,), 3)
                out, _ = run_nt(A, W, L.BF16, False, bias=b)
                assert L.lib().lnx_last_nt_kernel() == (L.NT_KERNEL_V7 if force == "LNX_NT_V7" else L.NT_KERNEL_V9), (force, M, N, K)
                ref = A.float() @ W.float().t() + b
                torch.testing.assert_close(out.float(), ref, rtol=1e-2, atol=1e-2, msg=lambda m: f"{force} M={M} N={N} K={K} margin={margin}: {m}")
    finally:
        L.check(L.lib().lnx_set_cu_margin(0), "lnx_set_cu_margin")


def test_persistent_kernels_on_two_streams_do_not_share_counters():
    """Two streams launching persistent NT products at the same time (what two plans in one process, or a plan beside a user's own
    stream, do): each stream draws from its own counter set (api.cpp: tile_slot_of), so tiles of one launch are never handed to the
    other.  40 interleaved launches per stream, every result checked."""
    M, N, K = M_SM, 384, 1152
    os.environ["LNX_NT_V7"] = "1"
    try:
        s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
        A1, A2 = _gpu_randn((M, K), 11).bfloat16(), _gpu_randn((M, K), 12).bfloat16()
        W1, W2 = _gpu_randn((N, K), 13, K**-0.5).bfloat16(), _gpu_randn((2 * N, K), 14, K**-0.5).bfloat16()
        ref1, ref2 = (A1.float() @ W1.float().t()), (A2.float() @ W2.float().t())
        torch.cuda.synchronize()
        outs1 = [torch.empty(M, N, device="cuda", dtype=torch.bfloat16) for _ in range(4)]
        outs2 = [torch.empty(M, 2 * N, device="cuda", dtype=torch.bfloat16) for _ in range(4)]

        def launch(A, W, out, stream):
            a = L.GemmArgs()
            a.dtype, a.M, a.N, a.K = L.BF16, A.shape[0], W.shape[0], A.shape[1]
            a.A, a.lda, a.W, a.ldw, a.C, a.ldc = _ptr(A), A.stride(0), _ptr(W), W.stride(0), _ptr(out), out.stride(0)
            L.check(L.lib().lnx_gemm_nt(C.byref(a), C.c_void_p(stream.cuda_stream)), "lnx_gemm_nt")

        for i in range(40):
            launch(A1, W1, outs1[i % 4], s1)
            launch(A2, W2, outs2[i % 4], s2)
        torch.cuda.synchronize()
        assert L.lib().lnx_last_nt_kernel() == L.NT_KERNEL_V7
        for o in outs1:
            torch.testing.assert_close(o.float(), ref1, rtol=1e-2, atol=1e-2)
        for o in outs2:
            torch.testing.assert_close(o.float(), ref2, rtol=1e-2, atol=1e-2)
    finally:
        os.environ.pop("LNX_NT_V7", None)


def test_tn_deferred_reduces_in_one_launch():
    """lnx_wgrad_args.defer + lnx_gemm_tn_flush (round 4): the four weight-gradient products of a RoPE block (mFormerV1_sm stage 3 at
    the benchmark's M) leave their split-K partial tiles in separate workspace regions; nothing reaches dW / db before the flush, one
    launch then sums them all -- same numbers as the immediate path, bit for bit (same partial tiles, same summation order).  A
    ninth pending product flushes the first eight by itself."""
    M = 256 * 199
    shapes = [(384, 1536), (1536, 384), (384, 384), (1152, 384)]  # fc2, fc1, proj, qkv: (N, K)
    g = torch.Generator(device="cuda").manual_seed(7)
    ops = []
    for i, (N, K) in enumerate(shapes):
        dY = torch.randn(M, N, device="cuda", generator=g).bfloat16()
        A = torch.randn(M, K, device="cuda", generator=g).bfloat16()
        ops.append((dY, A, N, K))
    ws = [torch.full((L.TN_WS_FLOATS,), float("nan"), device="cuda") for _ in shapes]

    def run(defer):
        outs = []
        for (dY, A, N, K), w in zip(ops, ws):
            dW = torch.full((N, K), 0.25, device="cuda")
            db = torch.full((N,), 0.25, device="cuda")
            a = L.WgradArgs()
            a.dtype, a.M, a.N, a.K = L.BF16, M, N, K
            a.dY, a.lddy, a.A, a.lda = _ptr(dY), N, _ptr(A), K
            a.dW, a.lddw, a.db = _ptr(dW), K, _ptr(db)
            a.ws, a.ws_floats, a.defer = _ptr(w), w.numel(), int(defer)
            L.check(L.lib().lnx_gemm_tn(C.byref(a), _stream()), "lnx_gemm_tn")
            outs.append((dW, db))
        return outs

    now = run(False)
    torch.cuda.synchronize()
    later = run(True)
    torch.cuda.synchronize()
    for dW, db in later:  # nothing summed yet
        assert torch.equal(dW, torch.full_like(dW, 0.25)) and torch.equal(db, torch.full_like(db, 0.25))
    L.check(L.lib().lnx_gemm_tn_flush(_stream()), "lnx_gemm_tn_flush")
    torch.cuda.synchronize()
    for (dW0, db0), (dW1, db1), (dY, A, N, K) in zip(now, later, ops):
        assert torch.equal(dW0, dW1) and torch.equal(db0, db1)
        torch.testing.assert_close(dW1.double(), dY.double().T @ A.double() + 0.25, rtol=1e-4, atol=2e-5 * M**0.5)
    L.check(L.lib().lnx_gemm_tn_flush(_stream()), "lnx_gemm_tn_flush")  # nothing pending: a no-op
    # nine pending products: the ninth makes the library flush the first eight
    dY, A, N, K = ops[2]
    many = []
    for i in range(9):
        dW = torch.zeros(N, K, device="cuda")
        w = torch.empty(L.TN_WS_FLOATS, device="cuda")
        a = L.WgradArgs()
        a.dtype, a.M, a.N, a.K = L.BF16, M, N, K
        a.dY, a.lddy, a.A, a.lda, a.dW, a.lddw = _ptr(dY), N, _ptr(A), K, _ptr(dW), K
        a.ws, a.ws_floats, a.defer = _ptr(w), w.numel(), 1
        L.check(L.lib().lnx_gemm_tn(C.byref(a), _stream()), "lnx_gemm_tn")
        many.append((dW, w))
    torch.cuda.synchronize()
    assert all(bool(dW.abs().sum() > 0) for dW, _ in many[:8]) and float(many[8][0].abs().sum()) == 0.0
    L.check(L.lib().lnx_gemm_tn_flush(_stream()), "lnx_gemm_tn_flush")
    torch.cuda.synchronize()
    assert all(torch.equal(dW, many[0][0]) for dW, _ in many)


def test_tn_deferred_reduces_belong_to_their_stream_and_can_be_discarded():
    """a flush asked for on another stream than the postponed products' is refused (it would order the reduces behind the wrong
    work) and leaves them pending; lnx_gemm_tn_discard() forgets them without touching dW -- what lnx_plan_backward does on entry and
    on its error paths, so that no descriptor with raw workspace / gradient pointers outlives a step that failed."""
    M, N, K = 8192, 384, 384
    g = torch.Generator(device="cuda").manual_seed(9)
    dY = torch.randn(M, N, device="cuda", generator=g).bfloat16()
    A = torch.randn(M, K, device="cuda", generator=g).bfloat16()
    dW = torch.zeros(N, K, device="cuda")
    w = torch.empty(L.TN_WS_FLOATS, device="cuda")
    a = L.WgradArgs()
    a.dtype, a.M, a.N, a.K = L.BF16, M, N, K
    a.dY, a.lddy, a.A, a.lda, a.dW, a.lddw = _ptr(dY), N, _ptr(A), K, _ptr(dW), K
    a.ws, a.ws_floats, a.defer = _ptr(w), w.numel(), 1
    other = torch.cuda.Stream()
    L.check(L.lib().lnx_gemm_tn(C.byref(a), _stream()), "lnx_gemm_tn")
    with pytest.raises(L.LnxError, match="another stream"):
        L.check(L.lib().lnx_gemm_tn_flush(C.c_void_p(other.cuda_stream)), "lnx_gemm_tn_flush")
    torch.cuda.synchronize()
    assert float(dW.abs().sum()) == 0.0
    assert L.lib().lnx_gemm_tn_discard() == 1
    L.check(L.lib().lnx_gemm_tn_flush(_stream()), "lnx_gemm_tn_flush")  # nothing left: a no-op
    torch.cuda.synchronize()
    assert float(dW.abs().sum()) == 0.0
    assert L.lib().lnx_gemm_tn_discard() == 0
    # and the ordinary path still works afterwards
    L.check(L.lib().lnx_gemm_tn(C.byref(a), _stream()), "lnx_gemm_tn")
    L.check(L.lib().lnx_gemm_tn_flush(_stream()), "lnx_gemm_tn_flush")
    torch.cuda.synchronize()
    torch.testing.assert_close(dW.double(), dY.double().T @ A.double(), rtol=1e-4, atol=2e-5 * M**0.5)
