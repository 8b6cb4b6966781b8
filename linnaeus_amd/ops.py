"""Tensor-level wrappers over the C ABI (one function per entry point of include/lnx.h).

These take torch CUDA tensors only to obtain device pointers, sizes and the current HIP
stream; all arithmetic happens in liblnx_hip.so.  Used by the per-op parity tests and by
the model glue.  Nothing here has a CPU path.
"""
from __future__ import annotations

import ctypes as C
from typing import Optional, Tuple

import torch

from . import _lib as L
from ._lib import F32, BF16  # noqa: F401

_TORCH_DT = {F32: torch.float32, BF16: torch.bfloat16}


def code_of(t: torch.Tensor) -> int:
    if t.dtype == torch.float32:
        return F32
    if t.dtype == torch.bfloat16:
        return BF16
    raise L.LnxError(f"unsupported tensor dtype {t.dtype}")


def torch_dtype(code: int) -> torch.dtype:
    return _TORCH_DT[code]


def _p(t: Optional[torch.Tensor]):
    if t is None:
        return None
    if not t.is_cuda:
        raise L.LnxError("linnaeus_amd kernels need CUDA(HIP) tensors; there is no CPU fallback")
    return C.c_void_p(t.data_ptr())


def _stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def _map(m) -> L.RowMap:
    return L.RowMap(*(m or (0, 0, 0)))


def gemm_nt(A, W, out, *, M=None, N=None, K=None, lda=None, ldw=None, ldc=None, bias=None, act=L.ACT_NONE, aux=None, c2=None,
            gamma=None, rowscale=None, rows_per_sample=0, res=None, c_map=None, a_patch=None, c_patch=None, dtype=None):
    """out = epilogue(A . W^T); see lnx_gemm_args.  a_patch / c_patch = (Hin, Win, Cin)."""
    a = L.GemmArgs()
    a.dtype = dtype if dtype is not None else code_of(W)
    a.M = M if M is not None else A.shape[0]
    a.N = N if N is not None else W.shape[0]
    a.K = K if K is not None else W.shape[1]
    a.A, a.lda = _p(A), (lda if lda is not None else (A.stride(0) if a_patch is None else 0))
    a.W, a.ldw = _p(W), (ldw if ldw is not None else W.stride(0))
    a.C, a.ldc = _p(out), (ldc if ldc is not None else (out.stride(0) if c_patch is None else 0))
    a.out_f32 = int(out.dtype == torch.float32)
    if a_patch is not None:
        a.a_mode, a.Hin, a.Win, a.Cin = L.ADDR_PATCH2, *a_patch
    if c_patch is not None:
        a.c_mode, a.Hin, a.Win, a.Cin = L.ADDR_PATCH2, *c_patch
    a.c_map = _map(c_map)
    a.bias = _p(bias)
    a.c2, a.ldc2 = _p(c2), (c2.stride(0) if c2 is not None else 0)
    a.act, a.aux, a.ldaux = act, _p(aux), (aux.stride(0) if aux is not None else 0)
    a.gamma, a.rowscale, a.rows_per_sample = _p(gamma), _p(rowscale), rows_per_sample
    a.res, a.ldres = _p(res), ((res.stride(0) if c_patch is None else 0) if res is not None else 0)
    L.check(L.lib().lnx_gemm_nt(C.byref(a), _stream()), "lnx_gemm_nt")
    return out


def gemm_nt_group(problems, accumulate=False):
    """Several small bf16 products in one launch (lnx_gemm_nt_group).  problems: [(A, W, out, bias or None, res or None), ...];
    accumulate: out_0 = bias_0 + res_0 + sum_j A_j . W_j^T (the outputs / epilogue operands of the other entries are ignored: pass None)."""
    arr = (L.GemmArgs * len(problems))()
    for a, (A, W, out, bias, res) in zip(arr, problems):
        a.dtype = code_of(W)
        a.M, a.N, a.K = A.shape[0], W.shape[0], W.shape[1]
        a.A, a.lda, a.W, a.ldw = _p(A), A.stride(0), _p(W), W.stride(0)
        ref = out if out is not None else problems[0][2]
        a.C, a.ldc, a.out_f32 = _p(out), (out.stride(0) if out is not None else 0), int(ref.dtype == torch.float32)
        a.bias = _p(bias)
        a.res, a.ldres = _p(res), (res.stride(0) if res is not None else 0)
    if not L.lib().lnx_gemm_nt_group_ok(arr, len(problems), int(accumulate)):
        raise L.LnxError("lnx_gemm_nt_group: the problem list does not qualify")
    L.check(L.lib().lnx_gemm_nt_group(arr, len(problems), int(accumulate), _stream()), "lnx_gemm_nt_group")


def quantize_fp8(x, *, amax=None):
    """(x8, scale): x8 = e4m3fn(x * 448 / max|x|) as a torch.float8_e4m3fn tensor, scale = max|x| / 448 (device scalar).
    `amax`: a device scalar to use instead of this tensor's own maximum (delayed scaling)."""
    x2 = x.reshape(-1, x.shape[-1])
    if amax is None:
        amax = torch.zeros(1, device=x.device, dtype=torch.float32)
        L.check(L.lib().lnx_amax(_p(x2), code_of(x2), C.c_int64(x2.stride(0)), x2.shape[0], x2.shape[1], _p(amax), _stream()), "lnx_amax")
    y = torch.empty(x2.shape, device=x.device, dtype=torch.uint8)
    scale = torch.empty(1, device=x.device, dtype=torch.float32)
    L.check(L.lib().lnx_quantize_fp8(_p(x2), code_of(x2), C.c_int64(x2.stride(0)), x2.shape[0], x2.shape[1], _p(amax), _p(y), C.c_int64(y.stride(0)), _p(scale), _stream()),
            "lnx_quantize_fp8")
    return y.view(torch.float8_e4m3fn).reshape(x.shape), scale


def gemm_nt_fp8(A8, a_scale, W8, w_scale, out, *, bias=None, act=L.ACT_NONE, aux=None, c2=None, rowscale=None, rows_per_sample=0, res=None):
    """out = epilogue(a_scale * w_scale * A8 . W8^T) with e4m3 operands; see lnx_gemm_nt_fp8."""
    a = L.GemmArgs()
    a.dtype = L.BF16
    a.M, a.N, a.K = A8.shape[0], W8.shape[0], W8.shape[1]
    a.A, a.lda = _p(A8), A8.stride(0)
    a.W, a.ldw = _p(W8), W8.stride(0)
    a.C, a.ldc = _p(out), out.stride(0)
    a.out_f32 = int(out.dtype == torch.float32)
    a.bias = _p(bias)
    a.c2, a.ldc2 = _p(c2), (c2.stride(0) if c2 is not None else 0)
    a.act, a.aux, a.ldaux = act, _p(aux), (aux.stride(0) if aux is not None else 0)
    a.rowscale, a.rows_per_sample = _p(rowscale), rows_per_sample
    a.res, a.ldres = _p(res), (res.stride(0) if res is not None else 0)
    L.check(L.lib().lnx_gemm_nt_fp8(C.byref(a), _p(a_scale), _p(w_scale), _stream()), "lnx_gemm_nt_fp8")
    return out


def quantize_mxfp8(x):
    """(x8, scales): OCP MXFP8 -- e4m3fn elements with one power-of-two (E8M0) scale per 32 consecutive elements of a row.
    x8: torch.float8_e4m3fn [rows, cols]; scales: uint8 [cols/128, rows, 4] (byte j of [ks, r] = block 4 ks + j of row r),
    the layout lnx_gemm_nt_mxfp8 streams.  See lnx_quantize_mxfp8."""
    x2 = x.reshape(-1, x.shape[-1])
    rows, cols = x2.shape
    y = torch.empty(rows, cols, device=x.device, dtype=torch.uint8)
    sc = torch.empty(cols // 128, rows, 4, device=x.device, dtype=torch.uint8)
    L.check(L.lib().lnx_quantize_mxfp8(_p(x2), code_of(x2), C.c_int64(x2.stride(0)), rows, cols, _p(y), C.c_int64(y.stride(0)), _p(sc), _stream()),
            "lnx_quantize_mxfp8")
    return y.view(torch.float8_e4m3fn), sc


def dequantize_mxfp8(x8, scales):
    """fp32 tensor an MXFP8 pair stands for (tests, debugging)."""
    rows, cols = x8.shape
    e = scales.permute(1, 0, 2).reshape(rows, cols // 32).to(torch.float32) - 127.0
    return (x8.to(torch.float32).view(rows, cols // 32, 32) * torch.exp2(e).unsqueeze(-1)).view(rows, cols)


def gemm_nt_mxfp8(A8, a_scales, W8, w_scales, out, *, bias=None, act=L.ACT_NONE, aux=None, c2=None, rowscale=None, rows_per_sample=0, res=None,
                  c8=None, c8_scales=None):
    """out = epilogue(dequant(A8) . dequant(W8)^T) with MXFP8 operands (block scales applied by the matrix core); see lnx_gemm_nt_mxfp8."""
    a = L.GemmArgs()
    a.dtype = L.BF16
    a.M, a.N, a.K = A8.shape[0], W8.shape[0], W8.shape[1]
    a.A, a.lda = _p(A8), A8.stride(0)
    a.W, a.ldw = _p(W8), W8.stride(0)
    a.C, a.ldc = _p(out), out.stride(0)
    a.out_f32 = int(out.dtype == torch.float32)
    a.bias = _p(bias)
    a.c2, a.ldc2 = _p(c2), (c2.stride(0) if c2 is not None else 0)
    a.act, a.aux, a.ldaux = act, _p(aux), (aux.stride(0) if aux is not None else 0)
    a.rowscale, a.rows_per_sample = _p(rowscale), rows_per_sample
    a.res, a.ldres = _p(res), (res.stride(0) if res is not None else 0)
    if c8 is not None:
        a.c8, a.ldc8, a.c8_scales = _p(c8), c8.stride(0), _p(c8_scales)
    assert tuple(a_scales.shape) == (a.K // 128, a.M, 4) and tuple(w_scales.shape) == (a.K // 128, a.N, 4)
    L.check(L.lib().lnx_gemm_nt_mxfp8(C.byref(a), _p(a_scales), _p(w_scales), _stream()), "lnx_gemm_nt_mxfp8")
    return out


def gemm_tn(dY, A, dW, *, M=None, N=None, K=None, lda=None, lddw=None, db=None, a_patch=None, k_perm_c=0, k_store=0, splits=0, dtype=None):
    a = L.WgradArgs()
    a.dtype = dtype if dtype is not None else code_of(dY)
    a.M = M if M is not None else dY.shape[0]
    a.N = N if N is not None else dY.shape[1]
    a.K = K if K is not None else (A.shape[1] if a_patch is None else 4 * a_patch[2])
    a.dY, a.lddy = _p(dY), dY.stride(0)
    a.A, a.lda = _p(A), (lda if lda is not None else (A.stride(0) if a_patch is None else 0))
    if a_patch is not None:
        a.a_mode, a.Hin, a.Win, a.Cin = L.ADDR_PATCH2, *a_patch
    a.dW, a.lddw = _p(dW), (lddw if lddw is not None else dW.stride(0))
    a.k_perm_c, a.db, a.splits, a.k_store = k_perm_c, _p(db), splits, k_store
    L.check(L.lib().lnx_gemm_tn(C.byref(a), _stream()), "lnx_gemm_tn")
    return dW


def layernorm_fwd(x, w, b, y, eps, *, M=None, C_=None, ldx=None, ldy=None, x_map=None, y_map=None, add=None, mean=None, rstd=None, y8=None, y8_scales=None):
    a = L.LnArgs()
    a.M = M if M is not None else x.shape[0]
    a.C = C_ if C_ is not None else x.shape[-1]
    a.eps = eps
    a.x, a.x_dtype, a.ldx, a.x_map = _p(x), code_of(x), (ldx if ldx is not None else x.stride(-2)), _map(x_map)
    a.w, a.b = _p(w), _p(b)
    a.y, a.y_dtype, a.ldy, a.y_map = _p(y), code_of(y), (ldy if ldy is not None else y.stride(-2)), _map(y_map)
    a.add, a.ldadd = _p(add), (add.stride(-2) if add is not None else 0)
    a.mean, a.rstd = _p(mean), _p(rstd)
    if y8 is not None:
        a.y8, a.ldy8, a.y8_scales = _p(y8), y8.stride(0), _p(y8_scales)
    L.check(L.lib().lnx_layernorm_fwd(C.byref(a), _stream()), "lnx_layernorm_fwd")
    return y


def layernorm_bwd(dy, x, w, mean, rstd, dx, *, M=None, C_=None, lddy=None, ldx=None, lddx=None, dy_map=None, x_map=None, gin=None,
                  ldgin=None, dw=None, db=None, relu_mask=False, ws=None, dx2=None, dx2_rowscale=None, dx2_rows_per_sample=0, dx2_8=None, dx2_8_scales=None, defer=False):
    a = L.LnBwdArgs()
    a.M = M if M is not None else dy.shape[0]
    a.C = C_ if C_ is not None else x.shape[-1]
    a.dy, a.dy_dtype, a.lddy, a.dy_map = _p(dy), code_of(dy), (lddy if lddy is not None else dy.stride(-2)), _map(dy_map)
    a.x, a.x_dtype, a.ldx, a.x_map = _p(x), code_of(x), (ldx if ldx is not None else x.stride(-2)), _map(x_map)
    a.w, a.mean, a.rstd = _p(w), _p(mean), _p(rstd)
    a.gin, a.ldgin = _p(gin), ((ldgin if ldgin is not None else gin.stride(-2)) if gin is not None else 0)
    a.dx, a.dx_dtype, a.lddx = _p(dx), code_of(dx), (lddx if lddx is not None else dx.stride(-2))
    a.dw, a.db, a.relu_mask = _p(dw), _p(db), int(relu_mask)
    a.ws, a.ws_floats = _p(ws), (ws.numel() if ws is not None else 0)
    a.defer = int(defer)  # column-sum reduce postponed to layernorm_bwd_flush() (needs ws, one per pending call)
    if dx2 is not None:  # second output: rowscale[m // rows_per_sample] * dx in dx2's type (+ its MXFP8 copy)
        a.dx2, a.dx2_dtype, a.lddx2 = _p(dx2), code_of(dx2), dx2.stride(-2)
        a.dx2_rowscale, a.dx2_rows_per_sample = _p(dx2_rowscale), int(dx2_rows_per_sample)
        if dx2_8 is not None:
            a.dx2_8, a.dx2_8_scales, a.lddx2_8 = _p(dx2_8), _p(dx2_8_scales), dx2_8.stride(-2)
    L.check(L.lib().lnx_layernorm_bwd(C.byref(a), _stream()), "lnx_layernorm_bwd")
    return dx


def layernorm_bwd_flush():
    L.check(L.lib().lnx_layernorm_bwd_flush(_stream()), "lnx_layernorm_bwd_flush")


def dwconv7(x, w49, bias, y, *, flip=False, res=None):
    B, H, W, Cc = x.shape
    a = L.DwconvArgs()
    a.B, a.H, a.W, a.C = B, H, W, Cc
    a.x, a.x_dtype, a.w49, a.bias = _p(x), code_of(x), _p(w49), _p(bias)
    a.flip, a.res, a.y, a.y_dtype = int(flip), _p(res), _p(y), code_of(y)
    L.check(L.lib().lnx_dwconv7_fwd(C.byref(a), _stream()), "lnx_dwconv7_fwd")
    return y


def dwconv7_wgrad(x, dy, dw, db):
    B, H, W, Cc = x.shape
    a = L.DwconvWgradArgs()
    a.B, a.H, a.W, a.C = B, H, W, Cc
    a.x, a.x_dtype, a.dy, a.dy_dtype, a.dw, a.db = _p(x), code_of(x), _p(dy), code_of(dy), _p(dw), _p(db)
    L.check(L.lib().lnx_dwconv7_wgrad(C.byref(a), _stream()), "lnx_dwconv7_wgrad")


def rope_cos_table(freqs, H, W, out=None, dsin=None):
    """cos(theta) table [H*W, heads, 32]; `dsin` (optional, [2, H*W, heads, 32]) receives -t_x sin(theta), -t_y sin(theta)."""
    heads = freqs.shape[1]
    if out is None:
        out = torch.empty(H * W, heads, 32, device=freqs.device, dtype=torch.float32)
    L.check(L.lib().lnx_rope_cos_table(_p(freqs), heads, H, W, _p(out), _p(dsin) if dsin is not None else None, _stream()), "lnx_rope_cos_table")
    return out


def rope_cos_tables(entries):
    """The tables of several blocks in one launch: entries = [(freqs [2,heads,32], H, W, cos_out, dsin_out or None), ...]."""
    arr = (L.RopeTable * len(entries))()
    for t, (freqs, H, W, out, dsin) in zip(arr, entries):
        t.freqs, t.cos_out, t.dsin_out = _p(freqs), _p(out), _p(dsin) if dsin is not None else None
        t.heads, t.H, t.W = freqs.shape[1], H, W
    L.check(L.lib().lnx_rope_cos_tables(arr, len(entries), _stream()), "lnx_rope_cos_tables")


def attn_bwd_ws_floats(B, N, heads):
    fn = L.lib().lnx_attn_bwd_ws_floats
    fn.restype = C.c_int64
    return int(fn(B, N, heads))


def attn_fwd(qkv, cos_tab, o, lse, B, N, E, heads, *, drop_mask=None, drop_rate=0.0):
    a = L.AttnArgs()
    a.dtype, a.B, a.N, a.E, a.heads = code_of(qkv), B, N, E, heads
    a.qkv, a.cos_tab, a.o, a.lse = _p(qkv), _p(cos_tab), _p(o), _p(lse)
    if drop_mask is not None:
        a.drop_mask, a.drop_inv_keep = _p(drop_mask), 1.0 / (1.0 - drop_rate)
    L.check(L.lib().lnx_attn_fwd(C.byref(a), _stream()), "lnx_attn_fwd")


def attn_bwd(qkv, cos_tab, o, lse, d_o, dqkv, delta, B, N, E, heads, *, dsin=None, dfreqs=None, drop_mask=None, drop_rate=0.0, defer_freqs=False):
    """dq/dk/dv into dqkv; with image tokens (E < N) also dfreqs [2, heads, 32] += the gradient of the RoPE frequencies
    (`dsin` = the second table of rope_cos_table).  defer_freqs: the fold into dfreqs waits for attn_bwd_flush(); the returned
    workspace (and dfreqs) must be kept alive until then."""
    a = L.AttnBwdArgs()
    a.dtype, a.B, a.N, a.E, a.heads = code_of(qkv), B, N, E, heads
    a.qkv, a.cos_tab, a.o, a.lse = _p(qkv), _p(cos_tab), _p(o), _p(lse)
    ws = torch.empty(max(attn_bwd_ws_floats(B, N, heads), 1), device=qkv.device, dtype=torch.float32)
    a.d_o, a.dqkv, a.freq_ws, a.delta = _p(d_o), _p(dqkv), _p(ws), _p(delta)
    if dsin is not None:
        a.dsin_tab, a.dfreqs = _p(dsin), _p(dfreqs)
    if drop_mask is not None:
        a.drop_mask, a.drop_inv_keep = _p(drop_mask), 1.0 / (1.0 - drop_rate)
    a.defer_freqs = int(bool(defer_freqs))
    L.check(L.lib().lnx_attn_bwd(C.byref(a), _stream()), "lnx_attn_bwd")
    return ws


def attn_bwd_flush():
    """Fold every postponed freqs gradient of this thread (one launch)."""
    L.check(L.lib().lnx_attn_bwd_flush(_stream()), "lnx_attn_bwd_flush")


def im2col_stem(x, patches):
    B, Cin, H, W = x.shape
    L.check(L.lib().lnx_im2col_stem(_p(x), B, Cin, H, W, _p(patches), code_of(patches), patches.stride(0), _stream()), "lnx_im2col_stem")


def stem_fwd(x, w, bias, ln_w, ln_b, y, *, patches=None, pre=None, mean=None, rstd=None, eps=1e-6):
    """conv 4x4/4 + bias (bf16 output) + channels-first LayerNorm in one launch: x fp32 [B, Cin, H, W], w bf16 [Cout, 64]"""
    a = L.StemArgs()
    a.x, a.w, a.bias, a.ln_w, a.ln_b = _p(x), _p(w), _p(bias), _p(ln_w), _p(ln_b)
    a.patches, a.pre, a.y, a.mean, a.rstd = _p(patches), _p(pre), _p(y), _p(mean), _p(rstd)
    a.B, a.Cin, a.H, a.W = x.shape
    a.Cout, a.eps = w.shape[0], eps
    L.check(L.lib().lnx_stem_fwd(C.byref(a), _stream()), "lnx_stem_fwd")


def scale_cast(inp, out, M, Cc, *, ldin=None, in_map=None, rowscale=None, rows_per_sample=0, ldout=None):
    L.check(L.lib().lnx_scale_cast(_p(inp), C.c_int64(ldin if ldin is not None else inp.stride(-2)), _map(in_map), _p(rowscale), rows_per_sample,
                                   _p(out), code_of(out), C.c_int64(ldout if ldout is not None else out.stride(-2)), M, Cc, _stream()), "lnx_scale_cast")


def layerscale_bwd(g, z, gamma, rowscale, rows_per_sample, dz, dgamma, M, Cc):
    L.check(L.lib().lnx_layerscale_bwd(_p(g), _p(z), code_of(z), _p(gamma), _p(rowscale), rows_per_sample, _p(dz), _p(dgamma), M, Cc, _stream()),
            "lnx_layerscale_bwd")


def fill_rows(vec, out, ldout, row_map, M, Cc):
    L.check(L.lib().lnx_fill_rows(_p(vec), _p(out), C.c_int64(ldout), _map(row_map), M, Cc, _stream()), "lnx_fill_rows")


def colsum_rows(inp, ldin, row_map, out, M, Cc):
    L.check(L.lib().lnx_colsum_rows(_p(inp), C.c_int64(ldin), _map(row_map), _p(out), M, Cc, _stream()), "lnx_colsum_rows")


def agg2_fwd(a, b, w2, bias1, out, M, Cc):
    L.check(L.lib().lnx_agg2_fwd(_p(a), _p(b), _p(w2), _p(bias1), _p(out), M, Cc, _stream()), "lnx_agg2_fwd")


def agg2_bwd(dout, a, b, w2, da, db, dw2, dbias1, M, Cc):
    L.check(L.lib().lnx_agg2_bwd(_p(dout), _p(a), _p(b), _p(w2), _p(da), _p(db), _p(dw2), _p(dbias1), M, Cc, _stream()), "lnx_agg2_bwd")


def pack_meta(meta, off, dim, out):
    L.check(L.lib().lnx_pack_meta(_p(meta), meta.shape[1], off, dim, _p(out), code_of(out), meta.shape[0], _stream()), "lnx_pack_meta")


def convmlp_fwd(ln, w1, b1, w2, b2, gamma, x, out, *, rowscale=None, rows_per_sample=0, z=None, y=None, ln_w=None, ln_b=None, ln_eps=1e-6,
                ln_out=None, mean=None, rstd=None):
    """`y` given (and `ln` None): the kernel applies the block LayerNorm (ln_w, ln_b, ln_eps) to y itself and writes ln_out / mean / rstd."""
    a = L.ConvMlpArgs()
    src = ln if y is None else y
    a.dtype, a.M, a.C = code_of(src), src.shape[0], src.shape[1]
    a.ln, a.w1, a.b1, a.w2, a.b2, a.gamma = _p(ln), _p(w1), _p(b1), _p(w2), _p(b2), _p(gamma)
    a.rowscale, a.rows_per_sample, a.x, a.out, a.z = _p(rowscale), rows_per_sample, _p(x), _p(out), _p(z)
    a.y, a.ln_w, a.ln_b, a.ln_eps, a.ln_out, a.mean, a.rstd = _p(y), _p(ln_w), _p(ln_b), ln_eps, _p(ln_out), _p(mean), _p(rstd)
    L.check(L.lib().lnx_convmlp_fwd(C.byref(a), _stream()), "lnx_convmlp_fwd")
    return out


def layerscale_apply_wgrad(s, t, w, b, gamma, dw, db, dgamma):
    """dw[c, :] += gamma[c] s[c, :], db[c] += gamma[c] t[c], dgamma[c] += sum_k w[c, k] s[c, k] + b[c] t[c]  (include/lnx.h: the
    LayerScale gradient without z; s / t = the pwconv2 weight / bias gradient of dY = rowscale * g, in zeroed scratch)."""
    L.check(L.lib().lnx_layerscale_apply_wgrad(_p(s), _p(t), C.c_int64(s.stride(0)), _p(w), _p(b), C.c_int64(w.stride(0)), _p(gamma), _p(dw), _p(db),
                                               C.c_int64(dw.stride(0)), _p(dgamma), w.shape[0], w.shape[1], _stream()), "lnx_layerscale_apply_wgrad")


def convmlp_bwd(g, ln, z, w1, b1, w2t, w1t, gamma, act, dh, dz, dln, dgamma, *, rowscale=None, rows_per_sample=0, y=None, ln_w=None, mean=None,
                rstd=None, d_ln_w=None, d_ln_b=None, ws=None, dz_plain=False):
    """`y` given: the LayerNorm backward runs in the kernel too -- `dln` receives the gradient wrt y, d_ln_w / d_ln_b are accumulated."""
    a = L.ConvMlpBwdArgs()
    a.dtype, a.M, a.C = code_of(ln), ln.shape[0], ln.shape[1]
    a.g, a.ln, a.z, a.w1, a.b1, a.w2t, a.w1t, a.gamma = _p(g), _p(ln), _p(z), _p(w1), _p(b1), _p(w2t), _p(w1t), _p(gamma)
    a.rowscale, a.rows_per_sample = _p(rowscale), rows_per_sample
    a.act, a.dh, a.dz, a.dln, a.dgamma = _p(act), _p(dh), _p(dz), _p(dln), _p(dgamma)
    a.y, a.ln_w, a.mean, a.rstd, a.d_ln_w, a.d_ln_b = _p(y), _p(ln_w), _p(mean), _p(rstd), _p(d_ln_w), _p(d_ln_b)
    a.ws, a.ws_floats = _p(ws), (ws.numel() if ws is not None else 0)
    a.dz_plain = int(dz_plain)
    L.check(L.lib().lnx_convmlp_bwd(C.byref(a), _stream()), "lnx_convmlp_bwd")


# ---- metadata-head chains in one launch per direction (metahead.hip; reference mFormerV1.py:282-311, res_norm_layer.py:23-30) ----
def _meta_head_buffers(B, C_, dev):
    f = lambda *shape: torch.empty(*shape, device=dev, dtype=torch.float32)  # noqa: E731
    return {"t0": f(B, 16), "h0": f(B, C_), "x": f(B, C_), "h1": f(B, C_), "n1": f(B, C_), "h2": f(B, C_),
            "m0": f(B), "r0": f(B), "m1": f(B), "r1": f(B), "m2": f(B), "r2": f(B)}


def meta_heads_fwd(meta, heads, tok, eps=1e-5):
    """heads: list of dicts with `off`, `dim`, `slot` (token row of a sample, 1 + m) and the parameters `w0` [C, dim], `b0`, `ln0_w`, `ln0_b`,
    `w1`, `b1`, `ln1_w`, `ln1_b`, `w2`, `b2`, `ln2_w`, `ln2_b`.  tok: [B, N, C] fp32, row `slot` of every sample is written.
    Returns the per-head activation dicts the backward needs."""
    B = meta.shape[0]
    arr = (L.MetaHeadArgs * len(heads))()
    keep, saved = [], []
    for i, h in enumerate(heads):
        C_ = h["w1"].shape[0]
        w0p = torch.zeros(C_, 16, device=meta.device)
        w0p[:, :h["dim"]] = h["w0"]
        bufs = _meta_head_buffers(B, C_, meta.device)
        t = tok[i] if isinstance(tok, (list, tuple)) else tok
        a = arr[i]
        a.B, a.C, a.dim, a.off, a.meta, a.meta_width, a.eps = B, C_, h["dim"], h["off"], _p(meta), meta.shape[1], eps
        a.w0, a.ldw0, a.b0, a.ln0_w, a.ln0_b = _p(w0p), 16, _p(h["b0"]), _p(h["ln0_w"]), _p(h["ln0_b"])
        a.w1, a.ldw1, a.b1, a.ln1_w, a.ln1_b = _p(h["w1"]), C_, _p(h["b1"]), _p(h["ln1_w"]), _p(h["ln1_b"])
        a.w2, a.ldw2, a.b2, a.ln2_w, a.ln2_b = _p(h["w2"]), C_, _p(h["b2"]), _p(h["ln2_w"]), _p(h["ln2_b"])
        for k_, v in bufs.items():
            setattr(a, k_, _p(v))
        a.tok, a.tok_row_stride, a.tok_row_offset = _p(t), t.shape[1] * C_, h["slot"] * C_
        keep.append(w0p)
        saved.append(bufs)
    L.check(L.lib().lnx_meta_heads_fwd(arr, len(heads), _stream()), "lnx_meta_heads_fwd")
    return saved


def meta_heads_bwd(g, heads, saved, grads):
    """g: [B, N, C] gradient of the token matrix (or a list, one per head); grads: per head a dict of fp32 tensors (torch layouts) that are ADDED to."""
    arr = (L.MetaHeadBwdArgs * len(heads))()
    keep = []
    for i, (h, sv, gr) in enumerate(zip(heads, saved, grads)):
        C_ = h["w1"].shape[0]
        gi = g[i] if isinstance(g, (list, tuple)) else g
        B = gi.shape[0]
        w1t, w2t = h["w1"].t().contiguous(), h["w2"].t().contiguous()
        dp = [torch.empty(B, C_, device=gi.device) for _ in range(3)]
        part = torch.empty(L.lib().lnx_meta_heads_bwd_part_floats(B, C_), device=gi.device)
        a = arr[i]
        a.B, a.C, a.dim = B, C_, h["dim"]
        a.g, a.g_row_stride, a.g_row_offset = _p(gi), gi.shape[1] * C_, h["slot"] * C_
        a.w1t, a.ldw1t, a.w2t, a.ldw2t = _p(w1t), C_, _p(w2t), C_
        a.ln0_w, a.ln1_w, a.ln2_w = _p(h["ln0_w"]), _p(h["ln1_w"]), _p(h["ln2_w"])
        for k_, v in sv.items():
            setattr(a, k_, _p(v))
        a.dp2, a.dp1, a.dp0, a.part = _p(dp[0]), _p(dp[1]), _p(dp[2]), _p(part)
        for k_ in ("w0", "b0", "ln0_w", "ln0_b", "w1", "b1", "ln1_w", "ln1_b", "w2", "b2", "ln2_w", "ln2_b"):
            setattr(a, "d_" + k_, _p(gr[k_]))
        keep += [w1t, w2t, part] + dp
    L.check(L.lib().lnx_meta_heads_bwd(arr, len(heads), _stream()), "lnx_meta_heads_bwd")
    torch.cuda.current_stream().synchronize()  # (the scratch tensors above die with this frame)
