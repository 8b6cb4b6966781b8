"""Weight gradient of the three 2x2 / stride-2 downsample convolutions of mFormerV1_sm (B = 256): the pipelined TN kernel with
the patch gather in its LDS-DMA source addresses against the same product on a pre-gathered (im2col) operand."""
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from linnaeus_amd import _lib as L


def run(dY, A, patch, dW, db, wsb, k_perm_c):
    a = L.WgradArgs()
    M, N = dY.shape
    K = A.shape[1] if patch is None else 4 * patch[2]
    a.dtype, a.M, a.N, a.K = L.BF16, M, N, K
    a.dY, a.lddy, a.A = dY.data_ptr(), dY.stride(0), A.data_ptr()
    if patch is None:
        a.lda = A.stride(0)
    else:
        a.a_mode, a.Hin, a.Win, a.Cin = L.ADDR_PATCH2, *patch
    a.dW, a.lddw, a.db, a.splits, a.k_perm_c, a.k_store = dW.data_ptr(), K, db.data_ptr(), 0, k_perm_c, 0
    a.ws, a.ws_floats = wsb.data_ptr(), wsb.numel()
    L.check(L.lib().lnx_gemm_tn(C.byref(a), C.c_void_p(torch.cuda.current_stream().cuda_stream)), "lnx_gemm_tn")


wsb = torch.empty(L.TN_WS_FLOATS, device="cuda")
for B, H, Cc in ((256, 56, 96), (256, 28, 192), (256, 14, 384)):
    N = 2 * Cc
    x = torch.randn(B, H, H, Cc, device="cuda").bfloat16()
    M = B * (H // 2) ** 2
    dY = torch.randn(M, N, device="cuda").bfloat16()
    A2 = torch.randn(M, 4 * Cc, device="cuda").bfloat16()
    dW = torch.zeros(N, 4 * Cc, device="cuda")
    db = torch.zeros(N, device="cuda")
    for name, fn in (("patch gather", lambda: run(dY, x, (H, H, Cc), dW, db, wsb, Cc)), ("plain operand", lambda: run(dY, A2, None, dW, db, wsb, 0))):
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            fn()
        e1.record()
        torch.cuda.synchronize()
        t = e0.elapsed_time(e1) / 20 * 1e-3
        print(f"M={M} N={N} K={4 * Cc} {name:14s}: {t * 1e6:7.1f} us  {2.0 * M * N * 4 * Cc / t / 1e12:6.1f} TFLOP/s", flush=True)
